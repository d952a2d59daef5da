#!/bin/bash
# round-5 call 4: RCCL teardown probe, epilogue decomposition, q/k-norm backward grid sweep
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
timeout 300 python tools/probes/epi_gemm_bench.py > $O/c4_epi_decomposition.jsonl 2> $O/c4_epi.err; cat $O/c4_epi_decomposition.jsonl
for parts in 256 384 512 768 1024 2048; do echo "parts $parts"; IVH_BWD_PARTS=$parts timeout 120 python tools/bench_qknorm.py 2>&1 | grep bwd; done > $O/c4_qk_bwd_grid_sweep.txt 2>&1; cat $O/c4_qk_bwd_grid_sweep.txt
timeout 1500 python tools/rccl_teardown_probe.py --trials 5 > $O/c4_rccl_teardown.json 2> $O/c4_rccl_teardown.err; cat $O/c4_rccl_teardown.json; tail -3 $O/c4_rccl_teardown.err
