#!/bin/bash
# round-5 call 11: decoder tail kernels: target row requested with y (forward); backward at 2 vs 3 waves per SIMD and 512 / 768 / 1024 workgroups
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "ln_l2" > $O/c11_tests.log 2>&1; tail -2 $O/c11_tests.log
for parts in 512 768; do echo "2 waves, parts $parts"; IVH_BWD_PARTS=$parts timeout 200 python tools/bench_decoder_tail.py 2>&1 | grep -v "^$" | cut -c1-200; done > $O/c11_lnl2_sweep.txt 2>&1
for parts in 512 768 1024; do echo "3 waves, parts $parts"; IVH_LIB_PATH=$R/tools/probes/ab_libs/lib_lnl2_pf_3waves.so IVH_BWD_PARTS=$parts timeout 200 python tools/bench_decoder_tail.py 2>&1 | grep -v "^$" | cut -c1-200; done >> $O/c11_lnl2_sweep.txt 2>&1
cat $O/c11_lnl2_sweep.txt
