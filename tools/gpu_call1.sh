#!/bin/bash
# round-5 call 1: changed tests, bench with the new fields, eager kernel order
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_bert_gpu.py -x -q -k "image_step or stage2_engine_step" > $O/c1_tests_bert.log 2>&1; tail -3 $O/c1_tests_bert.log
timeout 900 python -m pytest tests/test_multiproc_gpu.py -x -q -k "two_rank_step" -s > $O/c1_tests_dp.log 2>&1; tail -5 $O/c1_tests_dp.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "half_width or gemm_layouts" > $O/c1_tests_gemm.log 2>&1; tail -2 $O/c1_tests_gemm.log
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/c1_bench.json 2> $O/c1_bench.err; cut -c1-600 $O/c1_bench.json; tail -2 $O/c1_bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt1 -o kt -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-b32 --no-kernel-events > $O/c1_trace.log 2>&1
cd $R
KT=$(find /tmp/kt1 -name "*kernel_trace.csv" | head -1)
python tools/step_sequence.py $KT > $O/c1_step_sequence.md 2>&1; head -5 $O/c1_step_sequence.md
