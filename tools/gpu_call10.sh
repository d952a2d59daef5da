#!/bin/bash
# round-5 call 10: the whole GPU suite and smoke on the final tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/r5_gpu_suite_final.log 2>&1; tail -4 $O/r5_gpu_suite_final.log
python __graft_entry__.py --smoke > $O/c10_smoke.log 2>&1; tail -1 $O/c10_smoke.log
timeout 600 python bench.py > $O/c10_bench_default.json 2> $O/c10_bench_default.err; cut -c1-400 $O/c10_bench_default.json
