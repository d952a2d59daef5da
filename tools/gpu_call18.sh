#!/bin/bash
# round-5 call 18: what the driver runs at round end, on the final tree: GPU suite, smoke, the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/c18_gpu_suite.log 2>&1; tail -3 $O/c18_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/c18_smoke.log 2>&1; tail -1 $O/c18_smoke.log
( time timeout 900 python bench.py ) > $O/c18_bench_default.json 2> $O/c18_bench_default.err; cut -c1-400 $O/c18_bench_default.json; tail -4 $O/c18_bench_default.err
