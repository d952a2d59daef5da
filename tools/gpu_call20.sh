#!/bin/bash
# round-5 call 20: the GPU suite on the final tree (weak engine hooks, capture guard)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/c20_gpu_suite.log 2>&1; tail -3 $O/c20_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/c20_smoke.log 2>&1; tail -1 $O/c20_smoke.log
