#!/bin/bash
# round-5 call 16: the measurement pass on the final sources (stamps), then the whole GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
bash tools/gpu_profile.sh r5h > $O/r5h_profile.log 2>&1; tail -5 $O/r5h_profile.log | cut -c1-400
cp profiles/pmc_traffic.json $O/r5h_pmc_traffic.json; cp profiles/pmc_mfma_util.json $O/r5h_pmc_mfma_util.json
timeout 1500 python -m pytest tests/ -q -m gpu -x > $O/r5h_gpu_suite.log 2>&1; tail -3 $O/r5h_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r5h_smoke.log 2>&1; tail -2 $O/r5h_smoke.log
