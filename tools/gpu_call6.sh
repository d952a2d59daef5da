#!/bin/bash
# round-5 call 6: full GPU suite, power probe, the maintained measurement pass
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/r5_gpu_suite.log 2>&1; tail -4 $O/r5_gpu_suite.log
timeout 600 python tools/step_power_probe.py 5 > $O/r5_step_power_probe.jsonl 2> $O/r5_step_power_probe.err; cat $O/r5_step_power_probe.jsonl | cut -c1-260; tail -2 $O/r5_step_power_probe.err
bash tools/gpu_profile.sh r5
