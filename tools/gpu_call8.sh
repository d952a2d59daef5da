#!/bin/bash
# round-5 call 8 (kernel sources frozen): attention tests on the rebuilt library, secondary bench lines, the maintained measurement pass (final stamps)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fused_ops_gpu.py tests/test_torch_ops_gpu.py -q -k "attn or attention or Attention or qk" > $O/c8_tests_attn.log 2>&1; tail -3 $O/c8_tests_attn.log
python __graft_entry__.py --smoke > $O/c8_smoke.log 2>&1; tail -2 $O/c8_smoke.log
bash tools/gpu_secondary.sh r5sec --no-suite
bash tools/gpu_profile.sh r5f
