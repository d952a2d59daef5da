#!/bin/bash
# r2 call 31: tail split along K -- parity tests, then on/off timing on the shapes it applies to
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py -m gpu -x -q -k "gemm or fp8" 2>&1 | tail -15 > $O/call31_tests.log; cat $O/call31_tests.log
timeout 300 python tools/bench_gemm_split.py > $O/call31_split.jsonl 2> $O/call31_split.err; cat $O/call31_split.jsonl; tail -3 $O/call31_split.err
