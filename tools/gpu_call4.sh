#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
echo "== fp8 tests"; timeout 600 python -m pytest tests/test_fp8_gpu.py -m gpu -q > $O/c4_fp8_tests.log 2>&1; tail -25 $O/c4_fp8_tests.log
echo "== fp8 bench"; timeout 600 python tools/bench_fp8.py 16 > $O/c4_bench_fp8.jsonl 2> $O/c4_bench_fp8.err; cat $O/c4_bench_fp8.jsonl; tail -3 $O/c4_bench_fp8.err
echo "== attn prio/defer A/B"
for v in "0 0" "1 0" "0 1" "1 1"; do set -- $v; echo "PRIO=$1 DEFER=$2"; IVH_ATTN_PRIO=$1 IVH_ATTN_DEFER=$2 timeout 300 python tools/bench_attn.py --quick 2>/dev/null | grep 32x32 | grep '"B": 128' | cut -c1-160; done
