#!/bin/bash
# r2 call 32: tail split (long K only) -- tests, microbench, the step at B = 128 and B = 32 with the split on / off
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py -m gpu -x -q -k "gemm or fp8" 2>&1 | tail -15 > $O/call32_tests.log; cat $O/call32_tests.log
timeout 300 python tools/bench_gemm_split.py > $O/call32_split.jsonl 2> $O/call32_split.err; cut -c1-400 $O/call32_split.jsonl; tail -3 $O/call32_split.err
timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > $O/call32_bench.json 2> $O/call32_bench.err; python -c "
import json; d=json.load(open('$O/call32_bench.json')); print(d['value'], d['ms_per_step'], d['b32'])"; tail -2 $O/call32_bench.err
IVH_NO_SPLIT=1 timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > $O/call32_bench_nosplit.json 2> $O/call32_bench_nosplit.err; python -c "
import json; d=json.load(open('$O/call32_bench_nosplit.json')); print(d['value'], d['ms_per_step'], d['b32'])"; tail -2 $O/call32_bench_nosplit.err
