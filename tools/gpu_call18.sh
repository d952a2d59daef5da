#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fp8_gpu.py -x -q 2>&1 | tail -12 > gpurun_out/call18_fp8_tests.log
cat gpurun_out/call18_fp8_tests.log
timeout 600 python tools/bench_fp8.py 16 2>&1 | tail -16 | tee gpurun_out/call18_bench_fp8.jsonl
