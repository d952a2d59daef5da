#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fp8_gpu.py -x -q 2>&1 | tail -5 > gpurun_out/call19_fp8_tests.log
cat gpurun_out/call19_fp8_tests.log
timeout 900 python bench.py --model 6B --batch 16 --steps 3 --warmup 2 --no-cpu-baseline --no-b32 > gpurun_out/call19_6B_bf16.json 2> gpurun_out/call19_6B_bf16.err
head -c 400 gpurun_out/call19_6B_bf16.json; echo; tail -2 gpurun_out/call19_6B_bf16.err
timeout 900 python bench.py --model 6B --batch 16 --steps 3 --warmup 2 --no-cpu-baseline --no-b32 --fp8 > gpurun_out/call19_6B_fp8.json 2> gpurun_out/call19_6B_fp8.err
head -c 400 gpurun_out/call19_6B_fp8.json; echo; tail -2 gpurun_out/call19_6B_fp8.err
