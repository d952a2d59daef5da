#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fp8_gpu.py -x -q 2>&1 | tail -12 > gpurun_out/call16_fp8_tests.log
cat gpurun_out/call16_fp8_tests.log
timeout 900 python bench.py --model 6B --batch 16 --steps 3 --warmup 2 --no-cpu-baseline --no-b32 > gpurun_out/call16_6B_bf16.json 2> gpurun_out/call16_6B_bf16.err
head -c 700 gpurun_out/call16_6B_bf16.json; tail -3 gpurun_out/call16_6B_bf16.err
timeout 900 python bench.py --model 6B --batch 16 --steps 3 --warmup 2 --no-cpu-baseline --no-b32 --fp8 > gpurun_out/call16_6B_fp8.json 2> gpurun_out/call16_6B_fp8.err
head -c 700 gpurun_out/call16_6B_fp8.json; tail -3 gpurun_out/call16_6B_fp8.err
