#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/bench_stage2.py --batch 64 --steps 5 --warmup 2 > gpurun_out/call15_stage2.json 2> gpurun_out/call15_stage2.err
tail -c 1200 gpurun_out/call15_stage2.json; tail -5 gpurun_out/call15_stage2.err
timeout 600 python bench.py --model B14 --no-b32 --no-cpu-baseline > gpurun_out/call15_b14.json 2> gpurun_out/call15_b14.err
head -c 900 gpurun_out/call15_b14.json; tail -3 gpurun_out/call15_b14.err
