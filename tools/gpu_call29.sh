#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_flavours_gpu.py tests/test_fp8_gpu.py tests/test_bert_gpu.py -x -q 2>&1 | tail -4
timeout 600 python tools/bench_stage2.py --batch 64 --steps 5 --warmup 2 --batch-text --graph 2>/dev/null | tail -1 > gpurun_out/call29_stage2.json; cut -c1-220 gpurun_out/call29_stage2.json
timeout 600 python tools/bench_stage2.py --batch 64 --steps 5 --warmup 2 2>/dev/null | tail -1 | cut -c1-220
