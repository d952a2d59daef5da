#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bert_gpu.py tests/test_flavours_gpu.py -x -q -k "mlm or vtm or stage2 or vtc" 2>&1 | tail -4
timeout 600 python tools/bench_stage2.py --batch 64 --steps 5 --warmup 2 --graph > gpurun_out/call20_stage2_graph.json 2> gpurun_out/call20_stage2_graph.err
tail -c 900 gpurun_out/call20_stage2_graph.json; tail -12 gpurun_out/call20_stage2_graph.err
