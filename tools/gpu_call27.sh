#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bert_gpu.py -x -q -k "stage2" 2>&1 | tail -12
for f in "--batch-text" "--batch-text --graph"; do
timeout 600 python tools/bench_stage2.py --batch 64 --steps 5 --warmup 2 $f 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['launch_mode'][:20], d['batched_text_passes'], d['losses'])"
done
