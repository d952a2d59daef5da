#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -k "grouped or engine or graphed or distill" 2>&1 | tail -4
timeout 600 python bench.py --model B14 --no-b32 --no-cpu-baseline > gpurun_out/call23_b14.json 2> gpurun_out/call23_b14.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/call23_b14.json').read().split('\n') if l.startswith('{')][-1])
print('B14', d['value'], d['ms_per_step'], d['mfma_frac_of_step'])
PY
timeout 600 python bench.py --no-b32 --no-cpu-baseline --steps 5 > gpurun_out/call23_1b.json 2> gpurun_out/call23_1b.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/call23_1b.json').read().split('\n') if l.startswith('{')][-1])
print('1B', d['value'], d['ms_per_step'], d['mfma_frac_of_step'])
PY
