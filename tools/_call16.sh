#!/bin/bash
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
IVH_PARITY_NOTES=$out/r3c16_notes timeout 1500 python -m pytest tests/test_fp8_gpu.py tests/test_multiproc_gpu.py -q -m gpu -s 2>&1 | tail -40 > $out/r3c16_pytest.log
echo "pytest rc ${PIPESTATUS[0]}" > $out/r3c16_status.txt
for ws in tensor channel; do
  timeout 600 python bench.py --model 6B --fp8 --fp8-weight-scales $ws --steps 8 --warmup 3 --no-cpu-baseline --no-b32 > $out/r3c16_bench_6b_fp8_$ws.json 2> $out/r3c16_bench_6b_fp8_$ws.err
  echo "bench $ws rc $?" >> $out/r3c16_status.txt
done
cat $out/r3c16_status.txt; cat $out/r3c16_pytest.log | tail -25
for ws in tensor channel; do python - $out/r3c16_bench_6b_fp8_$ws.json <<'P'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["dtype"][:90], d["loss"])
P
done
