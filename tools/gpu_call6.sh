#!/bin/bash
# round 2, call 6: MFMA shape probe; backward row-kernel grid sweep
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/bench_rows.py probe > gpurun_out/call6_probe.jsonl 2>&1
cat gpurun_out/call6_probe.jsonl
for p in 512 1024 2048 4096; do
  IVH_BWD_PARTS=$p timeout 300 python tools/bench_rows.py rows 2>&1 | tail -4 >> gpurun_out/call6_rows.jsonl
done
cat gpurun_out/call6_rows.jsonl
