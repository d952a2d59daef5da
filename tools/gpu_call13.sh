#!/bin/bash
# round-5 call 13: gamma / weight columns hoisted into registers in the forward row kernels (bf16 stream): standalone rate, bitwise check, step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
for h in 0 1 2048 1024 0; do IVH_ROWS_HOIST=$h timeout 200 python tools/probes/rows_hoist_probe.py /tmp/rows_$h.pt 2>&1 | grep -v "^$" | cut -c1-200; done > $O/c13_rows_hoist.txt 2>&1
python tools/probes/rows_hoist_probe.py /tmp/rows_0.pt /tmp/rows_1.pt >> $O/c13_rows_hoist.txt 2>&1
python tools/probes/rows_hoist_probe.py /tmp/rows_0.pt /tmp/rows_2048.pt >> $O/c13_rows_hoist.txt 2>&1
cat $O/c13_rows_hoist.txt
for h in 0 1 0 1; do IVH_ROWS_HOIST=$h timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-b32 > $O/c13_bench_h$h.json 2> $O/c13_bench_h$h.err; python - <<PY
import json
d=json.loads(open("$O/c13_bench_h$h.json").read().strip().splitlines()[-1]); print("hoist $h", d["value"], d["ms_per_step"])
PY
done
