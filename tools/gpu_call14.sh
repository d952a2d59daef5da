#!/bin/bash
# round-5 call 14: tap gradient added inside the norm backward's loads (dres_extra): kernel + model tests, step A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "rmsnorm_add" > $O/c14_tests_kernels.log 2>&1; tail -3 $O/c14_tests_kernels.log
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q -k "not 6B" > $O/c14_tests_model.log 2>&1; tail -3 $O/c14_tests_model.log
for h in 0 1 0 1; do IVH_TAP_ON_LOAD=$h timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-b32 > $O/c14_bench_t$h.json 2> $O/c14_bench_t$h.err; python - <<PY
import json
d=json.loads(open("$O/c14_bench_t$h.json").read().strip().splitlines()[-1]); print("tap_on_load $h", d["value"], d["ms_per_step"], d["loss"])
PY
done
