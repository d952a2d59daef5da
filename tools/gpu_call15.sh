#!/bin/bash
# round-5 call 15: store-data hazard fix (wait states after scalar-offset 128-bit stores): tests, row-kernel rates before / after, step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
python tools/probes/tap_on_load_debug.py 2>&1 | grep " rep " > $O/c15_debug_after_fix.txt; cat $O/c15_debug_after_fix.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "rmsnorm or ln_l2 or qk" > $O/c15_tests_kernels.log 2>&1; tail -2 $O/c15_tests_kernels.log
for rows in 2 4; do IVH_BWD_ROWS=$rows timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "rmsnorm_add" > $O/c15_tests_rows$rows.log 2>&1; echo "IVH_BWD_ROWS=$rows: $(tail -1 $O/c15_tests_rows$rows.log)"; done
for lib in "" $R/tools/probes/ab_libs/lib_before_store_settle.so ""; do
  echo "lib=${lib:-current}"; IVH_LIB_PATH=$lib timeout 200 python tools/bench_rows.py rows 2>&1 | grep kernel; IVH_LIB_PATH=$lib timeout 200 python tools/bench_decoder_tail.py 2>&1 | grep '"kernel"' | cut -c1-120
  IVH_LIB_PATH=$lib timeout 200 python tools/probes/rows_hoist_probe.py /tmp/x.pt 2>&1 | grep kernel
done > $O/c15_row_rates.txt 2>&1; cat $O/c15_row_rates.txt
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q > $O/c15_tests_model.log 2>&1; tail -2 $O/c15_tests_model.log
for h in 0 1 0 1; do IVH_TAP_ON_LOAD=$h timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-b32 > $O/c15_bench_t$h.json 2> $O/c15_bench_t$h.err; python - <<PY
import json
d=json.loads(open("$O/c15_bench_t$h.json").read().strip().splitlines()[-1]); print("tap_on_load $h", d["value"], d["ms_per_step"], d["loss"])
PY
done
