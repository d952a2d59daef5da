#!/bin/bash
# round-5 call 17: decoder bias / norm gradients written straight into main_grad (no copy kernels): model + engine tests, bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_bert_gpu.py tests/test_flavours_gpu.py -x -q -k "not 6B" > $O/c17_tests.log 2>&1; tail -2 $O/c17_tests.log
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-b32 > $O/c17_bench.json 2> $O/c17_bench.err; python - <<PY
import json
d=json.loads(open("$O/c17_bench.json").read().strip().splitlines()[-1]); print("bench", d["value"], d["ms_per_step"], d["loss"], d["roofline"]["traffic"]["matches_current_sources"])
PY
timeout 300 python tools/step_sequence.py > $O/c17_step_sequence.md 2>&1; head -3 $O/c17_step_sequence.md; grep -c "copyBuffer" $O/c17_step_sequence.md
