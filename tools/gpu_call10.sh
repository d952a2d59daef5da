#!/bin/bash
# round 2, call 10: full GPU suite after the text tower / recompute / large-operand / torch.ops work, then the bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/call10_pytest.log
cat gpurun_out/call10_pytest.log
IVH_BWD_PARTS=256 timeout 200 python tools/bench_rows.py rows 2>&1 | tail -4 > gpurun_out/call10_rows256.jsonl
cat gpurun_out/call10_rows256.jsonl
timeout 900 python bench.py > gpurun_out/call10_bench.json 2> gpurun_out/call10_bench.err
tail -c 1500 gpurun_out/call10_bench.json
