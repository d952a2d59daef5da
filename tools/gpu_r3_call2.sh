#!/bin/bash
# round 3, call 2: full GPU suite (all failures), the bf16-stream backward's shape sweep, bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > $O/r3c2_pytest.log 2>&1; echo "pytest rc $?" > $O/r3c2_status.txt
: > $O/r3c2_rows.jsonl
for rows in 0 1 2 4; do for parts in 512 768 1024; do
  IVH_BWD_ROWS=$rows IVH_BWD_PARTS=$parts timeout 120 python tools/bench_rows.py rows16 >> $O/r3c2_rows.jsonl 2>> $O/r3c2_rows.err
done; done
timeout 120 python tools/bench_rows.py rows >> $O/r3c2_rows.jsonl 2>> $O/r3c2_rows.err
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r3c2_bench.json 2> $O/r3c2_bench.err; echo "bench rc $?" >> $O/r3c2_status.txt
tail -5 $O/r3c2_pytest.log; cat $O/r3c2_status.txt; cat $O/r3c2_rows.jsonl | cut -c1-200
