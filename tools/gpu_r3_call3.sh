#!/bin/bash
# round 3, call 3: full GPU suite after the fixes, backward grid sweep at 1 / 2 rows, bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > $O/r3c3_pytest.log 2>&1; echo "pytest rc $?" > $O/r3c3_status.txt
: > $O/r3c3_rows.jsonl
for rows in 1 2; do for parts in 256 384 512 640; do
  IVH_BWD_ROWS=$rows IVH_BWD_PARTS=$parts timeout 120 python tools/bench_rows.py rows16 >> $O/r3c3_rows.jsonl 2>> $O/r3c3_rows.err
done; done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r3c3_bench.json 2> $O/r3c3_bench.err; echo "bench rc $?" >> $O/r3c3_status.txt
tail -15 $O/r3c3_pytest.log | cut -c1-200; cat $O/r3c3_status.txt; grep bwd $O/r3c3_rows.jsonl | cut -c1-200
