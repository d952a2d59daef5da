#!/bin/bash
# round 3, call 4: full GPU suite, qk backward A/B, bench + rocprofv3 kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > $O/r3c4_pytest.log 2>&1; echo "pytest rc $?" > $O/r3c4_status.txt
: > $O/r3c4_rows.jsonl
for rows in 0 1; do for parts in 384 512 768; do
  IVH_BWD_ROWS=$rows IVH_BWD_PARTS=$parts timeout 120 python tools/bench_rows.py rows16 >> $O/r3c4_rows.jsonl 2>> $O/r3c4_rows.err
done; done
timeout 300 python bench.py --steps 10 --warmup 3 > $O/r3c4_bench.json 2> $O/r3c4_bench.err; echo "bench rc $?" >> $O/r3c4_status.txt
tail -12 $O/r3c4_pytest.log | cut -c1-200; cat $O/r3c4_status.txt; grep qk $O/r3c4_rows.jsonl | cut -c1-200
