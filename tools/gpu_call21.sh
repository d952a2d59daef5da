#!/bin/bash
# round-5 call 21: the default bench line and the 1-rank RCCL multi-GPU step (graph segments, weak block hook, teardown) on the final tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
( time timeout 600 python bench.py ) > $O/c21_bench_default.json 2> $O/c21_bench_default.err; echo "rc $?"; cut -c1-300 $O/c21_bench_default.json; tail -4 $O/c21_bench_default.err
( time timeout 600 python bench.py --force-dist --steps 6 --warmup 2 --no-cpu-baseline --no-b32 ) > $O/c21_bench_force_dist.json 2> $O/c21_bench_force_dist.err; echo "rc $?"; python - <<PY
import json
d=json.loads(open("$O/c21_bench_force_dist.json").read().strip().splitlines()[-1]); print({k:d.get(k) for k in ("value","ms_per_step","dist_mode","dist_note","rccl_ranks","graph_segments","reduce_buckets","loss")})
PY
tail -3 $O/c21_bench_force_dist.err
