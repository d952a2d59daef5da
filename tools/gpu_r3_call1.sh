#!/bin/bash
# round 3, call 1: full GPU test suite + bench at both residual-stream types + the 1-rank RCCL modes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > $O/r3c1_pytest.log 2>&1; echo "pytest rc $?" > $O/r3c1_status.txt
timeout 300 python bench.py --steps 10 --warmup 3 --residual bf16 > $O/r3c1_bench_bf16res.json 2> $O/r3c1_bench_bf16res.err; echo "bench bf16 rc $?" >> $O/r3c1_status.txt
timeout 300 python bench.py --steps 10 --warmup 3 --residual fp32 --no-cpu-baseline > $O/r3c1_bench_fp32res.json 2> $O/r3c1_bench_fp32res.err; echo "bench fp32 rc $?" >> $O/r3c1_status.txt
timeout 300 python bench.py --steps 8 --warmup 3 --force-dist --dist-mode auto --no-cpu-baseline --no-b32 > $O/r3c1_bench_dist_auto.json 2> $O/r3c1_bench_dist_auto.err; echo "dist auto rc $?" >> $O/r3c1_status.txt
timeout 300 python bench.py --steps 8 --warmup 3 --force-dist --dist-mode eager --no-cpu-baseline --no-b32 --no-kernel-events > $O/r3c1_bench_dist_eager.json 2> $O/r3c1_bench_dist_eager.err; echo "dist eager rc $?" >> $O/r3c1_status.txt
tail -3 $O/r3c1_pytest.log; cat $O/r3c1_status.txt
