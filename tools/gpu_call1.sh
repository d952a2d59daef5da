#!/bin/bash
# GPU box session 1 (round 2): attention kernels first, then the whole GPU suite, then bench A/B + profile.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out
mkdir -p $O
rm -f $O/parity_fullsize.json
echo "== attention tests" ; timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "flash_attn" > $O/c1_attn_tests.log 2>&1; tail -15 $O/c1_attn_tests.log
echo "== attention bench" ; timeout 600 python tools/bench_attn.py > $O/c1_bench_attn.jsonl 2> $O/c1_bench_attn.err; cat $O/c1_bench_attn.jsonl; tail -3 $O/c1_bench_attn.err
echo "== full gpu suite" ; timeout 1500 python -m pytest tests -m gpu -q > $O/c1_pytest_gpu.log 2>&1; tail -40 $O/c1_pytest_gpu.log
echo "== bench B=128 new attn" ; timeout 600 python bench.py --steps 10 --warmup 3 > $O/c1_bench_b128_attn32.json 2> $O/c1_bench_b128_attn32.err; cut -c1-600 $O/c1_bench_b128_attn32.json; tail -3 $O/c1_bench_b128_attn32.err
echo "== bench B=128 old attn" ; timeout 600 python bench.py --steps 10 --warmup 3 --attn-kernel 1 --no-cpu-baseline > $O/c1_bench_b128_attn16.json 2> $O/c1_bench_b128_attn16.err; cut -c1-400 $O/c1_bench_b128_attn16.json
echo "== bench B=32" ; timeout 600 python bench.py --steps 20 --warmup 5 --batch 32 --no-cpu-baseline > $O/c1_bench_b32.json 2> $O/c1_bench_b32.err; cut -c1-400 $O/c1_bench_b32.json
echo "== rocprof" ; cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_c1 -o bench -- python $OLDPWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events > $OLDPWD/$O/c1_prof.log 2>&1; cd $OLDPWD
DB=$(find /tmp/prof_c1 -name "*.db" | head -1); python tools/rocpd_stats.py "$DB" > $O/c1_prof_stats.md 2>> $O/c1_prof.log || ls -R /tmp/prof_c1 | head -30 >> $O/c1_prof.log
head -30 $O/c1_prof_stats.md
