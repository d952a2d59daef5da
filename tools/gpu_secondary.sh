#!/bin/bash
# GPU suite + the secondary bench lines (BASELINE configs[1], [3], [4]): run on the GPU box, results under gpurun_out/<tag>_*
TAG=${1:-sec}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=gpurun_out; mkdir -p $O
if [ "$2" != "--no-suite" ]; then timeout 1500 python -m pytest tests -m gpu -q > $O/${TAG}_pytest.log 2>&1; echo "pytest rc $?"; tail -8 $O/${TAG}_pytest.log | cut -c1-200; fi
echo "== stage 2"; timeout 400 python bench.py --model stage2-1B --steps 10 --warmup 3 > $O/${TAG}_stage2.json 2> $O/${TAG}_stage2.err; cut -c1-260 $O/${TAG}_stage2.json; tail -2 $O/${TAG}_stage2.err
echo "== B14"; timeout 400 python bench.py --model B14 --batch 256 --steps 20 --warmup 5 --no-cpu-baseline --no-b32 > $O/${TAG}_b14.json 2> $O/${TAG}_b14.err; cut -c1-260 $O/${TAG}_b14.json
echo "== 6B bf16"; timeout 600 python bench.py --model 6B --batch 16 --steps 6 --warmup 2 --no-cpu-baseline > $O/${TAG}_6b_bf16.json 2> $O/${TAG}_6b_bf16.err; cut -c1-260 $O/${TAG}_6b_bf16.json; tail -1 $O/${TAG}_6b_bf16.err
echo "== 6B fp8"; timeout 600 python bench.py --model 6B --batch 16 --fp8 --steps 6 --warmup 2 --no-cpu-baseline > $O/${TAG}_6b_fp8.json 2> $O/${TAG}_6b_fp8.err; cut -c1-260 $O/${TAG}_6b_fp8.json; tail -1 $O/${TAG}_6b_fp8.err
echo "== 1B at the recipe batch 32, kernel trace"; cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b32_$TAG -o bench -- python $OLDPWD/bench.py --batch 32 --steps 10 --warmup 3 --no-cpu-baseline --no-b32 --no-kernel-events > $OLDPWD/$O/${TAG}_b32_trace.log 2>&1; cd $OLDPWD
DB=$(find /tmp/prof_b32_$TAG -name "*.db" | head -1); if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB > $O/${TAG}_b32_kernel_stats.md 2>&1; head -16 $O/${TAG}_b32_kernel_stats.md | cut -c1-160; fi
