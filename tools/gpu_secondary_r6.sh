#!/bin/bash
# round 6: kernel traces + bench lines of the secondary BASELINE configs (B/14 student, stage-2 training step), and of the 1B line.
# gpurun -- 'bash tools/gpu_secondary_r6.sh <tag>'
TAG=${1:-sec}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
for M in B14 stage2-1B; do
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_${TAG}_$M -o bench -- python $R/bench.py --model $M --steps 4 --warmup 2 --no-cpu-baseline --no-b32 --no-kernel-events > $O/${TAG}_${M}_trace.log 2>&1
  cd $R
  DB=$(find /tmp/prof_${TAG}_$M -name "*.db" | head -1)
  if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB > $O/${TAG}_${M}_kernel_stats.md 2>&1; fi
  head -30 $O/${TAG}_${M}_kernel_stats.md | cut -c1-220
  python bench.py --model $M --steps 6 --warmup 2 --no-cpu-baseline > $O/${TAG}_${M}_bench.json 2> $O/${TAG}_${M}_bench.err
  tail -c 400 $O/${TAG}_${M}_bench.err
done
