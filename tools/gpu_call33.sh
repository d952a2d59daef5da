#!/bin/bash
# r2 call 33: kernel trace of the stage-2 step (graph + batched text passes)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_s2 -o s2 -- python $R/tools/bench_stage2.py --batch-text --steps 5 --warmup 2 > $O/call33_stage2.json 2> $O/call33_stage2.err
cd $R
cut -c1-300 $O/call33_stage2.json
DB=$(find /tmp/prof_s2 -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB > $O/call33_stage2_kernel_stats.md 2>&1; fi
head -70 $O/call33_stage2_kernel_stats.md | cut -c1-200
