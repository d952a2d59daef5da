#!/bin/bash
# r2 call 40: the N > 1 code path on a 1-rank RCCL group with the final sources; kernel trace of the stage-2 step with grouped text-tower weight gradients
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29541; O=$R/gpurun_out; mkdir -p $O
timeout 400 python bench.py --force-dist --steps 6 --warmup 2 --no-cpu-baseline --no-b32 --no-kernel-events > $O/call40_forcedist.json 2> $O/call40_forcedist.err; python -c "
import json; d=json.load(open('$O/call40_forcedist.json')); print(d['value'], d['ms_per_step'], d['dist_mode'], d['launch_mode'], d['host_enqueue_ms_per_step'], d['reduce_buckets'])"; tail -2 $O/call40_forcedist.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_s2 -o s2 -- python $R/tools/bench_stage2.py --batch-text --graph --group-wgrad --steps 5 --warmup 2 > $O/call40_stage2.json 2> $O/call40_stage2.err
cd $R
cut -c1-200 $O/call40_stage2.json
DB=$(find /tmp/prof_s2 -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB > $O/call40_stage2_kernel_stats.md 2>&1; fi
head -24 $O/call40_stage2_kernel_stats.md | cut -c1-160
