#!/bin/bash
# The measurement pass behind profiles/ (run on the GPU box: gpurun -- 'bash tools/gpu_profile.sh <tag>'):
#   1. rocprofv3 --kernel-trace --stats of `bench.py --steps 5`            -> gpurun_out/<tag>_kernel_stats.md / .csv
#   2. PMC traffic, two separate --pmc passes (FETCH_SIZE, WRITE_SIZE)     -> gpurun_out/<tag>_pmc_traffic.md, profiles/pmc_traffic.json
#   3. calibrated MfmaUtil (SQ_VALU_MFMA_BUSY_CYCLES vs a known-rate probe) -> gpurun_out/<tag>_mfma_util.md, profiles/pmc_mfma_util.json
#   4. the bench line again, carrying the freshly stamped traffic / utilisation -> gpurun_out/<tag>_bench.json
# Counters are collected in their own runs (no trace domains next to --pmc).  Copy what should be judged from gpurun_out/ into profiles/.
TAG=${1:-prof}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
cd /tmp
echo "== kernel trace"; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-b32 --no-kernel-events --no-secondary --no-contention > $O/${TAG}_trace.log 2>&1
cd $R
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB > $O/${TAG}_kernel_stats.md 2>&1; fi
CSV=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
if [ -n "$CSV" ]; then cp $CSV $O/${TAG}_kernel_stats.csv; fi
head -24 $O/${TAG}_kernel_stats.md | cut -c1-200
cd /tmp
echo "== pmc traffic"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$TAG -o f -- python $R/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-events --no-b32 --no-secondary --no-contention > $O/${TAG}_pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw_$TAG -o w -- python $R/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-events --no-b32 --no-secondary --no-contention > $O/${TAG}_pmc_write.log 2>&1
cd $R
python tools/pmc_traffic.py $(find /tmp/pf_$TAG -name "*counter_collection.csv" | head -1) $(find /tmp/pw_$TAG -name "*counter_collection.csv" | head -1) > $O/${TAG}_pmc_traffic.md 2>&1
head -16 $O/${TAG}_pmc_traffic.md | cut -c1-200; cp profiles/pmc_traffic.json $O/${TAG}_pmc_traffic.json 2>/dev/null
echo "== mfma calibration"; cd /tmp
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/mfmacal_$TAG -o m -- python $R/tools/mfma_calib_run.py --batch 128 > $O/${TAG}_mfmacal.log 2>&1
cd $R
# per-dispatch durations of the SAME pass (is a kernel slower under counter collection than in the plain trace?)
KT=$(find /tmp/mfmacal_$TAG -name "*kernel_trace.csv" | head -1)
if [ -n "$KT" ]; then python - "$KT" > $O/${TAG}_mfmacal_durations.md <<'PY'
import csv, sys, re, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1], newline="")):
    m = re.search(r"ivh::(\w+)", r["Kernel_Name"])
    acc[m.group(1) if m else r["Kernel_Name"][:40]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("| kernel | dispatches | mean us | median us | max us |\n|---|---:|---:|---:|---:|")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v.sort(); print(f"| `{k}` | {len(v)} | {sum(v)/len(v):.1f} | {v[len(v)//2]:.1f} | {v[-1]:.1f} |")
PY
fi
python tools/pmc_mfma.py $(find /tmp/mfmacal_$TAG -name "*counter_collection.csv" | head -1) $O/mfma_probe.json > $O/${TAG}_mfma_util.md 2>&1
head -20 $O/${TAG}_mfma_util.md | cut -c1-200; cp profiles/pmc_mfma_util.json $O/${TAG}_pmc_mfma_util.json 2>/dev/null
echo "== bench (carrying the stamped counters)"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; cut -c1-300 $O/${TAG}_bench.json; tail -2 $O/${TAG}_bench.err
