#!/bin/bash
# round 2, final measurement pass on the committed sources: bench line, rocprofv3 kernel trace, PMC traffic (2 passes), calibrated MfmaUtil (1 pass),
# then the bench line again so that it carries the freshly stamped traffic / utilisation.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
cd /tmp
echo "== kernel trace"; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_final -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-b32 --no-kernel-events > $O/final_trace.log 2>&1
cd $R
DB=$(find /tmp/prof_final -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocpd_stats.py $DB > $O/final_kernel_stats.md 2>&1; fi
CSV=$(find /tmp/prof_final -name "*kernel_stats.csv" | head -1)
if [ -n "$CSV" ]; then cp $CSV $O/final_kernel_stats.csv; fi
head -30 $O/final_kernel_stats.md
cd /tmp
echo "== pmc traffic"
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o f -- python $R/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-events --no-b32 > $O/final_pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o w -- python $R/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-events --no-b32 > $O/final_pmc_write.log 2>&1
cd $R
python tools/pmc_traffic.py $(find /tmp/pf -name "*counter_collection.csv" | head -1) $(find /tmp/pw -name "*counter_collection.csv" | head -1) > $O/final_pmc_traffic.md 2>&1; head -16 $O/final_pmc_traffic.md; cp profiles/pmc_traffic.json $O/pmc_traffic_final.json 2>/dev/null
echo "== mfma calibration"; cd /tmp; timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/mfmacal -o m -- python $R/tools/mfma_calib_run.py --batch 128 > $O/final_mfmacal.log 2>&1; cd $R
python tools/pmc_mfma.py $(find /tmp/mfmacal -name "*counter_collection.csv" | head -1) $O/mfma_probe.json > $O/final_mfma_util.md 2>&1; cat $O/final_mfma_util.md; cp profiles/pmc_mfma_util.json $O/pmc_mfma_util_final.json 2>/dev/null
echo "== bench (final json)"; timeout 900 python bench.py > $O/final_bench_b128.json 2> $O/final_bench_b128.err; cut -c1-400 $O/final_bench_b128.json; tail -2 $O/final_bench_b128.err
