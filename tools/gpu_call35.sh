#!/bin/bash
# r2 call 35: calibrated MfmaUtil at the bench batch (128), then the stamped bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
cd /tmp; timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/mfmacal -o m -- python $R/tools/mfma_calib_run.py --batch 128 > $O/final_mfmacal.log 2>&1; cd $R
python tools/pmc_mfma.py $(find /tmp/mfmacal -name "*counter_collection.csv" | head -1) $O/mfma_probe.json > $O/final_mfma_util.md 2>&1; cat $O/final_mfma_util.md; cp profiles/pmc_mfma_util.json $O/pmc_mfma_util_final.json 2>/dev/null
timeout 900 python bench.py > $O/final_bench_b128.json 2> $O/final_bench_b128.err; cut -c1-300 $O/final_bench_b128.json; tail -2 $O/final_bench_b128.err
