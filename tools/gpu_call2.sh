#!/bin/bash
# GPU box session 2 (round 2): failing tests re-run, attention SQ counters, defer variant A/B, MFMA-util calibration, PMC traffic.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=$R/gpurun_out
mkdir -p $O
echo "== targeted tests"; timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_model_gpu.py tests/test_kernels_gpu.py -m gpu -q -k "rccl or S14 or probe or 1B_student or final_feature" > $O/c2_tests.log 2>&1; tail -8 $O/c2_tests.log
echo "== defer A/B"; timeout 300 python tools/bench_attn.py --quick > $O/c2_attn_nodefer.jsonl 2>/dev/null; IVH_ATTN_DEFER=1 timeout 300 python tools/bench_attn.py --quick > $O/c2_attn_defer.jsonl 2>/dev/null; grep '"fwd"\|fwd' $O/c2_attn_nodefer.jsonl | grep 32x32; grep fwd $O/c2_attn_defer.jsonl | grep 32x32
IVH_ATTN_DEFER=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "flash_attn and mfma32" > $O/c2_attn_defer_tests.log 2>&1; tail -3 $O/c2_attn_defer_tests.log
cd /tmp
rocprofv3 -L > $O/c2_counters_list.txt 2>&1
echo "== SQ pass A"; timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/sqA -o a -- python $R/tools/bench_attn.py --quick > $O/c2_sqA.log 2>&1
echo "== SQ pass B"; timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d /tmp/sqB -o b -- python $R/tools/bench_attn.py --quick > $O/c2_sqB.log 2>&1
echo "== SQ pass C"; timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_WAVE32_LDS --output-format csv -d /tmp/sqC -o c -- python $R/tools/bench_attn.py --quick > $O/c2_sqC.log 2>&1
cd $R
python tools/pmc_sq.py $(find /tmp/sqA /tmp/sqB /tmp/sqC -name "*counter_collection.csv") --match attn > $O/c2_attn_sq.md 2>&1; head -120 $O/c2_attn_sq.md
echo "== mfma calibration"; cd /tmp; timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/mfmacal -o m -- python $R/tools/mfma_calib_run.py --batch 32 > $O/c2_mfmacal.log 2>&1; cd $R
python tools/pmc_mfma.py $(find /tmp/mfmacal -name "*counter_collection.csv" | head -1) $O/mfma_probe.json > $O/c2_mfma_util.md 2>&1; cat $O/c2_mfma_util.md; cp profiles/pmc_mfma_util.json $O/ 2>/dev/null
echo "== pmc traffic"; cd /tmp
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf -o f -- python $R/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-events --no-b32 > $O/c2_pmc_fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw -o w -- python $R/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-events --no-b32 > $O/c2_pmc_write.log 2>&1
cd $R
python tools/pmc_traffic.py $(find /tmp/pf -name "*counter_collection.csv" | head -1) $(find /tmp/pw -name "*counter_collection.csv" | head -1) > $O/c2_pmc_traffic.md 2>&1; head -30 $O/c2_pmc_traffic.md; cp profiles/pmc_traffic.json $O/pmc_traffic_c2.json 2>/dev/null
echo "== bench (new json)"; timeout 900 python bench.py --steps 10 --warmup 3 > $O/c2_bench_b128.json 2> $O/c2_bench_b128.err; cat $O/c2_bench_b128.json; tail -3 $O/c2_bench_b128.err
