#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
echo "== attention tests"; timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "flash_attn or probe" > $O/c3_attn_tests.log 2>&1; tail -4 $O/c3_attn_tests.log
echo "== attention bench"; timeout 600 python tools/bench_attn.py > $O/c3_bench_attn.jsonl 2> $O/c3_bench_attn.err; grep 32x32 $O/c3_bench_attn.jsonl
cd /tmp; timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/sqA -o a -- python $R/tools/bench_attn.py --quick > $O/c3_sqA.log 2>&1; cd $R
python tools/pmc_sq.py $(find /tmp/sqA -name "*counter_collection.csv") --match attn32 > $O/c3_attn_sq.md 2>&1; cat $O/c3_attn_sq.md
echo "== bench"; timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/c3_bench_b128.json 2> $O/c3_bench_b128.err; cut -c1-300 $O/c3_bench_b128.json; python -c "
import json; d=json.load(open('$O/c3_bench_b128.json')); print(d['value'], d['ms_per_step'], d['b32']); print({k:(v['avg_launch_us'],v['frac']) for k,v in d['other_kernels'].items()})"
