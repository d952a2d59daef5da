#!/bin/bash
# round 2, closing pass on the committed sources: GPU suite, tools/gpu_final.sh (trace, PMC, stamped bench line), then the secondary bench lines
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
echo "== gpu suite"; timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/final2_suite.log; cat $O/final2_suite.log
bash tools/gpu_final.sh
cd $R
echo "== stage 2"; timeout 400 python tools/bench_stage2.py --batch-text --graph --group-wgrad --steps 10 --warmup 3 > $O/final2_stage2.json 2> $O/final2_stage2.err; cut -c1-220 $O/final2_stage2.json
echo "== B14"; timeout 400 python bench.py --model B14 --steps 20 --warmup 5 --no-cpu-baseline > $O/final2_b14.json 2> $O/final2_b14.err; cut -c1-200 $O/final2_b14.json
echo "== 6B bf16"; timeout 600 python bench.py --model 6B --batch 16 --steps 6 --warmup 2 --no-cpu-baseline > $O/final2_6b_bf16.json 2> $O/final2_6b_bf16.err; cut -c1-200 $O/final2_6b_bf16.json; tail -1 $O/final2_6b_bf16.err
echo "== 6B fp8"; timeout 600 python bench.py --model 6B --batch 16 --fp8 --steps 6 --warmup 2 --no-cpu-baseline > $O/final2_6b_fp8.json 2> $O/final2_6b_fp8.err; cut -c1-200 $O/final2_6b_fp8.json; tail -1 $O/final2_6b_fp8.err
