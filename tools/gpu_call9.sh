#!/bin/bash
# round-5 call 9: attention with the first row peeled (L % 64 in {1, 33}): parity on both kernel families + same-box A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fused_ops_gpu.py tests/test_torch_ops_gpu.py -x -q -k "attn or attention or Attention" > $O/c9_tests_attn.log 2>&1; tail -4 $O/c9_tests_attn.log
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "golden or baseline_configs or protocol" > $O/c9_tests_model.log 2>&1; tail -3 $O/c9_tests_model.log
for p in 0 1; do IVH_ATTN_PEEL=$p timeout 300 python tools/bench_attn.py --quick > $O/c9_bench_attn_peel$p.txt 2>&1; echo "peel=$p"; tail -6 $O/c9_bench_attn_peel$p.txt; done
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-b32"
for i in 1 2; do
  IVH_ATTN_PEEL=0 timeout 600 $B > $O/c9_bench_peel0_$i.json 2> $O/c9_bench_peel0_$i.err
  timeout 600 $B > $O/c9_bench_peel1_$i.json 2> $O/c9_bench_peel1_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/c9_bench_peel*.json")):
    try:
        d = json.load(open(f)); ok = d["other_kernels"]
        print(f.split("c9_bench_")[1], d["ms_per_step"], d["mfma_frac_of_step"], d["encoder_fwd_bwd_frac"], "attn fwd/bwd us", ok["flash_attn_fwd"]["avg_launch_us"], ok["flash_attn_bwd"]["avg_launch_us"], "loss", d["loss"])
    except Exception as e:
        print(f, "ERR", e)
PY
