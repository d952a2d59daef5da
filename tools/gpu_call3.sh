#!/bin/bash
# round-5 call 3: rcp-free GELU in the 256^2 epilogues (same-box A/B against the previous library), protocol replay, 6B at its own shape
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fused_ops_gpu.py -x -q -k "gemm or mlp or Mlp" > $O/c3_tests_kernels.log 2>&1; tail -3 $O/c3_tests_kernels.log
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -s -k "protocol or golden or baseline_configs" > $O/c3_tests_model.log 2>&1; grep "reference loop replay" $O/c3_tests_model.log; tail -3 $O/c3_tests_model.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-b32"
PREV=$R/tools/probes/ab_libs/lib_gelu_as7126.so
for i in 1 2; do
  IVH_LIB_PATH=$PREV timeout 600 $B > $O/c3_bench_gelu_prev_$i.json 2> $O/c3_bench_gelu_prev_$i.err
  timeout 600 $B > $O/c3_bench_gelu_new_$i.json 2> $O/c3_bench_gelu_new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/c3_bench_*.json")):
    try:
        d = json.load(open(f)); g = d["roofline"]["gemm_family"]["by_kernel"]
        print(f.split("c3_bench_")[1], d["ms_per_step"], d["mfma_frac_of_step"], d["encoder_fwd_bwd_frac"], {k.split(" ")[0]: v["avg_launch_us"] for k, v in g.items()}, "loss", d["loss"])
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "own_shape" > $O/c3_tests_6b16.log 2>&1; tail -2 $O/c3_tests_6b16.log
