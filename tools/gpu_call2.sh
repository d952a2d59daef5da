#!/bin/bash
# round-5 call 2: q/k-norm backward (bytes in flight), 8-bit gelu' exchange: tests + same-box A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "qk_rmsnorm or 8_bit or gemm_epilogues" > $O/c2_tests_kernels.log 2>&1; tail -4 $O/c2_tests_kernels.log
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -s -k "8_bit" > $O/c2_tests_model.log 2>&1; tail -6 $O/c2_tests_model.log
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-b32"
IVH_QKBWD_B16=0 timeout 600 $B > $O/c2_bench_qk_generic.json 2> $O/c2_bench_qk_generic.err
timeout 600 $B > $O/c2_bench_qk_b16.json 2> $O/c2_bench_qk_b16.err
timeout 600 $B --gelu-exchange u8 > $O/c2_bench_qk_b16_gelu_u8.json 2> $O/c2_bench_qk_b16_gelu_u8.err
IVH_QKBWD_B16=0 timeout 600 $B > $O/c2_bench_qk_generic_again.json 2> $O/c2_bench_qk_generic_again.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/c2_bench_*.json")):
    try:
        d = json.load(open(f))
        ok = d["other_kernels"]; g = d["roofline"]["gemm_family"]["by_kernel"]
        print(f.split("c2_bench_")[1], d["ms_per_step"], d["mfma_frac_of_step"], "qk_bwd", ok["qk_rmsnorm_bwd"]["avg_launch_us"], {k.split(" ")[0]: v["avg_launch_us"] for k, v in g.items()})
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "own_shape" > $O/c2_tests_6b16.log 2>&1; tail -5 $O/c2_tests_6b16.log
