#!/bin/bash
# round-5 call 5: DPP column sums in the fc2-dgrad epilogue, q/k-norm backward on 768 workgroups: tests + same-box A/B; 1-rank RCCL bench with the real teardown
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_epilogues or qk_rmsnorm or grouped or half_width" > $O/c5_tests_kernels.log 2>&1; tail -3 $O/c5_tests_kernels.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_torch_ops_gpu.py -x -q -k "golden or engine or qk" > $O/c5_tests_model.log 2>&1; tail -3 $O/c5_tests_model.log
timeout 300 python tools/probes/epi_gemm_bench.py > $O/c5_epi_decomposition.jsonl 2> $O/c5_epi.err; grep -E "EPI3|dgrad_plain" $O/c5_epi_decomposition.jsonl
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-b32"
PREV=$R/tools/probes/ab_libs/lib_colsum_shuffles.so
for i in 1 2; do
  IVH_LIB_PATH=$PREV IVH_QKBWD_PARTS=512 timeout 600 $B > $O/c5_bench_prev_$i.json 2> $O/c5_bench_prev_$i.err
  timeout 600 $B > $O/c5_bench_new_$i.json 2> $O/c5_bench_new_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/c5_bench_*.json")):
    try:
        d = json.load(open(f)); g = d["roofline"]["gemm_family"]["by_kernel"]; ok = d["other_kernels"]
        print(f.split("c5_bench_")[1], d["ms_per_step"], d["mfma_frac_of_step"], d["encoder_fwd_bwd_frac"], "qk_bwd", ok["qk_rmsnorm_bwd"]["avg_launch_us"], {k.split(" ")[0]: v["avg_launch_us"] for k, v in g.items()}, "loss", d["loss"])
    except Exception as e:
        print(f, "ERR", e)
PY
timeout 600 python bench.py --steps 4 --warmup 2 --force-dist --no-cpu-baseline --no-b32 --no-kernel-events > $O/c5_bench_force_dist_teardown.json 2> $O/c5_bench_force_dist_teardown.err; echo "force-dist rc=$?"; cut -c1-200 $O/c5_bench_force_dist_teardown.json; tail -2 $O/c5_bench_force_dist_teardown.err
