#!/bin/bash
# round 6, final sources: the secondary bench lines (6B bf16 / fp8, recipe step with both teachers, 1-rank RCCL multi-GPU step) + the two-process rehearsal.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
C="--no-cpu-baseline --no-b32 --no-secondary --no-contention --no-kernel-events"
timeout 600 python bench.py --model 6B --steps 4 --warmup 2 $C > $O/r6_fin_6b_bf16.json 2> $O/r6_fin_6b_bf16.err; echo "6B bf16 rc $?"
timeout 600 python bench.py --model 6B --fp8 --steps 4 --warmup 2 $C > $O/r6_fin_6b_fp8.json 2> $O/r6_fin_6b_fp8.err; echo "6B fp8 rc $?"
timeout 900 python bench.py --with-teachers --steps 3 --warmup 1 $C > $O/r6_fin_recipe.json 2> $O/r6_fin_recipe.err; echo "recipe rc $?"
timeout 600 python bench.py --force-dist --steps 6 --warmup 2 $C > $O/r6_fin_force_dist.json 2> $O/r6_fin_force_dist.err; echo "force-dist rc $?"
bash tools/gpu_share_rehearsal.sh r6_fin > $O/r6_fin_share.log 2>&1; tail -8 $O/r6_fin_share.log
for f in 6b_bf16 6b_fp8 recipe force_dist; do python - $O/r6_fin_$f.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1], {k:d.get(k) for k in ("value","ms_per_step","mfma_frac_of_step","mfma_frac_nominal_equivalent","dist_mode","n_gpus")})
except Exception as e:
    print(sys.argv[1], "no line", e)
P
done
