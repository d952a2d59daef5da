"""Grouped wgrad of three 1B blocks at B = 128 (12 problems, 1278 tiles): time per launch for one walk-group size (env IVH_G2_GROUP_G, read once per
process).  GPU box only; needs the measurement patch that makes g2_decode read IVH_G2_GROUP_G: for g in 2 4 8 16 24; do IVH_G2_GROUP_G=$g python tools/probes/wgrad_group_walk_sweep.py; done"""
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
from internvideo_amd import ops
from tools.bench_gemm_vs_hipblaslt import t_of
from tools.bench_kernels import rnd
M = 53376
probs = []
for blk in range(3):
    for n_out, n_in in ((4224, 1408), (1408, 1408), (6144, 1408), (1408, 6144)):
        dy, x = rnd(M, n_out), rnd(M, n_in)
        probs.append((dy, x, torch.empty((n_out, n_in), dtype=torch.bfloat16, device="cuda")))
t = min(t_of(lambda: ops.gemm_grouped(probs, a_kc=False, b_kc=False), n=5) for _ in range(3))
fl = sum(2.0 * M * p[0].shape[1] * p[1].shape[1] for p in probs)
print(json.dumps(dict(group_g=os.environ.get("IVH_G2_GROUP_G", "8"), us=round(t * 1e6, 1), tflops=round(fl / t / 1e12, 1))), flush=True)
