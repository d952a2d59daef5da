"""B = 32 (13344 rows), the long-K GEMMs whose last round the K split (SPLIT) cuts today: is the half-width-tile plan better there?  GPU box only."""
import os, sys, json, torch
sys.path.insert(0, os.getcwd())
from internvideo_amd import lib, ops
from tools.bench_gemm_vs_hipblaslt import t_of
from tools.bench_kernels import rnd
L = lib.load()
m = 13344
for name, n, k in (("fc2_fwd", 1408, 6144), ("dgrad_qkv", 1408, 4224), ("dgrad_fc1", 1408, 6144), ("proj_fwd", 1408, 1408)):
    a, w = rnd(m, k), rnd(n, k)
    wt = w.t().contiguous()
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    res = {}
    for rep in range(3):
        for tag, split, half in (("ksplit", 1, 0), ("half", 0, 1), ("plain", 0, 0)):
            L.ivh_gemm256_debug_split(split); L.ivh_gemm256_debug_half(half)
            if name.startswith("dgrad"):
                t = t_of(lambda: ops.gemm(a, wt, a_kc=True, b_kc=False, out=out), n=20)
            else:
                t = t_of(lambda: ops.gemm(a, w, out=out), n=20)
            res.setdefault(tag, []).append(round(t * 1e6, 1))
    L.ivh_gemm256_debug_split(1); L.ivh_gemm256_debug_half(1)
    print(json.dumps(dict(shape=name, M=m, N=n, K=k, **res)), flush=True)
