"""Half-width tiles (gemm256.hip, HALF) on / off on the GEMMs of the 1B block at the bench batch (B = 128: 53376 rows) and the recipe's
per-GPU batch (B = 32: 13344 rows), the stage-2 tower and the clip decoders: forward (NT), dgrad, fc1 + GELU (EPI 2) and fc2 dgrad x gelu'
(EPI 3).  The kernel choice is left to the launch-time model; `half` lines say whether the plan is active.  GPU box only."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import lib, ops  # noqa: E402
from tools.bench_gemm_vs_hipblaslt import t_of  # noqa: E402
from tools.bench_kernels import rnd  # noqa: E402


def plan(L, m, n, k, b_kc=1):
    d = lib.GemmDesc()
    d.M, d.N, d.K, d.a_kc, d.b_kc, d.batch = m, n, k, 1, b_kc, 1
    d.lda, d.ldb, d.ldc = k, (k if b_kc else n), n
    out = (C.c_int32 * 4)()
    return L.ivh_gemm256_half_plan(C.byref(d), 0, out), list(out)


def main():
    L = lib.load()
    rows = {"b128": 53376, "b32": 13344, "stage2_b64": 13184}
    layers = [("qkv", 4224, 1408), ("proj", 1408, 1408), ("fc1", 6144, 1408), ("fc2", 1408, 6144)]
    only = sys.argv[1:] or list(rows)
    for tag in only:
        m = rows[tag]
        for name, n, k in layers:
            a, w, dy = rnd(m, k), rnd(n, k), rnd(m, n)
            out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
            dx = torch.empty((m, k), dtype=torch.bfloat16, device="cuda")
            bias = torch.zeros(n, device="cuda")
            fl = 2.0 * m * n * k
            line = dict(shape=f"{tag}_{name}", M=m, N=n, K=k, fwd_plan=plan(L, m, n, k), dgrad_plan=plan(L, m, k, n, 0))
            for half in (0, 1, 3, 0, 1, 3):
                L.ivh_gemm256_debug_half(half)
                t_f = t_of(lambda: ops.gemm(a, w, out=out, bias=bias), n=20)
                t_d = t_of(lambda: ops.gemm(dy, w, a_kc=True, b_kc=False, out=dx), n=20)          # [m, n] x [n, k] -> [m, k]: output width k
                key = ("plain", "half_row_round", "half_last", "half_xcd_block")[half]
                line.setdefault(f"fwd_us_{key}", []).append(round(t_f * 1e6, 1))
                line.setdefault(f"dgrad_us_{key}", []).append(round(t_d * 1e6, 1))
                if name == "fc1":                       # EPI 2 (gelu + gelu' copy)
                    t_e = t_of(lambda: ops.gemm(a, w, bias=bias, act="gelu_erf_d", want_preact=True), n=20)
                    line.setdefault(f"fc1_gelu_us_{key}", []).append(round(t_e * 1e6, 1))
                if name == "fc2":                       # EPI 3: dy [m, 1408] x W2 [1408, 6144] * gelu' -> [m, 6144]
                    dact = rnd(m, k)
                    t_e = t_of(lambda: ops.gemm(dy, w, a_kc=True, b_kc=False, dact_in=dact, act="gelu_erf_d", want_colsum=True), n=20)
                    line.setdefault(f"fc2_dgrad_gelu_us_{key}", []).append(round(t_e * 1e6, 1))
            L.ivh_gemm256_debug_half(1)
            print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
