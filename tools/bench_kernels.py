"""Kernel micro-benchmarks on the InternVideo2-1B shapes (B=32, L=417 -> M=13344).  GPU box only.
Prints one JSON line per kernel: achieved TFLOP/s (GEMM / attention) or GB/s (row-wise kernels)."""
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def rnd(*shape):
    return (torch.rand(*shape, device=DEV) * 2 - 1).to(torch.bfloat16)


def main():
    B, L, D, H, Hm = 32, 417, 1408, 16, 6144
    M = B * L
    res = []
    x = rnd(M, D)
    for name, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", Hm, D), ("fc2", D, Hm)):
        a = rnd(M, K); w = rnd(N, K); dy = rnd(M, N)
        t = timeit(lambda: ops.gemm(a, w))
        res.append(dict(kernel=f"gemm_fwd_{name}", M=M, N=N, K=K, ms=t * 1e3, tflops=2 * M * N * K / t / 1e12))
        t = timeit(lambda: ops.gemm(dy, w, a_kc=True, b_kc=False))          # dX = dY W
        res.append(dict(kernel=f"gemm_dgrad_{name}", M=M, N=K, K=N, ms=t * 1e3, tflops=2 * M * N * K / t / 1e12))
        t = timeit(lambda: ops.gemm(dy, a, a_kc=False, b_kc=False))         # dW = dY^T X
        res.append(dict(kernel=f"gemm_wgrad_{name}", M=N, N=K, K=M, ms=t * 1e3, tflops=2 * M * N * K / t / 1e12))
    a = rnd(M, D); w = rnd(Hm, D); bias = torch.zeros(Hm, device=DEV)
    t = timeit(lambda: ops.gemm(a, w, bias=bias, act="gelu_erf", want_preact=True))
    res.append(dict(kernel="gemm_fc1_bias_gelu_preact", ms=t * 1e3, tflops=2 * M * Hm * D / t / 1e12))
    qkv = rnd(M, 3 * D)
    out, lse = ops.flash_attn_fwd_packed(qkv, B, L, H)
    t = timeit(lambda: ops.flash_attn_fwd_packed(qkv, B, L, H))
    fl = 4 * B * H * L * L * (D // H)
    res.append(dict(kernel="flash_attn_fwd_hd88", ms=t * 1e3, tflops=fl / t / 1e12))
    dout = rnd(M, D)
    t = timeit(lambda: ops.flash_attn_bwd_packed(qkv, out, dout, lse, B, L, H))
    res.append(dict(kernel="flash_attn_bwd_hd88", ms=t * 1e3, tflops=2.5 * fl / t / 1e12))
    r = torch.randn(M, D, device=DEV); g = torch.ones(D, device=DEV); wv = torch.ones(D, device=DEV)
    t = timeit(lambda: ops.rmsnorm_add_fwd(r, x, g, None, L, wv, 1e-6))
    res.append(dict(kernel="rmsnorm_add_fwd", ms=t * 1e3, gbps=M * D * (4 + 2 + 4 + 2) / t / 1e9))
    ro, y, rstd = ops.rmsnorm_add_fwd(r, x, g, None, L, wv, 1e-6)
    dres = torch.randn(M, D, device=DEV)
    t = timeit(lambda: ops.rmsnorm_add_bwd(y, dres, ro, rstd, wv, x, g, None, L))
    res.append(dict(kernel="rmsnorm_add_bwd", ms=t * 1e3, gbps=M * D * (2 + 4 + 4 + 2 + 4 + 2) / t / 1e9))
    t = timeit(lambda: ops.qk_rmsnorm_fwd(qkv, wv, wv, 1e-6))
    res.append(dict(kernel="qk_rmsnorm_fwd", ms=t * 1e3, gbps=M * D * 2 * 4 / t / 1e9))
    t = timeit(lambda: ops.colsum_bf16(dout))
    res.append(dict(kernel="colsum_bf16", ms=t * 1e3, gbps=M * D * 2 / t / 1e9))
    n = 64 * 1024 * 1024
    p = torch.zeros(n, device=DEV); m1 = torch.zeros(n, device=DEV); m2 = torch.zeros(n, device=DEV)
    gr = torch.zeros(n, device=DEV, dtype=torch.bfloat16); sh = torch.zeros(n, device=DEV, dtype=torch.bfloat16)
    t = timeit(lambda: ops.adamw_step(p, m1, m2, gr, sh, 1e-3, 0.9, 0.98, 1e-6, 0.05, 1))
    res.append(dict(kernel="adamw", ms=t * 1e3, gbps=n * 28 / t / 1e9))
    for r_ in res:
        print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r_.items()}))


if __name__ == "__main__":
    main()
