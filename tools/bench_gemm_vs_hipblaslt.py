"""The 256 x 256 GEMM against the vendor library (torch.matmul -> hipBLASLt / rocBLAS) on the 1B step's shapes at B = 128: a reference point for the
power-limited ceiling (profiles/r2_gemm_power_limit_v1.md).  Measurement only -- the product never calls the library.  GPU box only."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import ops  # noqa: E402
from tools.bench_kernels import rnd  # noqa: E402


def t_of(fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / n * 1e-3)
    return sorted(ts)[len(ts) // 2]


def main():
    M, D, Hm = 128 * 417, 1408, 6144
    for name, m, n, k in (("fwd_qkv", M, 3 * D, D), ("fwd_proj", M, D, D), ("fwd_fc1", M, Hm, D), ("fwd_fc2", M, D, Hm), ("square_8k", 8192, 8192, 8192)):
        a, w = rnd(m, k), rnd(n, k)
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        fl = 2.0 * m * n * k
        t_ours = t_of(lambda: ops.gemm(a, w, out=out))
        t_lib = t_of(lambda: torch.matmul(a, w.t(), out=out))
        # dgrad-like (NN) and wgrad-like (TN) layouts
        dy = rnd(m, n)
        dx = torch.empty((m, k), dtype=torch.bfloat16, device="cuda")
        t_ours_d = t_of(lambda: ops.gemm(dy, w, a_kc=True, b_kc=False, out=dx))
        t_lib_d = t_of(lambda: torch.matmul(dy, w, out=dx))
        dw = torch.empty((n, k), dtype=torch.bfloat16, device="cuda")
        t_ours_w = t_of(lambda: ops.gemm(dy, a, a_kc=False, b_kc=False, out=dw), n=3)
        t_lib_w = t_of(lambda: torch.matmul(dy.t(), a, out=dw), n=3)
        print(json.dumps(dict(shape=name, M=m, N=n, K=k,
                              nt_ours_tflops=round(fl / t_ours / 1e12, 1), nt_lib_tflops=round(fl / t_lib / 1e12, 1),
                              dgrad_ours_tflops=round(fl / t_ours_d / 1e12, 1), dgrad_lib_tflops=round(fl / t_lib_d / 1e12, 1),
                              wgrad_ours_tflops=round(fl / t_ours_w / 1e12, 1), wgrad_lib_tflops=round(fl / t_lib_w / 1e12, 1))), flush=True)


if __name__ == "__main__":
    main()
