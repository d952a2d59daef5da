"""Attention micro-benchmark on the InternVideo2-1B shape (L = 417, 16 heads x 88) at B = 32 and 128.  GPU box only.
One JSON line per (B, kernel): time per launch and achieved TFLOP/s (forward 4 L^2 hd per head; backward 2.5x)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import ops  # noqa: E402
from tools.bench_kernels import rnd, timeit  # noqa: E402


def main():
    L, D, H = 417, 1408, 16
    for B in (32, 128):
        M = B * L
        qkv = rnd(M, 3 * D)
        out, lse = ops.flash_attn_fwd_packed(qkv, B, L, H)
        dout = rnd(M, D)
        fl = 4 * B * H * L * L * (D // H)
        t = timeit(lambda: ops.flash_attn_fwd_packed(qkv, B, L, H))
        print(json.dumps(dict(kernel="flash_attn_fwd_hd88", B=B, us=round(t * 1e6, 1), tflops=round(fl / t / 1e12, 1))))
        t = timeit(lambda: ops.flash_attn_bwd_packed(qkv, out, dout, lse, B, L, H))
        print(json.dumps(dict(kernel="flash_attn_bwd_hd88 (delta + dkdv + dq)", B=B, us=round(t * 1e6, 1), tflops=round(2.5 * fl / t / 1e12, 1))))


if __name__ == "__main__":
    main()
