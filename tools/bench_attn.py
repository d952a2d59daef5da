"""Attention micro-benchmark: both kernel families (16x16x32 / 32x32x16 MFMA) interleaved in one process, random data.  GPU box only.
Shapes: the InternVideo2-1B training shape (L = 417, 16 heads x 88) at B = 32 and 128, the stage-2 shape (L = 206), B/14 (hd 64) and
the 6B / teacher head dim (128).  One JSON line per (shape, family, direction): median microseconds per launch over interleaved
rounds and achieved TFLOP/s (forward 4 L^2 hd per head; backward 2.5x that), plus the HBM-floor time of the launch (q, k, v read
once, o written once; backward: q, k, v, o, do read once, dq, dk, dv written once) at 6.3 TB/s."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import ops  # noqa: E402
from tools.bench_kernels import rnd  # noqa: E402


def one(fn, iters=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    shapes = [(128, 417, 16, 88), (32, 417, 16, 88), (64, 206, 16, 88), (32, 411, 12, 64), (8, 2049, 16, 88), (16, 417, 25, 128)]
    if len(sys.argv) > 1 and sys.argv[1] == "--quick":
        shapes = shapes[:2]
    for B, L, H, hd in shapes:
        D = H * hd
        M = B * L
        qkv = rnd(M, 3 * D)
        dout = rnd(M, D)
        state = {}
        for impl in (1, 2):
            ops.set_attn_kernel(impl)
            state[impl] = ops.flash_attn_fwd_packed(qkv, B, L, H)
            for _ in range(3):
                ops.flash_attn_fwd_packed(qkv, B, L, H)
                ops.flash_attn_bwd_packed(qkv, state[impl][0], dout, state[impl][1], B, L, H)
        torch.cuda.synchronize()
        times = {(impl, d): [] for impl in (1, 2) for d in ("fwd", "bwd")}
        for _ in range(5):                                   # interleaved rounds: clock / thermal drift hits both families alike
            for impl in (1, 2):
                ops.set_attn_kernel(impl)
                out, lse = state[impl]
                times[(impl, "fwd")].append(one(lambda: ops.flash_attn_fwd_packed(qkv, B, L, H)))
                times[(impl, "bwd")].append(one(lambda: ops.flash_attn_bwd_packed(qkv, out, dout, lse, B, L, H)))
        ops.set_attn_kernel(0)
        fl = 4.0 * B * H * L * L * hd
        for (impl, d), ts in times.items():
            t = statistics.median(ts)
            f = fl if d == "fwd" else 2.5 * fl
            hbm = (4 if d == "fwd" else 8) * M * D * 2
            print(json.dumps(dict(kernel=f"flash_attn_{d}", family="16x16x32" if impl == 1 else "32x32x16", B=B, L=L, H=H, hd=hd,
                                  us=round(t * 1e6, 1), min_us=round(min(ts) * 1e6, 1), tflops=round(f / t / 1e12, 1),
                                  frac_mfma_peak=round(f / t / 2.5e15, 4), hbm_floor_us=round(hbm / 6.3e12 * 1e6, 1))), flush=True)


if __name__ == "__main__":
    main()
