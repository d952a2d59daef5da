"""Round-6 A/B: the 32x32 attention kernels compiled with / without the packed fp32 instructions (ivh_probe_attn32_unpacked).  GPU box only.
Bitwise equality of out / lse / dq / dk / dv first, then interleaved timing of forward and backward.  One JSON line per shape."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from internvideo_amd import ops  # noqa: E402
from internvideo_amd.lib import call  # noqa: E402
from tools.bench_kernels import rnd  # noqa: E402


def one(fn, iters=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def run(qkv, dout, B, L, H):
    out, lse = ops.flash_attn_fwd_packed(qkv, B, L, H)
    dqkv = ops.flash_attn_bwd_packed(qkv, out, dout, lse, B, L, H)
    return out, lse, dqkv


def main():
    ops.set_attn_kernel(2)
    call("ivh_probe_attn32_pingpong", 0)
    bad = []
    for B, L, H, hd in [(2, 1, 2, 88), (3, 33, 2, 88), (2, 64, 3, 64), (2, 97, 2, 88), (2, 161, 2, 64), (2, 257, 2, 88), (3, 417, 4, 88), (2, 833, 2, 88), (2, 130, 2, 128)]:
        qkv = rnd(B * L, 3 * H * hd)
        dout = rnd(B * L, H * hd)
        call("ivh_probe_attn32_unpacked", 0)
        ref = run(qkv, dout, B, L, H)
        call("ivh_probe_attn32_unpacked", 1)
        got = run(qkv, dout, B, L, H)
        torch.cuda.synchronize()
        for name, a, b in zip(("out", "lse", "dqkv"), ref, got):
            if not torch.equal(a, b):
                bad.append(dict(B=B, L=L, H=H, hd=hd, what=name, max=float((a.float() - b.float()).abs().max())))
    print(json.dumps(dict(check="unpacked_bitwise_vs_packed", mismatches=bad)), flush=True)
    for B, L, H, hd in [(128, 417, 16, 88), (112, 417, 16, 88), (32, 417, 16, 88), (64, 206, 16, 88), (256, 411, 12, 64), (16, 417, 25, 128), (8, 2049, 16, 88)]:
        qkv = rnd(B * L, 3 * H * hd)
        dout = rnd(B * L, H * hd)
        out, lse = ops.flash_attn_fwd_packed(qkv, B, L, H)
        ts = {(u, d): [] for u in (0, 1) for d in ("fwd", "bwd")}
        for u in (0, 1):
            call("ivh_probe_attn32_unpacked", u)
            for _ in range(3):
                run(qkv, dout, B, L, H)
        for _ in range(7):
            for u in (0, 1):
                call("ivh_probe_attn32_unpacked", u)
                ts[(u, "fwd")].append(one(lambda: ops.flash_attn_fwd_packed(qkv, B, L, H)))
                ts[(u, "bwd")].append(one(lambda: ops.flash_attn_bwd_packed(qkv, out, dout, lse, B, L, H)))
        med = {k: statistics.median(v) for k, v in ts.items()}
        print(json.dumps(dict(B=B, L=L, H=H, hd=hd, fwd_packed_us=round(med[(0, "fwd")], 1), fwd_unpacked_us=round(med[(1, "fwd")], 1),
                              bwd_packed_us=round(med[(0, "bwd")], 1), bwd_unpacked_us=round(med[(1, "bwd")], 1),
                              fwd_ratio=round(med[(1, "fwd")] / med[(0, "fwd")], 4), bwd_ratio=round(med[(1, "bwd")] / med[(0, "bwd")], 4))), flush=True)
    call("ivh_probe_attn32_unpacked", 1)
    ops.set_attn_kernel(0)


if __name__ == "__main__":
    main()
