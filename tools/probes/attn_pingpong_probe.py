"""Round-6 probe: the attention forward as 8-wave workgroups whose two wave groups run one segment apart
(attn32pp_fwd_kernel, flash_attn32.hip) against the shipped one-group kernel.  GPU box only.
Checks bitwise equality of out / lse (the arithmetic and its order are the same), then times the modes interleaved.
One JSON line per shape."""
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


def main():
    ops.set_attn_kernel(2)
    call("ivh_probe_attn32_unpacked", 0)        # mode 0 = the packed one-group kernel (the shipped default is mode 6)
    modes = (0, 4, 6, 7)
    # correctness on ragged / small shapes first
    bad = []
    for B, L, H, hd in [(2, 1, 2, 88), (3, 33, 2, 88), (2, 64, 3, 64), (2, 97, 2, 88), (2, 161, 2, 64), (2, 256, 2, 88), (2, 257, 2, 88), (3, 417, 4, 88), (2, 833, 2, 88),
                        (2, 130, 2, 128)]:
        qkv = rnd(B * L, 3 * H * hd)
        call("ivh_probe_attn32_pingpong", 0)
        call("ivh_probe_attn32_unpacked", 0)
        ref = ops.flash_attn_fwd_packed(qkv, B, L, H)
        for md in (1, 2, 3, 4, 5, 6, 7):
            call("ivh_probe_attn32_pingpong", md)
            got = ops.flash_attn_fwd_packed(qkv, B, L, H)
            torch.cuda.synchronize()
            if hd > 96 and md >= 3:
                continue
            if not (torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])):
                bad.append(dict(B=B, L=L, H=H, hd=hd, mode=md, max_out=float((ref[0].float() - got[0].float()).abs().max()),
                                max_lse=float((ref[1] - got[1]).abs().max())))
    print(json.dumps(dict(check="bitwise_vs_one_group_kernel", mismatches=bad)), flush=True)
    for B, L, H, hd in [(128, 417, 16, 88), (112, 417, 16, 88), (32, 417, 16, 88), (64, 206, 16, 88), (32, 411, 12, 64), (8, 2049, 16, 88)]:
        qkv = rnd(B * L, 3 * H * hd)
        ts = {md: [] for md in modes}
        for md in modes:
            call("ivh_probe_attn32_pingpong", md)
            for _ in range(3):
                ops.flash_attn_fwd_packed(qkv, B, L, H)
        for _ in range(7):
            for md in modes:
                call("ivh_probe_attn32_pingpong", md)
                ts[md].append(one(lambda: ops.flash_attn_fwd_packed(qkv, B, L, H)))
        fl = 4.0 * B * H * L * L * hd
        print(json.dumps(dict(B=B, L=L, H=H, hd=hd, **{f"mode{md}_us": round(statistics.median(ts[md]), 1) for md in modes},
                              **{f"mode{md}_frac_peak": round(fl / (statistics.median(ts[md]) * 1e-6) / 2.5e15, 4) for md in modes})), flush=True)
    call("ivh_probe_attn32_pingpong", 0)
    call("ivh_probe_attn32_unpacked", 1)
    ops.set_attn_kernel(0)


if __name__ == "__main__":
    main()
