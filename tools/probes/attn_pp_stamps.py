"""Segment timeline of the two-wave-group attention forward (attn32pp_fwd_kernel<…, STAMP>): s_memtime of workgroup 0's waves 0 (group A) and 4
(group B, same SIMD) at the start / end of every MFMA segment X(i) and softmax segment Y(i).  GPU box only.  Prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from internvideo_amd import ops  # noqa: E402
from internvideo_amd.lib import call  # noqa: E402
from tools.bench_kernels import rnd  # noqa: E402


def main():
    ops.set_attn_kernel(2)
    B, L, H, hd = 128, 417, 16, 88
    if "--long" in sys.argv:
        B, L = 8, 2049
    qkv = rnd(B * L, 3 * H * hd)
    buf = torch.zeros(512, dtype=torch.int64, device="cuda")
    mode = 4 if "--np" in sys.argv else 1
    call("ivh_probe_attn32_pingpong", mode)
    for _ in range(3):
        ops.flash_attn_fwd_packed(qkv, B, L, H)
    call("ivh_attn32_debug_stamps", buf.data_ptr(), 128)
    ops.flash_attn_fwd_packed(qkv, B, L, H)
    torch.cuda.synchronize()
    call("ivh_attn32_debug_stamps", None, 0)
    call("ivh_probe_attn32_pingpong", 0)
    t = buf.cpu().view(2, 256)
    nt = (L + 63) // 64
    n = min(5 * nt, 255)                      # five stamps per tile: X start, X end, Y start, Y end, after the DMA wait
    t0 = int(t[0, 0])
    rows = []
    for g in range(2):
        st = [int(x) - t0 for x in t[g, :n]]
        rows.append(st)
    segs = {"A": [], "B": []}
    for g, name in enumerate("AB"):
        st = rows[g]
        for i in range(min(nt, 8)):
            x0, x1, y0, y1, w = st[5 * i:5 * i + 5]
            segs[name].append(dict(tile=i, X=[x0, x1], Y=[y0, y1], dma_wait_end=w))
    # phase overlap: A's Y(i) against B's X(i) (global phase 2 i + 1)
    ov = []
    for i in range(min(nt, 8) - 1):
        a, b = segs["A"][i]["Y"], segs["B"][i]["X"]
        ov.append(dict(tile=i, A_Y=a[1] - a[0], B_X=b[1] - b[0], union=max(a[1], b[1]) - min(a[0], b[0])))
    print(json.dumps(dict(shape=[B, L, H, hd], mode=mode, clock="s_memtime ticks", segments=segs, phase_A_softmax_vs_B_mfma=ov)))


if __name__ == "__main__":
    main()
