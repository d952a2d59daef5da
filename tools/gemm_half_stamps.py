"""Per-tile timeline of one workgroup of a 256x256 GEMM launch with half-width tiles: s_memtime stamps (K loop start / end, drain, epilogue end)
of waves 0 and 4 of the workgroup IVH_G2_STAMP_WG, for half tiles off / interleaved / last.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import ops, lib  # noqa: E402

DEV = "cuda"
L = lib.load()
shapes = [(53376, 1408, 6144), (53376, 1408, 1408)]
for (m, n, k) in shapes:
    a = (torch.rand(m, k, device=DEV) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(n, k, device=DEV) * 2 - 1).to(torch.bfloat16)
    out = torch.empty((m, n), dtype=torch.bfloat16, device=DEV)
    for mode in (0, 1, 2):
        L.ivh_gemm256_debug_half(mode)
        for wg in ((0,) if mode == 0 else (0, 1, 2, 3, 4, 6, 15, 251 % 8 + 248)):
            os.environ["IVH_G2_STAMP_WG"] = str(wg)
            buf = torch.zeros(128, dtype=torch.int64, device=DEV)
            for _ in range(2):
                ops.gemm(a, b, out=out)
            L.ivh_gemm256_debug_stamps(buf.data_ptr())
            ops.gemm(a, b, out=out)
            torch.cuda.synchronize()
            L.ivh_gemm256_debug_stamps(None)
            st = [x for x in buf[:64].tolist() if x][:28]
            base = st[0]
            rel = [(x - base) / 100.0 for x in st]             # microseconds (100 MHz constant clock)
            tiles = [rel[i:i + 4] for i in range(0, len(rel), 4)]
            print(f"M={m} N={n} K={k} mode={mode} wg={wg} total {rel[-1]:.0f}: " + " | ".join(f"k {t[1] - t[0]:.0f} e {t[3] - t[1]:.0f} (t0 {t[0]:.0f})" for t in tiles if len(t) == 4), flush=True)
L.ivh_gemm256_debug_half(1)
