"""Where does a tile's time go?  s_memtime stamps of workgroup 0 of the 256x256 GEMM (100 MHz constant clock on gfx950).  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import ops, lib  # noqa: E402

DEV = "cuda"
L = lib.load()
ops.set_gemm_kernel(2)
for (m, n, k, skip, cap) in ((13344, 6144, 1408, 0, 0), (13344, 6144, 1408, 0, 64), (13344, 6144, 1408, 0, 8), (13344, 6144, 1408, 1, 64)):
    a = (torch.rand(m, k, device=DEV) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(n, k, device=DEV) * 2 - 1).to(torch.bfloat16)
    buf = torch.zeros(128, dtype=torch.int64, device=DEV)
    L.ivh_gemm256_debug(0, skip)
    L.ivh_gemm256_debug_max_wg(cap)
    for _ in range(3):
        ops.gemm(a, b)
    L.ivh_gemm256_debug_stamps(buf.data_ptr())
    ops.gemm(a, b)
    torch.cuda.synchronize()
    L.ivh_gemm256_debug_stamps(None)
    for g in range(2):
        st = buf[g * 64:(g + 1) * 64].tolist()
        st = [x for x in st if x][:20]
        base = st[0]
        rel = [(x - base) / 1000.0 for x in st]         # kilo-cycles (s_memtime ticks at the shader clock)
        print(f"M={m} N={n} K={k} skip={skip} cap={cap} wave {g * 4}: " + " | ".join(
            " ".join(f"{v:.1f}" for v in rel[i:i + 4]) for i in range(0, len(rel), 4)))
L.ivh_gemm256_debug(-1, 0)
L.ivh_gemm256_debug_max_wg(0)
ops.set_gemm_kernel(0)
