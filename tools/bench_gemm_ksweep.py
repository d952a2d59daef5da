"""Time of the 256x256 GEMM as a function of K at fixed M x N: the slope is the cost of a K step, the intercept the per-tile
overhead (prologue, epilogue, tile switch).  GPU box only."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import ops, lib  # noqa: E402

DEV = "cuda"


def rnd(*shape):
    return (torch.rand(*shape, device=DEV) * 2 - 1).to(torch.bfloat16)


def main():
    L = lib.load()
    ops.set_gemm_kernel(2)
    for (m, n) in ((13344, 6144), (13312, 6144), (13344, 1408), (8192, 8192)):
        for skip in (0, 1):
            L.ivh_gemm256_debug(0, skip)
            row = {}
            for k in (704, 1408, 2816, 5632):
                a = rnd(m, k); b = rnd(n, k)
                out = torch.empty((m, n), dtype=torch.bfloat16, device=DEV)
                ts = []
                for r in range(4):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(4):
                        ops.gemm(a, b, out=out)
                    e.record()
                    torch.cuda.synchronize()
                    if r:
                        ts.append(s.elapsed_time(e) / 4 * 1e3)
                row[k] = round(sorted(ts)[1], 1)
            tiles = ((m + 255) // 256) * ((n + 255) // 256)
            rounds = (tiles + 255) // 256
            slope = (row[5632] - row[1408]) / ((5632 - 1408) / 64) / rounds
            icpt = row[1408] / rounds - slope * 22
            print(json.dumps(dict(M=m, N=n, skip_stores=skip, us_by_K=row, tiles=tiles, rounds=rounds,
                                  us_per_kstep_round=round(slope, 3), us_per_tile_overhead=round(icpt, 2),
                                  tflops_K5632=round(2.0 * m * n * 5632 / row[5632] / 1e6, 1))), flush=True)
    L.ivh_gemm256_debug(-1, 0)
    ops.set_gemm_kernel(0)


if __name__ == "__main__":
    main()
