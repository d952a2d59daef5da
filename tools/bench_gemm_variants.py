"""A/B of the 256x256 GEMM's start-up skew / store cost on the forward shapes of one 1B block.  GPU box only."""
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
    M, D, Hm = 32 * 417, 1408, 6144
    ops.set_gemm_kernel(2)
    variants = [("auto", -1, 0), ("stagger0", 0, 0), ("stagger1", 1, 0), ("stagger2", 2, 0), ("stagger4", 4, 0), ("stagger8", 8, 0),
                ("nostore", 0, 1)]
    for name, m, n, k, a_kc, b_kc, pre in (("fwd_fc1", M, Hm, D, True, True, False), ("fwd_fc1_preact", M, Hm, D, True, True, True),
                                           ("fwd_qkv", M, 3 * D, D, True, True, False), ("dgrad_fc2", M, Hm, D, True, False, False),
                                           ("fwd_fc2", M, D, Hm, True, True, False), ("wgrad_fc1", Hm, D, M, False, False, False)):
        a = rnd(m, k) if a_kc else rnd(k, m)
        b = rnd(n, k) if b_kc else rnd(k, n)
        bias = torch.zeros(n, device=DEV) if pre else None
        out = torch.empty((m, n), dtype=torch.bfloat16, device=DEV)
        times = {v[0]: [] for v in variants}
        for r in range(5):
            for vn, st, sk in variants:
                L.ivh_gemm256_debug(st, sk)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(4):
                    if pre:
                        ops.gemm(a, b, bias=bias, act="gelu_erf", want_preact=True)
                    else:
                        ops.gemm(a, b, a_kc=a_kc, b_kc=b_kc, out=out)
                e.record()
                torch.cuda.synchronize()
                if r:
                    times[vn].append(s.elapsed_time(e) / 4 * 1e3)
        print(json.dumps(dict(shape=name, **{vn: round(sorted(t)[len(t) // 2], 1) for vn, t in times.items()},
                              tflops_auto=round(2.0 * m * n * k / (sorted(times["auto"])[2] * 1e-6) / 1e12, 1))), flush=True)
    L.ivh_gemm256_debug(-1, 0)
    ops.set_gemm_kernel(0)


if __name__ == "__main__":
    main()
