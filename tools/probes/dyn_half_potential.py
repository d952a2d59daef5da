"""Round-6 pricing of half-width tiles under device-side row counts (not built): at the row counts DropPath skipping produces at B = 128
(kept clips 104 ... 122 of 128, 417 rows each) time the N-edge GEMMs of the 1B block as the device-count kernels run them today
(short K: whole tiles only; long K: whole tiles + tail split) against the host-planned half-width launch on the same rows.  GPU box only."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from internvideo_amd import lib, ops  # noqa: E402
from tools.bench_gemm_vs_hipblaslt import t_of  # noqa: E402
from tools.bench_kernels import rnd  # noqa: E402


def main():
    L = lib.load()
    ops.set_gemm_kernel(2)
    # (name, output width, contraction, dgrad?)   dgrad: A = dy [m, K], B = W [K, width] rows-contiguous
    shapes = [("proj_fwd", 1408, 1408, False), ("proj_dgrad", 1408, 1408, True), ("qkv_fwd", 4224, 1408, False), ("fc2_fwd", 1408, 6144, False),
              ("qkv_dgrad", 1408, 4224, True), ("fc1_dgrad", 1408, 6144, True)]
    tot = {}
    for kept in (104, 108, 111, 112, 113, 116, 120, 122):
        m = kept * 417
        md = torch.tensor([m], dtype=torch.int32, device="cuda")
        line = dict(kept_clips=kept, M=m)
        for name, n, k, dg in shapes:
            a = rnd(128 * 417, k)
            w = rnd(k, n) if dg else rnd(n, k)
            out = torch.empty((128 * 417, n), dtype=torch.bfloat16, device="cuda")
            a_m = a[:m].contiguous()
            out_m = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
            if dg:
                f_dyn = lambda: ops.gemm(a, w, a_kc=True, b_kc=False, out=out, m_dev=md)                    # noqa: E731
                f_host = lambda: ops.gemm(a_m, w, a_kc=True, b_kc=False, out=out_m)                          # noqa: E731
            else:
                f_dyn = lambda: ops.gemm(a, w, out=out, m_dev=md)                                             # noqa: E731
                f_host = lambda: ops.gemm(a_m, w, out=out_m)                                                  # noqa: E731
            res = {}
            for rep in range(2):
                L.ivh_gemm256_debug_half(1); L.ivh_gemm256_debug_split(1)
                res.setdefault("dyn_today", []).append(t_of(f_dyn, n=20) * 1e6)
                L.ivh_gemm256_debug_split(0)                                                                  # host plan: half-width tiles only
                res.setdefault("host_half", []).append(t_of(f_host, n=20) * 1e6)
                L.ivh_gemm256_debug_half(0)
                res.setdefault("host_plain", []).append(t_of(f_host, n=20) * 1e6)
                L.ivh_gemm256_debug_half(1); L.ivh_gemm256_debug_split(1)
                res.setdefault("host_default", []).append(t_of(f_host, n=20) * 1e6)
            line[name] = {k_: round(min(v), 1) for k_, v in res.items()}
            for k_, v in res.items():
                tot[k_] = tot.get(k_, 0.0) + min(v)
        print(json.dumps(line), flush=True)
    print(json.dumps(dict(sum_over_all_us={k_: round(v, 1) for k_, v in tot.items()},
                          half_over_today=round(tot["host_half"] / tot["dyn_today"], 4), best_host_over_today=round(tot["host_default"] / tot["dyn_today"], 4))))
    L.ivh_gemm256_debug_half(1); L.ivh_gemm256_debug_split(1)
    ops.set_gemm_kernel(0)


if __name__ == "__main__":
    main()
