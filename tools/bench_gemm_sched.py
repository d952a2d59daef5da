"""A/B of the two K-loop schedules of the 256x256 GEMM (0 = two-group ping-pong, 1 = rolling) on the 1B step's shapes; checks that the
results are bit-identical.  GPU box only.   python tools/bench_gemm_sched.py [B]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import lib, ops  # noqa: E402
from tools.bench_kernels import rnd  # noqa: E402


def main():
    L = lib.load()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    M, D, Hm = B * 417, 1408, 6144
    ops.set_gemm_kernel(2)
    shapes = [("fwd_qkv", M, 3 * D, D, True, True), ("fwd_proj", M, D, D, True, True), ("fwd_fc1", M, Hm, D, True, True), ("fwd_fc2", M, D, Hm, True, True),
              ("dgrad_fc1", M, D, Hm, True, False), ("dgrad_fc2", M, Hm, D, True, False), ("wgrad_fc1", Hm, D, M, False, False),
              ("ragged", 1000, 520, 776, True, True), ("square_8k", 8192, 8192, 8192, True, True)]
    for name, m, n, k, a_kc, b_kc in shapes:
        a = rnd(m, k) if a_kc else rnd(k, m)
        b = rnd(n, k) if b_kc else rnd(k, n)
        outs, times = {}, {0: [], 1: []}
        for sched in (0, 1):
            L.ivh_gemm256_debug_sched(sched)
            outs[sched] = ops.gemm(a, b, a_kc=a_kc, b_kc=b_kc)
        torch.cuda.synchronize()
        same = bool(torch.equal(outs[0], outs[1]))
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        for r in range(6):
            for sched in (0, 1):
                L.ivh_gemm256_debug_sched(sched)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(4):
                    ops.gemm(a, b, a_kc=a_kc, b_kc=b_kc, out=out)
                e.record()
                torch.cuda.synchronize()
                if r:
                    times[sched].append(s.elapsed_time(e) / 4 * 1e3)
        med = {s_: sorted(t)[len(t) // 2] for s_, t in times.items()}
        fl = 2.0 * m * n * k
        print(json.dumps(dict(shape=name, M=m, N=n, K=k, identical=same, us_pingpong=round(med[0], 1), us_rolling=round(med[1], 1),
                              tflops_pingpong=round(fl / med[0] / 1e6, 1), tflops_rolling=round(fl / med[1] / 1e6, 1),
                              speedup=round(med[0] / med[1], 3))), flush=True)
    L.ivh_gemm256_debug_sched(0)
    ops.set_gemm_kernel(0)


if __name__ == "__main__":
    main()
