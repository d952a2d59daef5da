"""GEMM micro-benchmark on the twelve GEMM shapes of one InternVideo2-1B block (B = 32, L = 417 -> M = 13344), both kernels,
interleaved rounds in one process (random bf16 operands).  GPU box only.  One JSON line per (shape, kernel)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import ops  # noqa: E402

DEV = "cuda"


def rnd(*shape):
    return (torch.rand(*shape, device=DEV) * 2 - 1).to(torch.bfloat16)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    M, D, Hm = 32 * 417, 1408, 6144
    shapes = []
    for name, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", Hm, D), ("fc2", D, Hm)):
        shapes.append((f"fwd_{name}", M, N, K, True, True))
        shapes.append((f"dgrad_{name}", M, K, N, True, False))      # dx[M,K] = dy[M,N] W[N,K]
        shapes.append((f"wgrad_{name}", N, K, M, False, False))     # dW[N,K] = dy[M,N]^T x[M,K]
    for name, m, n, k, a_kc, b_kc in shapes:
        a = rnd(m, k) if a_kc else rnd(k, m)
        b = rnd(n, k) if b_kc else rnd(k, n)
        out = torch.empty((m, n), dtype=torch.bfloat16, device=DEV)
        times = {1: [], 2: []}
        for r in range(rounds + 1):
            for kern in (1, 2):
                ops.set_gemm_kernel(kern)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(4):
                    ops.gemm(a, b, a_kc=a_kc, b_kc=b_kc, out=out)
                e.record()
                torch.cuda.synchronize()
                if r:
                    times[kern].append(s.elapsed_time(e) / 4 * 1e-3)
        for kern in (1, 2):
            t = sorted(times[kern])[len(times[kern]) // 2]
            print(json.dumps(dict(shape=name, M=m, N=n, K=k, kernel={1: "128x128", 2: "256x256"}[kern], us=round(t * 1e6, 1),
                                  tflops=round(2.0 * m * n * k / t / 1e12, 1), min_us=round(min(times[kern]) * 1e6, 1))), flush=True)
    ops.set_gemm_kernel(0)


if __name__ == "__main__":
    main()
