"""fp8 vs bf16 GEMM micro-benchmark on the InternVideo2-6B block shapes (BASELINE configs[4]: width 3200, MLP 12800, 16 x 224^2 frames,
mask 0.8 -> L = 833; B clips -> M = B * 833 rows).  GPU box only.  One JSON line per (layer, direction, dtype): microseconds, TFLOP/s
and the fraction of the dense MFMA peak of that dtype (bf16 2.5 PFLOP/s, fp8 5 PFLOP/s), plus the quantisation passes' GB/s."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import ops  # noqa: E402
from tools.bench_kernels import rnd, timeit  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    M, D, Hm = B * 833, 3200, 12800
    for name, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("fc1", Hm, D), ("fc2", D, Hm)):
        x = rnd(M, K); w = (rnd(N, K).float() * 0.02).to(torch.bfloat16); dy = rnd(M, N)
        xq, xqt, sx = ops.fp8_quantize(x, True); wq, wqt, sw = ops.fp8_quantize(w, True); dq, dqt, sd = ops.fp8_quantize(dy, True)
        fl = 2.0 * M * N * K
        cases = [("fwd", lambda: ops.gemm(x, w), lambda: ops.gemm_fp8(xq, wq, sx, sw)),
                 ("dgrad", lambda: ops.gemm(dy, w, a_kc=True, b_kc=False), lambda: ops.gemm_fp8(dq, wqt, sd, sw, k=N)),
                 ("wgrad", lambda: ops.gemm(dy, x, a_kc=False, b_kc=False), lambda: ops.gemm_fp8(dqt, xqt, sd, sx))]
        for d, f16, f8 in cases:
            t16, t8 = timeit(f16, iters=10, warmup=3), timeit(f8, iters=10, warmup=3)
            ops.set_gemm_fp8_kernel(1)
            t8s = timeit(f8, iters=10, warmup=3)                 # the 128^2 e4m3 kernel, for reference
            ops.set_gemm_fp8_kernel(0)
            print(json.dumps(dict(layer=name, dir=d, M=M, N=N, K=K, bf16_us=round(t16 * 1e6, 1), bf16_tflops=round(fl / t16 / 1e12, 1),
                                  bf16_frac_of_2500=round(fl / t16 / 2.5e15, 4), fp8_us=round(t8 * 1e6, 1), fp8_tflops=round(fl / t8 / 1e12, 1),
                                  fp8_frac_of_5000=round(fl / t8 / 5e15, 4), fp8_128_tflops=round(fl / t8s / 1e12, 1), speedup=round(t16 / t8, 3))), flush=True)
        tq = timeit(lambda: ops.fp8_quantize(x, True), iters=10, warmup=3)
        print(json.dumps(dict(kernel="fp8_quantize (amax + e4m3 + transposed copy)", M=M, K=K, us=round(tq * 1e6, 1),
                              gbps=round(M * K * (2 + 2 + 1 + 1) / tq / 1e9, 1))), flush=True)


if __name__ == "__main__":
    main()
