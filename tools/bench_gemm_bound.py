"""What bounds the K loop of the 256x256 GEMM: the plain NT kernel with one ingredient removed at a time (results are garbage).
  mode 0 = the kernel, 1 = no MFMAs, 2 = no LDS-DMA inside the loop, 3 = no fragment ds_reads.  GPU box only."""
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
    names = {0: "full", 1: "no_mfma", 2: "no_dma", 3: "no_ds_read", 4: "no_b_ds_read", 5: "b_from_a_regs"}
    for name, m, n, k in (("fwd_fc1", M, Hm, D), ("fwd_fc2", M, D, Hm), ("fwd_qkv", M, 3 * D, D), ("square_8k", 8192, 8192, 8192)):
        a, b = rnd(m, k), rnd(n, k)
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        times = {v: [] for v in names}
        for r in range(6):
            for mode in names:
                L.ivh_gemm256_debug_ablate(mode)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(4):
                    ops.gemm(a, b, out=out)
                e.record()
                torch.cuda.synchronize()
                if r:
                    times[mode].append(s.elapsed_time(e) / 4 * 1e3)
        L.ivh_gemm256_debug_ablate(0)
        med = {names[v]: round(sorted(t)[len(t) // 2], 1) for v, t in times.items()}
        print(json.dumps(dict(shape=name, M=m, N=n, K=k, us=med, tflops_full=round(2.0 * m * n * k / (med["full"] * 1e-6) / 1e12, 1),
                              mfma_floor_us_at_2377TF=round(2.0 * m * n * k / 2.377e15 * 1e6, 1))), flush=True)
    ops.set_gemm_kernel(0)


if __name__ == "__main__":
    main()
