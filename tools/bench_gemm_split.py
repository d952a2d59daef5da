"""Tail split along K (gemm256.hip, SPLIT) on / off on the GEMMs whose last tile round is mostly empty: the 1B block at the recipe's per-GPU
batch 32 (13344 rows), the stage-2 vision tower (64 x 206 rows), ViT-B/14 at B = 128 and the 6B fc1 at B = 16.  Forward (NT) and dgrad
layouts, kernel choice left to the launch-time model (which knows about the split).  GPU box only."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import lib, ops  # noqa: E402
from tools.bench_gemm_vs_hipblaslt import t_of  # noqa: E402
from tools.bench_kernels import rnd  # noqa: E402


def main():
    L = lib.load()
    shapes = [("1B_b32_proj", 13344, 1408, 1408), ("1B_b32_fc2", 13344, 1408, 6144), ("1B_b32_qkv", 13344, 4224, 1408), ("1B_b32_fc1", 13344, 6144, 1408),
              ("stage2_b64_proj", 13184, 1408, 1408), ("stage2_b64_fc2", 13184, 1408, 6144), ("B14_b128_proj", 52608, 768, 768),
              ("B14_b128_fc2", 52608, 768, 3072), ("6B_b16_fc1", 13328, 12800, 3200), ("6B_b16_fc2", 13328, 3200, 12800),
              ("bert_b192_dense", 6144, 1024, 1024), ("bert_b192_ffn2", 6144, 1024, 4096)]
    for name, m, n, k in shapes:
        a, w, dy = rnd(m, k), rnd(n, k), rnd(m, n)
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        dx = torch.empty((m, k), dtype=torch.bfloat16, device="cuda")
        fl = 2.0 * m * n * k
        line = dict(shape=name, M=m, N=n, K=k)
        for split in (0, 1):
            L.ivh_gemm256_debug_split(split)
            t_f = t_of(lambda: ops.gemm(a, w, out=out), n=20)
            t_d = t_of(lambda: ops.gemm(dy, w, a_kc=True, b_kc=False, out=dx), n=20)          # [m, n] x [n, k] -> [m, k]: output width k
            tag = "split" if split else "plain"
            line[f"fwd_us_{tag}"] = round(t_f * 1e6, 1); line[f"fwd_tflops_{tag}"] = round(fl / t_f / 1e12, 1)
            line[f"dgrad_us_{tag}"] = round(t_d * 1e6, 1); line[f"dgrad_tflops_{tag}"] = round(fl / t_d / 1e12, 1)
        L.ivh_gemm256_debug_split(1)
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
