"""Fine-tuning / inference regime: InternVideo2-1B classifier, 8 x 224^2 clips, NO masking -> L = 2049 tokens per clip, the
long-sequence attention regime (attention is 19 % of the FLOPs instead of 4 %).  Forward + cross-entropy + backward in plain autograd (the
reference's engine_for_finetuning.py drives the model that way), and forward only (inference).  GPU box only.

    python tools/bench_finetune.py [--batch 16] [--steps 5] [--warmup 2] [--frames 8]

One JSON line: clips/s and ms for training-mode forward + backward and for inference forward, the algorithmic TFLOP per clip, the fraction of
the dense bf16 MFMA peak, peak memory.  Random-init weights, synthetic clips, drop_path 0 in the timing (the recipe's 0.3 changes no shape)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import internvideo2 as FT  # noqa: E402

PEAK = 2500.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--classes", type=int, default=400)
    a = ap.parse_args()
    torch.manual_seed(0)
    dev = "cuda"
    with torch.device(dev):
        model = FT.internvideo2_1B_patch14_224(num_frames=a.frames, num_classes=a.classes, drop_path_rate=0.0)
    model.train()
    B, T = a.batch, a.frames
    L = 1 + T * 256
    D, Hm, depth = 1408, 6144, 40
    fwd_flop = depth * (2.0 * L * D * (3 * D + D + 2 * Hm) + 4.0 * L * L * D) + 2.0 * T * 256 * 588 * D
    video = torch.rand((B, 3, T, 224, 224), device=dev).to(torch.bfloat16)
    labels = torch.randint(0, a.classes, (B,), device=dev)
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    train_ms, fwd_ms, bwd_ms, infer_ms = [], [], [], []
    for it in range(a.warmup + a.steps):
        model.zero_grad(set_to_none=True)
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        logits = model(video)
        loss = torch.nn.functional.cross_entropy(logits.float(), labels)
        e1.record()
        loss.backward()
        e2.record()
        torch.cuda.synchronize()
        if it >= a.warmup:
            train_ms.append(e0.elapsed_time(e2)); fwd_ms.append(e0.elapsed_time(e1)); bwd_ms.append(e1.elapsed_time(e2))
    peak_train = torch.cuda.max_memory_allocated() / 1e9
    model.zero_grad(set_to_none=True)
    model.eval()
    with torch.no_grad():
        for it in range(a.warmup + a.steps):
            e0, e1 = ev(), ev()
            e0.record()
            model(video)
            e1.record()
            torch.cuda.synchronize()
            if it >= a.warmup:
                infer_ms.append(e0.elapsed_time(e1))
    t_tr, t_inf = float(np.median(train_ms)), float(np.median(infer_ms))
    print(json.dumps(dict(metric="clips/sec, InternVideo2-1B classifier (fine-tuning regime, no masking), 8x224^2 -> L=2049, bf16, 1 GPU",
                          train_clips_per_s=round(B / t_tr * 1e3, 2), train_ms=round(t_tr, 2), forward_ms=round(float(np.median(fwd_ms)), 2),
                          backward_ms=round(float(np.median(bwd_ms)), 2), train_mfma_frac=round(3 * fwd_flop * B / (t_tr * 1e-3) / 1e12 / PEAK, 4),
                          infer_clips_per_s=round(B / t_inf * 1e3, 2), infer_ms=round(t_inf, 2),
                          infer_mfma_frac=round(fwd_flop * B / (t_inf * 1e-3) / 1e12 / PEAK, 4), batch=B, seq_len=L,
                          tflop_per_clip_fwd=round(fwd_flop / 1e12, 3), attention_share_of_flops=round(depth * 4.0 * L * L * D / fwd_flop, 3),
                          peak_mem_gb_train=round(peak_train, 1), loss=round(float(loss), 4), dtype="bf16", data="synthetic",
                          launch_mode="eager autograd (no HIP graph, no fused optimizer)")), flush=True)


if __name__ == "__main__":
    main()
