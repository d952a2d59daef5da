"""Workload for ONE `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass that calibrates its own normalisation (GPU box only):
  1. the known-rate MFMA stream `ivh_probe_mfma_rate` (256 workgroups x 4 waves x iters x 8 MFMAs 32x32x16) timed with HIP events in this
     very process -> its achieved fraction of the 2.5 PFLOP/s dense bf16 peak is known independently of any counter;
  2. one eager InternVideo2-1B training step (B = --batch), so that the GEMM / attention kernels get counter rows in the same CSV.
tools/pmc_mfma.py then scales every kernel's raw BUSY / GUI_ACTIVE ratio by (known fraction of the probe) / (raw ratio of the probe).
    cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <out> -- python tools/mfma_calib_run.py --batch 32
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from internvideo_amd import internvideo2_pretrain as M, ops  # noqa: E402
from internvideo_amd.engine import IVTrainEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "mfma_probe.json"))
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    iters = 20000                                  # 160 k MFMAs per wave ~ 2.2 ms at peak
    for _ in range(2):
        ops.probe_mfma_rate(iters)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fl = ops.probe_mfma_rate(iters); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    t = sorted(ts)[len(ts) // 2]
    probe = dict(kernel="probe_mfma_rate_kernel", flops_per_launch=fl, median_s=t, tflops=fl / t / 1e12, frac_of_2500=fl / t / 2.5e15,
                 mfma_per_wave=8 * iters, expected_busy_cycles_per_simd=32 * 8 * iters, launches=7)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(probe, open(a.out, "w"), indent=1)
    print(json.dumps(probe), flush=True)
    # one eager training step
    torch.manual_seed(0)
    with torch.device(dev):
        model = M.pretrain_internvideo2_1B_patch14_224(drop_path_rate=0.25, num_frames=8, clip_return_layer=6, mae_return_layer=4)
    model.residual_dtype = "bf16"                   # what bench.py runs by default
    model.train()
    eng = IVTrainEngine(model, lr=1.5e-4)
    B, L = a.batch, 417
    video = torch.rand((B, 3, 8, 224, 224), device=dev).to(torch.bfloat16)
    perm = torch.rand((B, 8, 256), device=dev).argsort(-1)
    mask = torch.ones((B, 8, 256), dtype=torch.bool, device=dev)
    mask.scatter_(2, perm[:, :, :52], False)
    mask = torch.cat([torch.zeros((B, 1), dtype=torch.bool, device=dev), mask.reshape(B, -1)], 1).to(torch.uint8)
    unit = lambda *s: torch.nn.functional.normalize(torch.randn(*s, device=dev), dim=-1).to(torch.bfloat16)   # noqa: E731
    tg = (unit(6, B, L, 3200), unit(B, 768), unit(4, B, L - 1, 1408))
    for _ in range(2):
        eng.train_step(video, mask, tg, vis_inv=M.build_gather_indices(mask, dev, L=L, check=False))
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
