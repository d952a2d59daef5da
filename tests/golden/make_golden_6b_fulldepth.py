"""Full-DEPTH golden digest of BASELINE configs[4]'s encoder: the REFERENCE's own `PretrainInternVideo2` at the 6B geometry
(pretrain_internvideo2_6B_patch14_224: 48 blocks x 3200, 25 heads of 128, mlp_ratio 4, clip_return_layer 6, mae_return_layer 4), fp32 CPU,
unfused path, FORWARD only (the depth-2 digest student_6Bshape pins the width, this one pins all 48 blocks).

    python tests/golden/make_golden_6b_fulldepth.py      (authoring container only: needs /root/reference; ~35 GB of RAM, a few minutes)

4 frames of 224^2 with 52 visible patches per frame (L = 209) instead of the recipe's 16 (L = 833): depth, width, heads and every weight
shape are the real model's; the shorter sequence keeps the CPU forward and the GPU test in seconds (attention at hd 128 / L = 833 is
measured separately by bench.py --model 6B).  Parameters: oracle.iter_synthetic_params(cfg, seed 0, gamma 0.3) streamed tensor by tensor
into the reference module (LayerScale 0.3: every one of the 48 blocks carries signal without the residual stream drowning the first ones);
inputs: oracle.synthetic_batch(cfg, 1, 52, seed 0).  Stored: the digest of make_golden_fullsize (first 3 rows + 16 fixed projections of
every row of the three outputs) and the four losses  ->  tests/golden/student_6B_fulldepth_digest.npz."""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_loader  # noqa: E402
from make_golden_fullsize import digest  # noqa: E402
from oracle import internvideo2_oracle as O  # noqa: E402

GAMMA = 0.3


def config(frames: int = 4):
    return O.StudentConfig(embed_dim=3200, depth=48, num_heads=25, mlp_ratio=4.0, num_frames=frames, attn_pool_num_heads=16, clip_embed_dim=768,
                           clip_teacher_embed_dim=3200, clip_teacher_final_dim=768, clip_return_layer=6, mae_teacher_embed_dim=1408, mae_return_layer=4)


def main():
    # `--frames 16`: configs[4] at its OWN shape -- 16 x 224^2, 52 visible patches per frame, L = 833
    # (single_modality/scripts/pretraining/6B_pt.sh:47-50) -> tests/golden/student_6B_fulldepth_16f_digest.npz
    frames = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 4
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = config(frames)
    t0 = time.time()
    model = ref_loader.build_reference_student(cfg)
    sd = dict(model.named_parameters())
    n = 0
    with torch.no_grad():
        for k, t in O.iter_synthetic_params(cfg, seed=0, gamma=GAMMA):
            sd[k].copy_(t.reshape(sd[k].shape))
            n += t.numel()
    assert n == sum(p.numel() for p in model.parameters()), "a parameter was not filled"
    print(f"reference 6B model built and filled ({n / 1e9:.2f} G parameters) in {time.time() - t0:.0f} s", flush=True)
    video, mask, targets = O.synthetic_batch(cfg, 1, 52, seed=0)
    model.eval()                                            # drop_path 0 either way; no dropout in the model
    t0 = time.time()
    with torch.no_grad():
        oc, of, om = model(video, torch.from_numpy(mask))
    tc, tf, tm = targets
    l_mid = (2 - 2 * (oc * tc).sum(dim=-1)).mean(); l_fin = (2 - 2 * (of * tf).sum(dim=-1)).mean(); l_mae = (2 - 2 * (om * tm).sum(dim=-1)).mean()
    loss = l_mid + l_fin + l_mae
    print(f"forward {time.time() - t0:.0f} s, loss {loss.item():.6f}, shapes {tuple(oc.shape)} {tuple(of.shape)} {tuple(om.shape)}", flush=True)
    d = {"losses": np.array([loss.item(), l_mid.item(), l_fin.item(), l_mae.item()], dtype=np.float64),
         "meta": np.array([1, 52, 0], dtype=np.int64), "gamma": np.array([GAMMA]), "frames": np.array([frames], dtype=np.int64)}
    for name, t in (("x_clip_align", oc), ("x_align", of), ("x_mae_align", om)):
        d[name + ":rows"], d[name + ":proj"] = digest(t)
    path = os.path.join(HERE, "student_6B_fulldepth_digest.npz" if frames == 4 else f"student_6B_fulldepth_{frames}f_digest.npz")
    np.savez_compressed(path, **d)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KB")


if __name__ == "__main__":
    main()
