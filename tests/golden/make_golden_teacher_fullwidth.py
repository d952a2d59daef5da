"""Real-width golden digest for the frozen CLIP teacher: the REFERENCE's own `InternVL_CLIP`
(single_modality/models/internvl_clip_vision.py) at InternVL-6B's width and sequence geometry -- 3200 wide, 25 heads of 128, MLP 12800,
224^2 frames of 16 x 16 patches (257-token per-frame sequences), 8 frames, 16 pooling heads, clip_embed_dim 768 -- with the depth cut to 2
(the 48-block model is 24 GB of fp32 parameters), fp32 CPU, unfused path:

    python tests/golden/make_golden_teacher_fullwidth.py      (authoring container only: needs /root/reference)

Inputs: synthetic_teacher_params(seed 12), one clip from PCG64(120).  Stored (tests/golden/clip_teacher_fullwidth_digest.npz): for the tapped
targets z, the pooled feature x and the pooled-attention map: first three rows in full + 16 fixed random projections of every row.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_loader  # noqa: E402
from oracle import internvideo2_oracle as O  # noqa: E402


def config():
    return O.StudentConfig(img_size=224, embed_dim=3200, depth=2, num_heads=25, mlp_ratio=4.0, num_frames=8, attn_pool_num_heads=16,
                           clip_embed_dim=768, clip_return_layer=2, has_mae=False)


def inputs(cfg):
    rng = np.random.Generator(np.random.PCG64(120))
    return torch.from_numpy(rng.random((1, cfg.in_chans, cfg.num_frames, cfg.img_size, cfg.img_size), dtype=np.float32))


def projection(C: int) -> np.ndarray:
    return np.random.Generator(np.random.PCG64(777 + C)).standard_normal((C, 16)).astype(np.float32) / np.sqrt(C).astype(np.float32)


def main():
    assert ref_loader.available(), "needs /root/reference"
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = config()
    p = O.synthetic_teacher_params(cfg, seed=12)
    m = ref_loader.build_reference_clip_teacher(cfg)
    m.load_state_dict(p, strict=True)
    m.eval()
    video = inputs(cfg)
    with torch.no_grad():
        z, x, attn = m(video)
    out = {"meta": np.array([12, 120], dtype=np.int64)}
    for name, t in (("z", z), ("x", x), ("attn", attn)):
        rows = t.double().numpy().reshape(-1, t.shape[-1])
        out[name + ":rows"] = rows[:3].astype(np.float32)
        out[name + ":proj"] = (rows @ projection(rows.shape[1]).astype(np.float64)).astype(np.float32)
        out[name + ":shape"] = np.array(t.shape, dtype=np.int64)
    path = os.path.join(HERE, "clip_teacher_fullwidth_digest.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", {k: tuple(int(i) for i in v) for k, v in out.items() if k.endswith(":shape")})


if __name__ == "__main__":
    main()
