"""Real-geometry golden digest for the frozen VideoMAE teacher: the REFERENCE's own `VisionTransformer`
(single_modality/models/videomae.py:207-312) at VideoMAE-g's width and sequence geometry -- 1408 wide, 16 heads of 88, MLP 48/11, 16 frames
of 224^2 with 14-pixel patches and tubelet 2 (2048 tokens on the 8 x 16 x 16 sinusoid table) -- with the depth cut to 2,
fp32 CPU, flash_attn_func replaced by the stand-in of tests/golden/ref_loader.py that follows flash_attn's documented contract (so the
attention runs exactly as videomae.py:91-96 codes it):

    python tests/golden/make_golden_mae_teacher_fullwidth.py      (authoring container only: needs /root/reference)

Inputs: mae_teacher_params(seed 14), one clip from synthetic_mae_batch(seed 14).  Stored (tests/golden/mae_teacher_fullwidth_digest.npz):
the l2-normalised targets z of the full-sequence forward: first three rows in full + 16 fixed random projections of every row.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
from functools import partial

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_loader  # noqa: E402
from oracle import internvideo2_oracle as O  # noqa: E402


def config():
    return O.MaeConfig(img_size=224, patch_size=14, tubelet_size=2, num_frames=16, enc_dim=1408, enc_depth=2, enc_heads=16,
                       dec_dim=32, dec_depth=1, dec_heads=2, mlp_ratio=48 / 11, qkv_bias=True, init_values=0.0)


def projection(C: int) -> np.ndarray:
    return np.random.Generator(np.random.PCG64(777 + C)).standard_normal((C, 16)).astype(np.float32) / np.sqrt(C).astype(np.float32)


def main():
    assert ref_loader.available(), "needs /root/reference"
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = config()
    ref = ref_loader.load_sm_videomae_teacher()
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref.VisionTransformer(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.enc_dim, depth=cfg.enc_depth, num_heads=cfg.enc_heads,
                                  mlp_ratio=cfg.mlp_ratio, qkv_bias=True, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6),
                                  all_frames=cfg.num_frames, tubelet_size=cfg.tubelet_size, mae_return_layer=2)
    m.load_state_dict(O.mae_teacher_params(cfg, seed=14), strict=True)
    m.eval()
    video, _ = O.synthetic_mae_batch(cfg, 1, 1024, seed=14)
    with torch.no_grad():
        z = m(video)
    rows = z.double().numpy().reshape(-1, z.shape[-1])
    out = {"meta": np.array([14, 14], dtype=np.int64), "z:shape": np.array(z.shape, dtype=np.int64),
           "z:rows": rows[:3].astype(np.float32), "z:proj": (rows @ projection(rows.shape[1]).astype(np.float64)).astype(np.float32),
           "pos_embed:rows": m.pos_embed.detach().double().numpy().reshape(-1, cfg.enc_dim)[[0, 255, 2047]].astype(np.float32)}
    path = os.path.join(HERE, "mae_teacher_fullwidth_digest.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB z", tuple(z.shape))


if __name__ == "__main__":
    main()
