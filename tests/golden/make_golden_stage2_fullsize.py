"""Config-size golden digest for the stage-2 vision encoder: the REFERENCE's own multi_modality `PretrainInternVideo2`
(models/backbones/internvideo2/internvideo2.py, `pretrain_internvideo2_1b_patch14_224`: 40 x 1408, 4 x 224^2 frames, random mask 0.8 ->
L = 206; BASELINE configs[3], scripts/pretraining/stage2/1B/config.py:43-74), fp32 CPU, unfused path:

    python tests/golden/make_golden_stage2_fullsize.py      (authoring container only: needs /root/reference)

Inputs = the first two clips of what tests/test_fullsize_gpu.py::test_stage2_1B_vision_tower_and_vtc_loss_at_config_size feeds the oracle and
the HIP path (synthetic_params(seed 3); video / mask from PCG64(33), clips rounded to bf16 as the recipe feeds them).  Stored
(tests/golden/stage2_vision_1B_digest.npz): for x_vis, x_pool_vis, x_clip_align, x_align the first three rows in full and 16 fixed random
projections of every row.
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

B, CLIPS = 64, 2


def inputs():
    """the test's generator stream: all 64 clips are drawn, the first CLIPS are used"""
    n_keep = 1024 - int(1024 * 0.8)
    rng = np.random.Generator(np.random.PCG64(33))
    video = torch.from_numpy(rng.random((B, 3, 4, 224, 224), dtype=np.float32))
    mask = np.ones((B, 1024), dtype=bool)
    for b in range(B):
        mask[b, rng.permutation(1024)[:n_keep]] = False
    mask = np.concatenate([np.zeros((B, 1), dtype=bool), mask], axis=1)
    return video[:CLIPS].to(torch.bfloat16).float(), mask[:CLIPS]


def projection(C: int) -> np.ndarray:
    return np.random.Generator(np.random.PCG64(777 + C)).standard_normal((C, 16)).astype(np.float32) / np.sqrt(C).astype(np.float32)


def main():
    assert ref_loader.available(), "needs /root/reference"
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = O.StudentConfig(embed_dim=1408, depth=40, num_heads=16, mlp_ratio=48 / 11, num_frames=4, clip_return_layer=6, has_mae=False,
                          sep_image_video_pos_embed=True)
    params = O.synthetic_params(cfg, seed=3)
    m = ref_loader.build_reference_mm_vision(cfg)
    m.load_state_dict(params, strict=True)
    m.train()
    video, mask = inputs()
    with torch.no_grad():
        x_vis, x_pool, x_clip, x_align = m(video, torch.from_numpy(mask), False)
    out = {"meta": np.array([B, CLIPS, 33, 3], dtype=np.int64)}
    for name, t in (("x_vis", x_vis), ("x_pool_vis", x_pool), ("x_clip_align", x_clip), ("x_align", x_align)):
        rows = t.double().numpy().reshape(-1, t.shape[-1])
        out[name + ":rows"] = rows[:3].astype(np.float32)
        out[name + ":proj"] = (rows @ projection(rows.shape[1]).astype(np.float64)).astype(np.float32)
        out[name + ":shape"] = np.array(t.shape, dtype=np.int64)
    path = os.path.join(HERE, "stage2_vision_1B_digest.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", {k: tuple(v) for k, v in out.items() if k.endswith(":shape")})


if __name__ == "__main__":
    main()
