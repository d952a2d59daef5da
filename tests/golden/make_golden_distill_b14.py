"""Config-size golden digest for the distillation student: the REFERENCE's own `DistInternVideo2`
(single_modality/models/internvideo2_distill.py) at distill_internvideo2_base_patch14_224's geometry (ViT-B/14, 8 x 224^2, global mask 0.8 ->
411 visible tokens, six 1408-wide CLIP decoders), fp32 CPU forward + the two distillation losses of engine_for_distill.py:107-121 + backward:

    python tests/golden/make_golden_distill_b14.py      (authoring container only: needs /root/reference)

Inputs = what tests/test_fullsize_gpu.py::test_distill_B14_forward_and_backward_match_oracle feeds the oracle and the HIP path
(synthetic_params(seed 0); video / mask / targets from PCG64(21)).  Stored (tests/golden/distill_B14_digest.npz): x_clip_align and x_align
(first rows + 16 fixed random projections of every row), the loss, corners / norms of sampled gradients.
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

MATS = ["blocks.0.attn.qkv.weight", "blocks.11.mlp.fc2.weight", "patch_embed.proj.weight", "clip_decoder.0.head.weight", "clip_decoder.5.head.weight",
        "final_clip_decoder.head.weight", "clip_projector.cross_attn.v.weight"]


def config():
    return O.StudentConfig(embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, num_frames=8, clip_teacher_embed_dim=1408, clip_return_layer=6,
                           has_mae=False)


def inputs():
    rng = np.random.Generator(np.random.PCG64(21))
    video = torch.from_numpy(rng.random((1, 3, 8, 224, 224), dtype=np.float32))
    mask = np.ones((1, 2048), dtype=bool)
    mask[0, rng.permutation(2048)[:410]] = False
    mask = np.concatenate([np.zeros((1, 1), dtype=bool), mask], axis=1)
    unit = lambda shape: torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal(shape).astype(np.float32)), dim=-1)   # noqa: E731
    tc, tf = unit((6, 1, 411, 1408)), unit((1, 768))
    return video, mask, tc, tf


def projection(C: int) -> np.ndarray:
    return np.random.Generator(np.random.PCG64(777 + C)).standard_normal((C, 16)).astype(np.float32) / np.sqrt(C).astype(np.float32)


def main():
    assert ref_loader.available(), "needs /root/reference"
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = config()
    params = O.synthetic_params(cfg, seed=0)
    video, mask, tc, tf = inputs()
    m = ref_loader.build_reference_distill(cfg)
    m.load_state_dict(params, strict=True)
    m.train()
    oc, of = m(video, torch.from_numpy(mask))
    loss = (2 - 2 * (oc * tc).sum(-1)).mean() + (2 - 2 * (of * tf).sum(-1)).mean()
    loss.backward()
    d = {"loss": np.array([loss.item()], dtype=np.float64)}
    for name, t in (("x_clip_align", oc), ("x_align", of)):
        rows = t.detach().double().numpy().reshape(-1, t.shape[-1])
        d[name + ":rows"] = rows[:3].astype(np.float32)
        d[name + ":proj"] = (rows @ projection(rows.shape[1]).astype(np.float64)).astype(np.float32)
        d[name + ":shape"] = np.array(t.shape, dtype=np.int64)
    sd = dict(m.named_parameters())
    for k in MATS:
        g = sd[k].grad.detach()
        g2 = g.reshape(g.shape[0], -1)
        d["grad:" + k + ":corner"] = g2[:16, :16].numpy().copy()
        d["grad:" + k + ":norm"] = np.array([g.double().norm().item()], dtype=np.float64)
    path = os.path.join(HERE, "distill_B14_digest.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB loss", loss.item(), tuple(oc.shape), tuple(of.shape))


if __name__ == "__main__":
    main()
