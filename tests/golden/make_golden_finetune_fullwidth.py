"""Real-width, full-sequence golden digest for the fine-tuning classifier: the REFERENCE's own `InternVideo2`
(single_modality/models/internvideo2.py) at the 1B model's width and sequence length -- 1408 wide, 16 heads of 88, MLP 48/11, 8 frames of
224^2 with NO masking (L = 2049 tokens: 33 key tiles per attention head where the pre-training step has 7), 16 pooling heads, 400 classes --
with the depth cut to 4, fp32 CPU forward + cross-entropy + backward:

    python tests/golden/make_golden_finetune_fullwidth.py      (authoring container only: needs /root/reference)

Inputs: synthetic_finetune_params(400 classes, seed 18), one clip of synthetic_batch(seed 18), label 123.  Stored
(tests/golden/finetune_fullwidth_digest.npz): the 400 logits, the loss, corners / norms of sampled gradients.
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

CLASSES, SEED, LABEL = 400, 18, 123
MATS = ["head.weight", "pos_embed", "blocks.0.attn.qkv.weight", "blocks.3.mlp.fc2.weight", "patch_embed.proj.weight", "clip_projector.cross_attn.k.weight"]


def config():
    return O.StudentConfig(embed_dim=1408, depth=4, num_heads=16, mlp_ratio=48 / 11, num_frames=8, attn_pool_num_heads=16, clip_embed_dim=768)


def main():
    assert ref_loader.available(), "needs /root/reference"
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = config()
    params = O.synthetic_finetune_params(cfg, CLASSES, seed=SEED)
    video, _, _ = O.synthetic_batch(cfg, 1, 52, seed=SEED)
    m = ref_loader.build_reference_finetune(cfg, CLASSES)
    m.load_state_dict(params, strict=True)
    m.train()
    logits = m(video)
    loss = torch.nn.functional.cross_entropy(logits.float(), torch.tensor([LABEL]))
    loss.backward()
    d = {"meta": np.array([CLASSES, SEED, LABEL], dtype=np.int64), "logits": logits.detach().numpy().astype(np.float32),
         "loss": np.array([loss.item()], dtype=np.float64)}
    sd = dict(m.named_parameters())
    for k in MATS:
        g = sd[k].grad.detach()
        g2 = g.reshape(-1, g.shape[-1])
        d["grad:" + k + ":corner"] = g2[:16, :16].numpy().copy()
        d["grad:" + k + ":norm"] = np.array([g.double().norm().item()], dtype=np.float64)
    path = os.path.join(HERE, "finetune_fullwidth_digest.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB logits", tuple(logits.shape), "loss", loss.item())


if __name__ == "__main__":
    main()
