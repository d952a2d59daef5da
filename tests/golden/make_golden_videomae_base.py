"""Real-geometry golden digest for the VideoMAE pixel path: the REFERENCE's own
`pretrain_mae_base_patch16_224` geometry (InternVideo1/Pretrain/VideoMAE/modeling_pretrain.py:416-434: ViT-B/16 encoder, 4-block 384-wide
decoder, 16 frames of 224^2, tubelet 2 -> 1568 tokens, mask ratio 0.9 -> 157 visible) with the engine's labels and MSE
(engine_for_pretraining.py:53-106), fp32 CPU forward + backward:

    python tests/golden/make_golden_videomae_base.py      (authoring container only: needs /root/reference)

Inputs: synthetic_mae_params(MaeConfig(), seed 16), synthetic_mae_batch(B = 1, 1411 masked tokens, seed 16).  Stored
(tests/golden/videomae_base_digest.npz): the predictions (first three rows + 16 fixed random projections of every row), the loss, corners /
norms of sampled gradients.
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

B, N_MASK, SEED = 1, 1411, 16
MATS = ["encoder.blocks.0.attn.qkv.weight", "encoder.blocks.11.mlp.fc2.weight", "encoder_to_decoder.weight", "decoder.head.weight",
        "encoder.patch_embed.proj.weight", "decoder.blocks.3.attn.proj.weight"]


def projection(C: int) -> np.ndarray:
    return np.random.Generator(np.random.PCG64(777 + C)).standard_normal((C, 16)).astype(np.float32) / np.sqrt(C).astype(np.float32)


def labels_of(video, mm, cfg):
    from einops import rearrange
    mean = torch.as_tensor((0.485, 0.456, 0.406))[None, :, None, None, None]
    std = torch.as_tensor((0.229, 0.224, 0.225))[None, :, None, None, None]
    unnorm = video * std + mean                                                              # ME:66-74
    sq = rearrange(unnorm, 'b c (t p0) (h p1) (w p2) -> b (t h w) (p0 p1 p2) c', p0=cfg.tubelet_size, p1=cfg.patch_size, p2=cfg.patch_size)
    nrm = (sq - sq.mean(dim=-2, keepdim=True)) / (sq.var(dim=-2, unbiased=True, keepdim=True).sqrt() + 1e-6)
    patch = rearrange(nrm, 'b n p c -> b n (p c)')
    return patch[mm].reshape(video.shape[0], -1, patch.shape[-1])                              # ME:95-98


def main():
    assert ref_loader.available(), "needs /root/reference"
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = O.named_mae_config("mae_base")
    params = O.synthetic_mae_params(cfg, seed=SEED)
    video, mask = O.synthetic_mae_batch(cfg, B, N_MASK, seed=SEED)
    mm = torch.from_numpy(mask)
    labels = labels_of(video, mm, cfg)
    m = ref_loader.build_reference_videomae(cfg)
    m.load_state_dict(params, strict=True)
    m.train()
    m.decoder.with_fp16 = False                       # CPU run: no cuda autocast region
    out = m(video, mm)
    loss = torch.nn.MSELoss()(input=out.float(), target=labels)                                # ME:53,101-106
    loss.backward()
    rows = out.detach().double().numpy().reshape(-1, out.shape[-1])
    d = {"meta": np.array([B, N_MASK, SEED], dtype=np.int64), "out:shape": np.array(out.shape, dtype=np.int64),
         "out:rows": rows[:3].astype(np.float32), "out:proj": (rows @ projection(rows.shape[1]).astype(np.float64)).astype(np.float32),
         "loss": np.array([loss.item()], dtype=np.float64)}
    sd = dict(m.named_parameters())
    for k in MATS:
        g = sd[k].grad.detach()
        g2 = g.reshape(g.shape[0], -1)
        d["grad:" + k + ":corner"] = g2[:16, :16].numpy().copy()
        d["grad:" + k + ":norm"] = np.array([g.double().norm().item()], dtype=np.float64)
    path = os.path.join(HERE, "videomae_base_digest.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB out", tuple(out.shape), "loss", loss.item())


if __name__ == "__main__":
    main()
