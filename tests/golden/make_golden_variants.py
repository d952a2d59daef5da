"""Golden fixtures for the two constructor variants no shipped recipe uses but the reference's PretrainInternVideo2 accepts (B1 contract):
`sep_pos_embed=True` (P:479-495, 639-655, 696-712, 726-734) and `clip_norm_type = mae_norm_type = 'none'` (P:358-363, 396-401).

    python tests/golden/make_golden_variants.py          (authoring container only: needs /root/reference)

RUNS THE REFERENCE'S OWN MODULE on CPU (unfused fp32 path, tests/golden/ref_loader.py) on the deterministic synthetic parameters / inputs of
oracle.internvideo2_oracle and stores outputs, the distillation loss of engines/engine_for_pretraining.py:131-148 and gradients in
tests/golden/variants.npz.  The separable tables have no oracle generator: their values (the reference's sincos initialisation plus a seeded
perturbation, so that every table has a non-trivial gradient) are stored in the fixture as inputs (`in:*`)."""
from __future__ import annotations

import contextlib
import io
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

SEP = ["pos_embed_spatial", "pos_embed_temporal", "pos_embed_cls", "clip_pos_embed_spatial", "clip_pos_embed_temporal", "clip_pos_embed_cls",
       "mae_pos_embed_spatial", "mae_pos_embed_temporal"]
GRADS = ["cls_token", "patch_embed.proj.bias", "blocks.0.norm1.weight", "blocks.1.ls2.gamma", "blocks.2.mlp.fc2.bias", "clip_decoder.0.norm.weight",
         "clip_decoder.0.head.bias", "mae_decoder.0.norm.bias", "mae_decoder.1.head.0.bias", "final_clip_decoder.head.bias",
         "final_clip_decoder.norm.weight"]


def build(cfg, drop_path_rate=0.0, **kw):
    ref = ref_loader.load_sm_pretrain()
    with contextlib.redirect_stdout(io.StringIO()):
        return ref.PretrainInternVideo2(
            in_chans=cfg.in_chans, patch_size=cfg.patch_size, img_size=cfg.img_size, qkv_bias=False, drop_path_rate=drop_path_rate, embed_dim=cfg.embed_dim,
            num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, init_values=1e-5, qk_normalization=True, depth=cfg.depth, use_flash_attn=False,
            use_fused_rmsnorm=False, use_fused_mlp=False, attn_pool_num_heads=cfg.attn_pool_num_heads, clip_embed_dim=cfg.clip_embed_dim,
            num_frames=cfg.num_frames, tubelet_size=cfg.tubelet_size, clip_teacher_embed_dim=cfg.clip_teacher_embed_dim,
            clip_teacher_final_dim=cfg.clip_teacher_final_dim, clip_return_layer=cfg.clip_return_layer,
            clip_student_return_interval=cfg.clip_student_return_interval, mae_teacher_embed_dim=cfg.mae_teacher_embed_dim,
            mae_return_layer=cfg.mae_return_layer, mae_student_return_interval=cfg.mae_student_return_interval, **kw)


def run(tag, d, cfg, params, video, mask, targets, **kw):
    m = build(cfg, **kw).train()
    sd = m.state_dict()
    load = {k: v for k, v in params.items() if k in sd}
    if kw.get("sep_pos_embed"):
        rng = np.random.Generator(np.random.PCG64(77))
        for k in SEP:
            t = sd[k].clone() + torch.from_numpy(rng.standard_normal(tuple(sd[k].shape)).astype(np.float32)) * 0.02
            load[k] = t
            d[f"{tag}:in:{k}"] = t.numpy().copy()
    m.load_state_dict(load, strict=True)
    oc, of, om = m(video, torch.from_numpy(mask))
    tc, tf, tm = targets
    if kw.get("clip_norm_type") == "none":          # un-normalised student features: the loss formula of the engine is unchanged
        pass
    loss = (2 - 2 * (oc * tc).sum(-1)).mean() + (2 - 2 * (of * tf).sum(-1)).mean() + (2 - 2 * (om * tm).sum(-1)).mean()
    loss.backward()
    named = dict(m.named_parameters())
    d[f"{tag}:x_clip_align"], d[f"{tag}:x_align"], d[f"{tag}:x_mae_align"] = (t.detach().numpy() for t in (oc, of, om))
    d[f"{tag}:loss"] = np.array([loss.item()], dtype=np.float64)
    for k in GRADS + (SEP if kw.get("sep_pos_embed") else ["pos_embed", "clip_pos_embed", "mae_pos_embed"]):
        g = named[k].grad
        if g is not None:
            d[f"{tag}:grad:{k}"] = g.detach().numpy().copy()
    print(f"{tag}: loss {loss.item():.6f}, |x_clip| mean row norm {oc.norm(dim=-1).mean().item():.3f}")


def run_droppath(d, cfg, params, video, mask, targets, rate=0.3, seed=123):
    """the reference with drop_path_rate > 0 in train mode (what the recipes run: 0.25 / 0.3): timm's DropPath draws one uniform per sample
    and call; a recording subclass keeps them (call order: block i drop_path1, drop_path2; block 0 has rate 0 -> nn.Identity, no draw) so that
    the MI355X model can be fed the SAME draws (`model._dp_uniform`)."""
    ref = ref_loader.load_sm_pretrain()
    base = ref.DropPath
    draws = []

    class Recording(base):
        def forward(self, x):
            if not self.drop_prob or not self.training:
                return x
            keep = 1 - self.drop_prob
            u = torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype, device=x.device)
            draws.append((len(draws), u.reshape(-1).clone()))
            return x.div(keep) * (keep + u).floor_()

    ref.DropPath = Recording
    try:
        m = build(cfg, drop_path_rate=rate).train()
    finally:
        ref.DropPath = base
    m.load_state_dict(params, strict=True)
    torch.manual_seed(seed)
    oc, of, om = m(video, torch.from_numpy(mask))
    tc, tf, tm = targets
    loss = (2 - 2 * (oc * tc).sum(-1)).mean() + (2 - 2 * (of * tf).sum(-1)).mean() + (2 - 2 * (om * tm).sum(-1)).mean()
    loss.backward()
    B = video.shape[0]
    U = np.full((cfg.depth, 2, B), 0.999, dtype=np.float32)            # blocks without a DropPath (rate 0): any draw keeps the sample
    live = [i for i in range(cfg.depth) if float(torch.linspace(0, rate, cfg.depth)[i]) > 0]
    assert len(draws) == 2 * len(live), (len(draws), live)
    for n, (_, u) in enumerate(draws):
        U[live[n // 2], n % 2] = u.numpy()
    named = dict(m.named_parameters())
    d["dp:uniform"] = U
    d["dp:rate"] = np.array([rate], dtype=np.float64)
    d["dp:x_clip_align"], d["dp:x_align"], d["dp:x_mae_align"] = (t.detach().numpy() for t in (oc, of, om))
    d["dp:loss"] = np.array([loss.item()], dtype=np.float64)
    for k in GRADS + ["pos_embed", "clip_pos_embed", "mae_pos_embed"]:
        if named[k].grad is not None:
            d[f"dp:grad:{k}"] = named[k].grad.detach().numpy().copy()
    dropped = int((np.floor((1 - np.linspace(0, rate, cfg.depth))[:, None, None] + U) == 0).sum())
    print(f"dp: loss {loss.item():.6f}, {len(draws)} DropPath calls, {dropped} dropped (block, branch, sample) triples")


def main():
    assert ref_loader.available(), "needs the reference tree (IV_REFERENCE_ROOT)"
    cfg = O.named_config("tiny64")
    B, n_vis, seed = 2, 5, 11
    params = O.synthetic_params(cfg, seed=seed)
    video, mask, targets = O.synthetic_batch(cfg, B, n_vis, seed=seed)
    d = {"meta": np.array([B, n_vis, seed], dtype=np.int64)}
    run("sep", d, cfg, params, video, mask, targets, sep_pos_embed=True, clip_norm_type="l2", mae_norm_type="l2")
    run("none", d, cfg, params, video, mask, targets, sep_pos_embed=False, clip_norm_type="none", mae_norm_type="none")
    B4 = 4                                                              # a few more samples so that some branches really drop
    video4, mask4, targets4 = O.synthetic_batch(cfg, B4, n_vis, seed=seed + 1)
    d["dp:meta"] = np.array([B4, n_vis, seed + 1], dtype=np.int64)
    run_droppath(d, cfg, params, video4, mask4, targets4)
    path = os.path.join(HERE, "variants.npz")
    np.savez_compressed(path, **d)
    print(f"wrote {path}: {len(d)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
