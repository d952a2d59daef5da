"""Golden fixture for `sep_pos_embed=True` in the two reference models that the round-5 mirrors still refused (VERDICT r5 missing 4):
the distillation student DistInternVideo2 (InternVideo2/single_modality/models/internvideo2_distill.py:481-494, 551-563, 622-637, 677-692)
and the fine-tuning classifier InternVideo2 (models/internvideo2.py:390-397, 454-465, 510-525).

    python tests/golden/make_golden_sep_pos.py          (authoring container only: needs /root/reference)

RUNS THE REFERENCE'S OWN MODULES on CPU (unfused fp32 path, tests/golden/ref_loader.py) on the deterministic synthetic parameters / inputs of
oracle.internvideo2_oracle.  The separable tables have no oracle generator: their values (the reference's sincos initialisation plus a seeded
perturbation, so that every table -- the zero-initialised cls rows included -- has a non-trivial value and gradient) are stored as inputs
(`*:in:*`).  Stored: outputs, the loss of engines/engine_for_distill.py:107-110 / cross-entropy, the gradients of every separable table and of a
few trunk parameters.  -> tests/golden/sep_pos.npz"""
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

DIST_SEP = ["pos_embed_spatial", "pos_embed_temporal", "pos_embed_cls", "clip_pos_embed_spatial", "clip_pos_embed_temporal", "clip_pos_embed_cls"]
FT_SEP = ["pos_embed_spatial", "pos_embed_temporal", "pos_embed_cls"]
TRUNK = ["cls_token", "patch_embed.proj.bias", "blocks.0.norm1.weight", "blocks.1.ls2.gamma"]


def perturbed(m, params, keys, d, tag, seed):
    sd = m.state_dict()
    load = {k: v for k, v in params.items() if k in sd}
    rng = np.random.Generator(np.random.PCG64(seed))
    for k in keys:
        t = sd[k].clone() + torch.from_numpy(rng.standard_normal(tuple(sd[k].shape)).astype(np.float32)) * 0.02
        load[k] = t
        d[f"{tag}:in:{k}"] = t.numpy().copy()
    missing = set(sd) - set(load)
    assert not missing, missing
    m.load_state_dict(load, strict=True)


def main():
    assert ref_loader.available(), "needs the reference tree (IV_REFERENCE_ROOT)"
    d = {}
    # ---- distillation student
    cfg = O.named_config("dist64")
    params = O.synthetic_params(cfg, seed=2)
    video, mask, targets = O.synthetic_batch(cfg, 2, 4, seed=2)
    m = ref_loader.build_reference_distill(cfg, sep_pos_embed=True).train()
    assert "pos_embed" not in m.state_dict() and set(DIST_SEP) <= set(m.state_dict())
    perturbed(m, params, DIST_SEP, d, "dist", 78)
    oc, of = m(video, torch.from_numpy(mask))
    l_mid = (2 - 2 * (oc * targets[0]).sum(dim=-1)).mean()
    l_fin = (2 - 2 * (of * targets[1]).sum(dim=-1)).mean()
    loss = l_mid + l_fin
    loss.backward()
    named = dict(m.named_parameters())
    d["dist:x_clip_align"], d["dist:x_align"] = oc.detach().numpy(), of.detach().numpy()
    d["dist:losses"] = np.array([loss.item(), l_mid.item(), l_fin.item()])
    for k in DIST_SEP + TRUNK:
        d[f"dist:grad:{k}"] = named[k].grad.detach().numpy().copy()
    print(f"distill sep_pos_embed: loss {loss.item():.6f}")
    # ---- fine-tuning classifier
    cfg = O.named_config("tiny88")
    params = O.synthetic_finetune_params(cfg, 10, seed=12)
    video, _, _ = O.synthetic_batch(cfg, 2, 5, seed=12)
    labels = torch.tensor([3, 7])
    m = ref_loader.build_reference_finetune(cfg, 10, sep_pos_embed=True).train()
    assert "pos_embed" not in m.state_dict() and set(FT_SEP) <= set(m.state_dict())
    perturbed(m, params, FT_SEP, d, "ft", 79)
    logits = m(video)
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    named = dict(m.named_parameters())
    d["ft:logits"], d["ft:loss"] = logits.detach().numpy(), np.array([loss.item()])
    for k in FT_SEP + TRUNK + ["head.bias"]:
        d[f"ft:grad:{k}"] = named[k].grad.detach().numpy().copy()
    print(f"finetune sep_pos_embed: loss {loss.item():.6f}")
    path = os.path.join(HERE, "sep_pos.npz")
    np.savez_compressed(path, **d)
    print(f"wrote {path}: {len(d)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
