"""Golden fixture for FRAME-LEVEL (3-D) contrastive features: `get_sim(vision_proj [B,L,C], text_proj [B,C], temp, agg_method)` and
`VTC_VTM_Loss.vtc_loss(..., agg_method=)` of InternVideo2/multi_modality/models/criterions.py:15-103 -- the branches :31-39 (vision carries the
frame axis) and :40-48 (text carries it), agg_method "mean" and "max" -- refused by the round-5 mirror (VERDICT r5 missing 5).

    python tests/golden/make_golden_vtc3d.py          (authoring container only: needs /root/reference)

RUNS THE REFERENCE'S OWN criterions.py on CPU (fp32) on seeded inputs; stores the inputs' seeds' products: similarities, losses (with and without
idx) and the gradients of both feature tensors.  -> tests/golden/vtc3d.npz"""
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


def main():
    assert ref_loader.available(), "needs the reference tree (IV_REFERENCE_ROOT)"
    get_sim, VTC = ref_loader.load_mm_criterions_functions()
    rng = np.random.Generator(np.random.PCG64(31))
    B, L, C = 12, 4, 512
    d = {"meta": np.array([B, L, C], dtype=np.int64)}
    d["v3"] = rng.standard_normal((B, L, C)).astype(np.float32)
    d["t2"] = rng.standard_normal((B, C)).astype(np.float32)
    d["t3"] = rng.standard_normal((B, L, C)).astype(np.float32)
    d["v2"] = rng.standard_normal((B, C)).astype(np.float32)
    d["idx"] = np.array([0, 1, 2, 3, 3, 5, 6, 7, 1, 9, 10, 0], dtype=np.int64)
    temp = torch.tensor(0.07)
    crit = VTC(False)
    for tag, vk, tk in (("vis", "v3", "t2"), ("txt", "v2", "t3")):
        for agg in ("mean", "max"):
            v = torch.from_numpy(d[vk]).clone().requires_grad_(True)
            t = torch.from_numpy(d[tk]).clone().requires_grad_(True)
            s1, s2 = get_sim(v, t, temp, agg_method=agg)
            loss = crit.vtc_loss(v, t, torch.from_numpy(d["idx"]), temp, all_gather=False, agg_method=agg)
            loss.backward()
            loss_noidx = crit.vtc_loss(v.detach(), t.detach(), None, temp, all_gather=False, agg_method=agg)
            k = f"{tag}:{agg}:"
            d[k + "sim_v2t"], d[k + "sim_t2v"] = s1.detach().numpy(), s2.detach().numpy()
            d[k + "loss"] = np.array([loss.item(), loss_noidx.item()])
            d[k + "grad_v"], d[k + "grad_t"] = v.grad.numpy().copy(), t.grad.numpy().copy()
            print(f"{k} sim {tuple(s1.shape)} / {tuple(s2.shape)} loss {loss.item():.6f} (no idx {loss_noidx.item():.6f})")
    path = os.path.join(HERE, "vtc3d.npz")
    np.savez_compressed(path, **d)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
