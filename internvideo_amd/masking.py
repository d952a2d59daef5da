"""Mask generators and teacher-target gathers of the InternVideo2 pre-training recipes (host-side index work + one HIP copy kernel).

Mirrors, with the same call signatures and the same consumption of numpy's GLOBAL RNG (so `np.random.seed(s)` reproduces the
reference's masks bit for bit):
  * single_modality/datasets/masking_generator.py:4-49   TubeMaskingGenerator / RandomMaskingGenerator (per-sample, DataLoader side)
  * multi_modality/models/mask.py:5-37                   TubeMaskingGenerator / RandomMaskingGenerator (batched, model side)
  * single_modality/engines/engine_for_pretraining.py:105-116 (distill :89-98, stage-2 visual :206-220): attention-guided mask from
    the CLIP teacher's pooled attention map by `torch.multinomial` without replacement
  * engine_for_pretraining.py:118-125: `norm_clip_middle[~mask].reshape(K,B,-1,C)` / `norm_mae[~mask[:,1:]]` teacher-target gathers.
Convention everywhere: True / 1 = masked, the cls column (index 0) is never masked.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import ops


class TubeMaskingGenerator:
    """SM datasets/masking_generator.py:4-26: one per-frame pattern tiled over the frames."""

    def __init__(self, input_size, mask_ratio):
        self.frames, self.height, self.width = input_size
        self.num_patches_per_frame = self.height * self.width
        self.total_patches = self.frames * self.num_patches_per_frame
        self.num_masks_per_frame = int(mask_ratio * self.num_patches_per_frame)
        self.total_masks = self.frames * self.num_masks_per_frame

    def __repr__(self):                                       # (sic: the reference's log line spells it this way, masking_generator.py:12-16)
        return f"Maks: total patches {self.total_patches}, mask patches {self.total_masks}"

    def __call__(self):
        mask_per_frame = np.hstack([np.zeros(self.num_patches_per_frame - self.num_masks_per_frame),
                                    np.ones(self.num_masks_per_frame)])
        np.random.shuffle(mask_per_frame)
        return np.tile(mask_per_frame, (self.frames, 1)).flatten()


class RandomMaskingGenerator:
    """SM datasets/masking_generator.py:29-49."""

    def __init__(self, input_size, mask_ratio):
        if not isinstance(input_size, tuple):
            input_size = (input_size,) * 3
        self.frames, self.height, self.width = input_size
        self.num_patches = self.frames * self.height * self.width
        self.num_mask = int(mask_ratio * self.num_patches)

    def __repr__(self):                                       # masking_generator.py:38-41
        return f"Maks: total patches {self.num_patches}, mask patches {self.num_mask}"

    def __call__(self):
        mask = np.hstack([np.zeros(self.num_patches - self.num_mask), np.ones(self.num_mask)])
        np.random.shuffle(mask)
        return mask


def tube_masks(input_size, mask_ratio, batch, device="cuda") -> torch.Tensor:
    """MM models/mask.py:5-19 `TubeMaskingGenerator(input_size, mask_ratio, batch, device)` -> bool (batch, T*H*W)."""
    g = TubeMaskingGenerator(tuple(input_size), mask_ratio)
    m = np.stack([g() for _ in range(batch)]).astype(bool)
    return torch.from_numpy(m).to(device, non_blocking=True)


def random_masks(input_size, mask_ratio, batch, device="cuda") -> torch.Tensor:
    """MM models/mask.py:22-37 `RandomMaskingGenerator(input_size, mask_ratio, batch, device)` -> bool (batch, T*H*W)."""
    g = RandomMaskingGenerator(tuple(input_size), mask_ratio)
    m = np.stack([g() for _ in range(batch)]).astype(bool)
    return torch.from_numpy(m).to(device, non_blocking=True)


def with_cls_column(mask: torch.Tensor) -> torch.Tensor:
    """`torch.cat((zeros(B,1), mask.flatten(1)), 1).to(bool)` (engine_for_pretraining.py:63-66): prepend the always-visible cls column."""
    m = mask.reshape(mask.shape[0], -1).to(torch.bool)
    return torch.cat([torch.zeros((m.shape[0], 1), dtype=torch.bool, device=m.device), m], dim=1)


def mask_from_importance(importance: torch.Tensor, B: int, mask_ratio: float) -> torch.Tensor:
    """engine_for_pretraining.py:106-116 after the multinomial draw: importance int64 (BT, N) = a sampled permutation per frame;
    the first N_vis = N - int(N * mask_ratio) entries stay visible.  -> bool (B, 1 + T*N) on importance's device."""
    BT, N = importance.shape
    n_vis = N - int(N * mask_ratio)
    m = torch.ones((BT, N), dtype=torch.bool, device=importance.device)
    m.scatter_(1, importance[:, :n_vis], False)
    return with_cls_column(m.view(B, -1))


def visible_tokens(attn_shape, B: int, mask_ratio: float) -> int:
    """kept tokens per clip (cls included) of the attention-guided mask, from the SHAPE of the teacher's map alone (no host sync): the map has
    one row per frame -- (B*T, H*W), InternVL teacher, engine_for_pretraining.py:105-116 -- or one per clip -- (B, T*H*W), InternVideo2
    teacher, engine_for_distill.py:89-98; every row keeps N - int(N * mask_ratio) of its N tokens."""
    rows, n = int(attn_shape[0]), int(attn_shape[1])
    if rows % B:
        raise ValueError(f"attention map with {rows} rows for {B} clips")
    return 1 + (rows // B) * (n - int(n * mask_ratio))


def attention_guided_mask(attn: torch.Tensor, B: int, mask_ratio: float, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """engine_for_pretraining.py:105-116: attn (BT, N) non-negative pooled attention of the CLIP teacher over one frame's patches -- or
    engine_for_distill.py:89-98: attn (B, T*N) over the whole clip (the InternVideo2 teacher; one draw per clip, frames keep different counts);
    `torch.multinomial(attn, N)` (without replacement) orders the patches, the first N_vis are kept.  The draw uses the RNG of
    attn's device exactly as the reference does, and stays on the device (no host round trip); like the reference's, it is not
    reproducible across back ends, so parity tests feed `mask_from_importance` a fixed draw."""
    importance = torch.multinomial(attn.float(), attn.shape[1], generator=generator)
    return mask_from_importance(importance, B, mask_ratio)


def gather_visible(t: torch.Tensor, mask: Optional[torch.Tensor] = None, vis_idx: Optional[torch.Tensor] = None,
                   drop_cls: bool = False) -> torch.Tensor:
    """`t[~mask].reshape(K, B, -1, C)` for t (K,B,1+N,C) / (B,1+N,C) and mask (B,1+N)   (engine_for_pretraining.py:118-121), or with
    drop_cls the MAE flavour `t[~mask[:, 1:]]` for t (K,B,N,C) (:123-125).  Bit-exact row copy by the HIP gather kernel; pass the
    `vis_idx` the student step already built (internvideo2_pretrain.build_gather_indices) to skip the compaction."""
    if vis_idx is None:
        from .internvideo2_pretrain import build_gather_indices
        vis_idx, _ = build_gather_indices(mask, t.device)
    return ops.gather_rows(t.contiguous(), vis_idx, skip=1 if drop_cls else 0)
