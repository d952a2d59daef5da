"""3-D sin-cos positional tables (InternVideo2/single_modality/models/pos_embed.py:9-131) -- host-side numpy, run once
at construction.  Layout per token: [temporal D/4 | spatial-h 3D/8 | spatial-w 3D/8], each block [sin | cos] of
pos * 10000^(-j / (block/2)); tokens ordered t-major then row-major (h, w); optional leading zero row for cls."""
from __future__ import annotations

import numpy as np


def _axis_table(dim: int, positions: np.ndarray) -> np.ndarray:
    if dim % 2:
        raise ValueError("sincos block width must be even")
    freq = np.arange(dim // 2, dtype=np.float32)
    freq /= dim / 2.0
    freq = 1.0 / 10000 ** freq
    ang = np.einsum("m,d->md", positions.reshape(-1), freq)
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def get_2d_sincos_pos_embed(embed_dim: int, grid_size: int, cls_token: bool = False) -> np.ndarray:
    ax = np.arange(grid_size, dtype=np.float32)
    first, second = np.meshgrid(ax, ax)           # meshgrid(w, h): the reference feeds the w coordinate to the first half
    tab = np.concatenate([_axis_table(embed_dim // 2, first), _axis_table(embed_dim // 2, second)], axis=1)
    if cls_token:
        tab = np.concatenate([np.zeros([1, embed_dim]), tab], axis=0)
    return tab


def get_1d_sincos_pos_embed(embed_dim: int, t_size: int, cls_token: bool = False) -> np.ndarray:
    tab = _axis_table(embed_dim, np.arange(t_size, dtype=np.float32))
    if cls_token:
        tab = np.concatenate([np.zeros([1, embed_dim]), tab], axis=0)
    return tab


def get_3d_sincos_pos_embed(embed_dim: int, grid_size: int, t_size: int, cls_token: bool = False) -> np.ndarray:
    if embed_dim % 4:
        raise ValueError("embed_dim must be a multiple of 4")
    d_t, d_s = embed_dim // 4, embed_dim // 4 * 3
    spatial = get_2d_sincos_pos_embed(d_s, grid_size)                        # (h*w, 3D/4)
    temporal = get_1d_sincos_pos_embed(d_t, t_size)                          # (t, D/4)
    n_sp = grid_size * grid_size
    tab = np.concatenate([np.repeat(temporal[:, None, :], n_sp, axis=1),
                          np.repeat(spatial[None, :, :], t_size, axis=0)], axis=-1).reshape(-1, embed_dim)
    if cls_token:
        tab = np.concatenate([np.zeros([1, embed_dim]), tab], axis=0)
    return tab


def _resize_table(table, n_extra: int, t_old: int, t_new: int, s_old: int, s_new: int):
    """(1, n_extra + t_old*s_old^2, D) -> (1, n_extra + t_new*s_new^2, D): linear along time, then bicubic (align_corners=False)
    over the (h, w) grid of every frame; the extra (cls) rows are carried over unchanged."""
    import torch
    import torch.nn.functional as F
    D = table.shape[-1]
    extra, grid = table[:, :n_extra], table[:, n_extra:]
    if t_old != t_new:
        g = grid.reshape(1, t_old, s_old * s_old, D).permute(0, 2, 3, 1).reshape(s_old * s_old, D, t_old)
        g = F.interpolate(g, size=t_new, mode='linear')
        grid = g.reshape(1, s_old * s_old, D, t_new).permute(0, 3, 1, 2).reshape(1, t_new * s_old * s_old, D)
    if s_old != s_new:
        g = grid.reshape(t_new, s_old, s_old, D).permute(0, 3, 1, 2)
        g = F.interpolate(g, size=(s_new, s_new), mode='bicubic', align_corners=False)
        grid = g.permute(0, 2, 3, 1).reshape(1, t_new * s_new * s_new, D)
    return torch.cat((extra, grid), dim=1)


def interpolate_pos_embed_internvideo2(checkpoint_model: dict, model, orig_t_size: int = 8) -> None:
    """Checkpoint-load-time resize of `pos_embed` / `clip_pos_embed` to the model's (frames, grid) -- what
    multi_modality/models/backbones/internvideo2/pos_embed.py:183-235 does before `load_state_dict` (in place on the dict).
    Host-side, runs once per checkpoint; separable tables are rejected like the reference (:237-238)."""
    if 'pos_embed_spatial' in checkpoint_model or 'pos_embed_temporal' in checkpoint_model:
        raise NotImplementedError
    num_patches = model.patch_embed.num_patches
    n_extra = model.pos_embed.shape[-2] - num_patches
    t_new = model.num_frames // model.tubelet_size if hasattr(model, "num_frames") else model.patch_embed.grid_size[0]
    s_new = int((num_patches // t_new) ** 0.5)
    for name in ('pos_embed', 'clip_pos_embed'):
        if name in checkpoint_model:
            tab = checkpoint_model[name]
            s_old = int(((tab.shape[-2] - n_extra) // orig_t_size) ** 0.5)
            if orig_t_size != t_new or s_old != s_new:
                checkpoint_model[name] = _resize_table(tab, n_extra, orig_t_size, t_new, s_old, s_new)


def interpolate_pos_embed(checkpoint_model: dict, model, orig_t_size: int = 4, pos_name: str = 'vision_encoder.pos_embed') -> None:
    """The single-table form (multi_modality/models/backbones/internvideo2/pos_embed.py:137-182): resize `checkpoint_model[pos_name]` to the
    model's (model.T frames, grid); a missing key is left alone (the reference's `if pos_name in checkpoint_model`)."""
    if pos_name not in checkpoint_model:
        return
    num_patches = model.patch_embed.num_patches
    n_extra = model.pos_embed.shape[-2] - num_patches
    t_new = model.T
    s_new = int((num_patches // t_new) ** 0.5)
    tab = checkpoint_model[pos_name]
    s_old = int(((tab.shape[-2] - n_extra) // orig_t_size) ** 0.5)
    if orig_t_size != t_new or s_old != s_new:
        checkpoint_model[pos_name] = _resize_table(tab, n_extra, orig_t_size, t_new, s_old, s_new)


def interpolate_pos_embed_internvideo2_new(checkpoint_model: dict, model, orig_t_size: int = 8) -> None:
    """The key-scanning form (pos_embed.py:239-298): every key that contains 'pos_embed' (so 'clip_pos_embed' and prefixed names such as
    'vision_encoder.pos_embed' too) except the image tables ('img_pos_embed') is resized; no such key is an error (the reference asserts),
    separable tables are rejected."""
    names = [k for k in checkpoint_model.keys() if 'pos_embed' in k and 'img_pos_embed' not in k]
    assert len(names) > 0, list(checkpoint_model.keys())
    if 'pos_embed_spatial' in checkpoint_model or 'pos_embed_temporal' in checkpoint_model:
        raise NotImplementedError
    num_patches = model.patch_embed.num_patches
    n_extra = model.pos_embed.shape[-2] - num_patches
    t_new = model.num_frames // model.tubelet_size
    s_new = int((num_patches // t_new) ** 0.5)
    for name in names:
        tab = checkpoint_model[name]
        s_old = int(((tab.shape[-2] - n_extra) // orig_t_size) ** 0.5)
        if orig_t_size != t_new or s_old != s_new:
            checkpoint_model[name] = _resize_table(tab, n_extra, orig_t_size, t_new, s_old, s_new)
