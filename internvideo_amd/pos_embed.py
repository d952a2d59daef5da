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
