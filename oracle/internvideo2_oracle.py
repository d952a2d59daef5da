"""CPU ORACLE for the InternVideo2 masked video-ViT hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch, functional, fp32/fp64 CPU restatement of the algorithm the
reference implements in
  /root/reference/InternVideo2/single_modality/models/internvideo2_pretrain.py   ("P:")
  /root/reference/InternVideo2/single_modality/models/pos_embed.py               ("PE:")
  /root/reference/InternVideo2/single_modality/engines/engine_for_pretraining.py ("E:")
  /root/reference/InternVideo2/single_modality/datasets/masking_generator.py     ("MG:")
  /root/reference/InternVideo2/multi_modality/models/criterions.py               ("C:")
  /root/reference/InternVideo2/single_modality/models/internvideo2_distill.py    ("D:")
  /root/reference/InternVideo2/multi_modality/models/backbones/internvideo2/internvideo2.py ("V:")
  /root/reference/InternVideo2/multi_modality/models/mask.py                     ("MK:")
  /root/reference/InternVideo2/single_modality/models/internvl_clip_vision.py    ("T:")
  /root/reference/InternVideo2/single_modality/models/internvideo2.py            ("F:")
  /root/reference/InternVideo2/single_modality/models/videomae.py                ("VT:")
  /root/reference/InternVideo1/Pretrain/VideoMAE/modeling_pretrain.py, modeling_finetune.py, engine_for_pretraining.py ("MP:", "MF:", "ME:")
  /root/reference/InternVideo2/multi_modality/models/utils.py                    ("U:")
Every function cites the reference file:line it follows.

Rules (see DESIGN.md "oracle"):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this;
    the product package `internvideo_amd` never does.
  * parity pinning: the reference ships NO golden vectors for this path, so
    the oracle is pinned against outputs of the reference's own modules executed on CPU in the
    authoring container (tests/golden/make_golden.py -> tests/golden/*.npz, committed) and
    checked by tests/test_oracle_golden.py.
  * parameters are passed as a flat dict keyed by the reference's state_dict names, so the
    reference's state_dict, the product's state_dict and a synthetic dict are interchangeable.

Floating point work is plain torch on CPU (fp32 by default, fp64 on request); index / mask
work is numpy integer arithmetic and is compared bit-exactly.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class StudentConfig:
    """Constructor surface of PretrainInternVideo2 (P:406-443) that changes the arithmetic."""
    img_size: int = 224
    patch_size: int = 14
    in_chans: int = 3
    embed_dim: int = 1408
    depth: int = 40
    num_heads: int = 16
    mlp_ratio: float = 48 / 11
    num_frames: int = 8
    tubelet_size: int = 1
    attn_pool_num_heads: int = 16
    clip_embed_dim: int = 768
    clip_teacher_embed_dim: int = 3200
    clip_teacher_final_dim: int = 768
    clip_return_layer: int = 1
    clip_student_return_interval: int = 1
    mae_teacher_embed_dim: int = 1408
    mae_return_layer: int = 1
    mae_student_return_interval: int = 1
    layerscale_force_fp32: bool = True      # P:261-262 (numerically irrelevant in fp32)
    gelu: str = "erf"                       # "erf" = unfused Mlp (P:224); "tanh" = flash_attn FusedMLP
    rms_eps: float = 1e-6                   # P:467-469
    ln_eps: float = 1e-5                    # P:525,532,541,550
    # --- flavours other than the SM pre-training student (encoder_forward below) ---
    has_mae: bool = True                    # False: DistInternVideo2 (D:417-697) / stage-2 vision encoder (V:381-685)
    clip_decoder_kind: str = "linear"       # D:412-414 `clip_student_decoder`: "linear" | "mlp"
    clip_return_index_override: Optional[Tuple[int, ...]] = None      # D:462-466 `clip_student_return_index`
    sep_image_video_pos_embed: bool = False  # V:449-459

    @property
    def grid(self) -> Tuple[int, int, int]:
        g = self.img_size // self.patch_size
        return (self.num_frames // self.tubelet_size, g, g)          # P:313-317

    @property
    def num_patches(self) -> int:
        t, h, w = self.grid
        return t * h * w

    @property
    def mlp_hidden(self) -> int:
        return int(self.embed_dim * self.mlp_ratio)                  # P:267

    @property
    def clip_return_index(self) -> List[int]:
        if self.clip_return_index_override:
            return list(self.clip_return_index_override)
        return [self.depth - int(i * self.clip_student_return_interval) - 1
                for i in range(self.clip_return_layer)]              # P:453-455

    @property
    def mae_return_index(self) -> List[int]:
        if not self.has_mae:
            return []
        return [self.depth - int(i * self.mae_student_return_interval) - 1
                for i in range(self.mae_return_layer)]               # P:460-462


# --------------------------------------------------------------------------------------
# positional embedding (float64 numpy, as the reference)            PE:9-131
# --------------------------------------------------------------------------------------
def sincos_1d(embed_dim: int, pos: np.ndarray) -> np.ndarray:
    """PE:113-131.  out[m] = [sin(pos*omega) | cos(pos*omega)], omega_j = 10000^(-j/(D/2))."""
    assert embed_dim % 2 == 0
    omega = np.arange(embed_dim // 2, dtype=np.float32)
    omega /= embed_dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def sincos_2d_from_grid(embed_dim: int, grid: np.ndarray) -> np.ndarray:
    """PE:98-110: first half of the dims encodes grid[0], second half grid[1]."""
    assert embed_dim % 2 == 0
    return np.concatenate([sincos_1d(embed_dim // 2, grid[0]),
                           sincos_1d(embed_dim // 2, grid[1])], axis=1)


def sincos_pos_embed_3d(embed_dim: int, grid_size: int, t_size: int, cls_token: bool = False) -> np.ndarray:
    """PE:9-54.  [temporal D/4 | spatial 3D/4]; meshgrid(grid_w, grid_h) ("w goes first");
    token order t-major then row-major (h, w); optional leading all-zero cls row."""
    assert embed_dim % 4 == 0
    d_sp, d_t = embed_dim // 4 * 3, embed_dim // 4
    gh = np.arange(grid_size, dtype=np.float32)
    gw = np.arange(grid_size, dtype=np.float32)
    grid = np.stack(np.meshgrid(gw, gh), axis=0).reshape([2, 1, grid_size, grid_size])
    sp = sincos_2d_from_grid(d_sp, grid)                              # (H*W, 3D/4)
    tp = sincos_1d(d_t, np.arange(t_size, dtype=np.float32))          # (T, D/4)
    tp = np.repeat(tp[:, None, :], grid_size ** 2, axis=1)            # (T, HW, D/4)
    sp = np.repeat(sp[None, :, :], t_size, axis=0)                    # (T, HW, 3D/4)
    pe = np.concatenate([tp, sp], axis=-1).reshape([-1, embed_dim])
    if cls_token:
        pe = np.concatenate([np.zeros([1, embed_dim]), pe], axis=0)
    return pe


# --------------------------------------------------------------------------------------
# masks and gather indices (integer work: bit-exact)
# --------------------------------------------------------------------------------------
def tube_mask(input_size: Tuple[int, int, int], mask_ratio: float, rng: np.random.RandomState) -> np.ndarray:
    """MG:4-26.  One per-frame pattern (zeros first, then ones, shuffled) tiled over frames.
    `rng.shuffle` on a RandomState seeded with s reproduces `np.random.seed(s); np.random.shuffle`."""
    frames, h, w = input_size
    per_frame = h * w
    n_mask = int(mask_ratio * per_frame)
    m = np.hstack([np.zeros(per_frame - n_mask), np.ones(n_mask)])
    rng.shuffle(m)
    return np.tile(m, (frames, 1)).flatten()


def random_mask(input_size: Tuple[int, int, int], mask_ratio: float, rng: np.random.RandomState) -> np.ndarray:
    """MG:29-49 / MM mask.py:22-37."""
    frames, h, w = input_size
    n = frames * h * w
    n_mask = int(mask_ratio * n)
    m = np.hstack([np.zeros(n - n_mask), np.ones(n_mask)])
    rng.shuffle(m)
    return m


def attention_mask_from_importance(importance: np.ndarray, B: int, mask_ratio: float) -> np.ndarray:
    """E:105-116 with the multinomial draw `importance` (BT, N) given as an input (device RNG is
    not reproducible across back ends, SURVEY.md 7 "hard parts").  Returns bool (B, 1+T*N),
    True = masked, column 0 (cls) False."""
    BT, N = importance.shape
    n_vis = N - int(N * mask_ratio)
    m = np.ones((BT, N), dtype=bool)
    rows = np.arange(BT)[:, None].repeat(n_vis, 1)
    m[rows, importance[:, :n_vis]] = False
    m = m.reshape(B, -1)
    return np.concatenate([np.zeros((B, 1), dtype=bool), m], axis=1)


def visible_indices(mask: np.ndarray) -> np.ndarray:
    """P:659  `x[~mask].reshape(B, -1, C)`: ascending token index of the kept tokens, every
    row must keep the same count.  Returns int32 (B, L)."""
    mask = np.asarray(mask).astype(bool)
    B = mask.shape[0]
    keep = ~mask
    counts = keep.sum(1)
    if not (counts == counts[0]).all():
        raise ValueError("every row of the mask must keep the same number of tokens (P:659 reshape)")
    idx = np.nonzero(keep)[1].reshape(B, int(counts[0]))
    return idx.astype(np.int32)


# --------------------------------------------------------------------------------------
# elementary ops
# --------------------------------------------------------------------------------------
def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """P:117-128.  x * rsqrt(mean(x^2) + eps) * w  (fp32 inside; no mean-centering, no bias)."""
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def gelu(x: torch.Tensor, kind: str) -> torch.Tensor:
    if kind == "erf":                       # nn.GELU(), P:224
        return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))
    if kind == "tanh":                      # flash_attn FusedMLP default
        return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))
    raise ValueError(kind)


def layernorm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float) -> torch.Tensor:
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def patch_embed(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, tubelet: int, patch: int) -> torch.Tensor:
    """P:300-331.  Conv3d with kernel = stride = (tubelet, p, p) == per-patch dot product with the
    weight flattened in (c, dt, dy, dx) order; tokens ordered t-major, then (h, w) row-major.
    x (B,C,T,H,W) -> (B, T'*h*w, D)."""
    B, C, T, H, W = x.shape
    t, h, wd = T // tubelet, H // patch, W // patch
    cols = x.reshape(B, C, t, tubelet, h, patch, wd, patch).permute(0, 2, 4, 6, 1, 3, 5, 7)
    cols = cols.reshape(B, t * h * wd, C * tubelet * patch * patch)
    return cols @ w.reshape(w.shape[0], -1).t() + b


def attention(x: torch.Tensor, p: Dict[str, torch.Tensor], pre: str, num_heads: int, eps: float,
              want_probs: bool = False):
    """P:173-191 (`_naive_attn`).  qkv GEMM without bias, packing (three, head, d) P:175;
    q/k RMSNorm over the flattened head axis (length D) P:178-181; q scaled by hd^-0.5 before
    QK^T P:183; plain softmax over keys P:185; proj with bias P:189."""
    B, N, C = x.shape
    hd = C // num_heads
    qkv = x @ p[pre + "qkv.weight"].t()
    if (pre + "qkv.bias") in p:
        qkv = qkv + p[pre + "qkv.bias"]
    qkv = qkv.reshape(B, N, 3, num_heads, hd)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]                 # (B,N,H,hd)
    if (pre + "q_norm.weight") in p:
        q = rmsnorm(q.reshape(B, N, C), p[pre + "q_norm.weight"], eps).reshape(B, N, num_heads, hd)
        k = rmsnorm(k.reshape(B, N, C), p[pre + "k_norm.weight"], eps).reshape(B, N, num_heads, hd)
    q, k, v = (t.permute(0, 2, 1, 3) for t in (q, k, v))               # (B,H,N,hd)
    att = (q * hd ** -0.5) @ k.transpose(-2, -1)
    att = att.softmax(dim=-1)
    ctx = (att @ v).transpose(1, 2).reshape(B, N, C)
    out = ctx @ p[pre + "proj.weight"].t() + p[pre + "proj.bias"]
    if want_probs:
        return out, dict(q=q, k=k, v=v, ctx=ctx, probs=att)
    return out


def mlp(x: torch.Tensor, p: Dict[str, torch.Tensor], pre: str, kind: str) -> torch.Tensor:
    """P:220-244: fc2(act(fc1(x)))."""
    h = gelu(x @ p[pre + "fc1.weight"].t() + p[pre + "fc1.bias"], kind)
    return h @ p[pre + "fc2.weight"].t() + p[pre + "fc2.bias"]


def block(x: torch.Tensor, p: Dict[str, torch.Tensor], i: int, cfg: StudentConfig) -> torch.Tensor:
    """P:279-292, plain pre-norm form (mathematically identical to the fused residual protocol,
    SURVEY.md appendix A.4).  LayerScale multiplies the branch before the residual add P:284-291."""
    pre = f"blocks.{i}."
    a = attention(rmsnorm(x, p[pre + "norm1.weight"], cfg.rms_eps), p, pre + "attn.", cfg.num_heads, cfg.rms_eps)
    if (pre + "ls1.gamma") in p:
        a = a * p[pre + "ls1.gamma"]
    x = x + a
    m = mlp(rmsnorm(x, p[pre + "norm2.weight"], cfg.rms_eps), p, pre + "mlp.", cfg.gelu)
    if (pre + "ls2.gamma") in p:
        m = m * p[pre + "ls2.gamma"]
    return x + m


def attention_pool(x: torch.Tensor, p: Dict[str, torch.Tensor], pre: str, num_heads: int, ln_eps: float) -> torch.Tensor:
    """P:18-114 (`AttentionPoolingBlock`): query = mean over ALL tokens incl. cls P:110; three
    LayerNorms P:99-101; q/k/v Linear weight-only + separate bias params P:33-40,61-68; q scaled
    P:70; softmax over keys; proj to out_dim with bias P:77."""
    B, N, C = x.shape
    hd = C // num_heads
    xq = x.mean(1, keepdim=True)
    q_in = layernorm(xq, p[pre + "norm1_q.weight"], p[pre + "norm1_q.bias"], ln_eps)
    k_in = layernorm(x, p[pre + "norm1_k.weight"], p[pre + "norm1_k.bias"], ln_eps)
    v_in = layernorm(x, p[pre + "norm1_v.weight"], p[pre + "norm1_v.bias"], ln_eps)
    ca = pre + "cross_attn."
    q = q_in @ p[ca + "q.weight"].t() + p[ca + "q_bias"]
    k = k_in @ p[ca + "k.weight"].t() + p[ca + "k_bias"]
    v = v_in @ p[ca + "v.weight"].t() + p[ca + "v_bias"]
    q = q.reshape(B, 1, num_heads, hd).permute(0, 2, 1, 3) * hd ** -0.5
    k = k.reshape(B, N, num_heads, hd).permute(0, 2, 1, 3)
    v = v.reshape(B, N, num_heads, hd).permute(0, 2, 1, 3)
    att = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, 1, C)
    o = o @ p[ca + "proj.weight"].t() + p[ca + "proj.bias"]
    return o.squeeze(1)


def linear_decoder(x: torch.Tensor, p: Dict[str, torch.Tensor], pre: str, ln_eps: float, norm_type: str = "l2") -> torch.Tensor:
    """P:334-365: Linear -> LayerNorm -> x / ||x||_2 (no epsilon in the division)."""
    y = layernorm(x @ p[pre + "head.weight"].t() + p[pre + "head.bias"],
                  p[pre + "norm.weight"], p[pre + "norm.bias"], ln_eps)
    if norm_type == "l2":
        y = y / y.norm(dim=-1, keepdim=True)
    return y


def mlp_decoder(x: torch.Tensor, p: Dict[str, torch.Tensor], pre: str, ln_eps: float, norm_type: str = "l2") -> torch.Tensor:
    """P:368-403: Linear -> GELU(erf) -> Linear -> LayerNorm -> l2."""
    h = gelu(x @ p[pre + "head.0.weight"].t() + p[pre + "head.0.bias"], "erf")
    y = h @ p[pre + "head.2.weight"].t() + p[pre + "head.2.bias"]
    y = layernorm(y, p[pre + "norm.weight"], p[pre + "norm.bias"], ln_eps)
    if norm_type == "l2":
        y = y / y.norm(dim=-1, keepdim=True)
    return y


# --------------------------------------------------------------------------------------
# the student forward                                                P:629-744
# --------------------------------------------------------------------------------------
def gather_rows(t: torch.Tensor, idx: np.ndarray) -> torch.Tensor:
    """t (B,N,C) or (1,N,C); idx (B,L) int -> (B,L,C)."""
    ii = torch.from_numpy(np.asarray(idx)).long()
    if t.shape[0] == 1:
        return t[0][ii]
    return torch.gather(t, 1, ii[:, :, None].expand(-1, -1, t.shape[-1]))


def student_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, mask: np.ndarray, cfg: StudentConfig,
                    return_blocks: bool = False):
    """P:629-744.  x (B,3,T,H,W); mask bool (B, 1+N) True=masked, col 0 False.
    Returns (x_clip_align (K,B,L,Cc), x_align (B,Cf), x_mae_align (K',B,L-1,Cm)) and optionally
    the list of residual-stream values after every block."""
    dt = p["pos_embed"].dtype
    x = x.to(dt)
    tok = patch_embed(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"],
                      cfg.tubelet_size, cfg.patch_size)                                   # P:630-632
    B = tok.shape[0]
    tok = torch.cat([p["cls_token"].expand(B, -1, -1), tok], dim=1) + p["pos_embed"]        # P:635-656
    idx = visible_indices(mask)                                                           # P:659
    h = gather_rows(tok, idx)
    blocks_out = []
    taps_clip, taps_mae = [], []
    for i in range(cfg.depth):                                                            # P:664-683
        h = block(h, p, i, cfg)
        if return_blocks:
            blocks_out.append(h)
        if i in cfg.clip_return_index:
            taps_clip.append(h)
        if i in cfg.mae_return_index:
            taps_mae.append(h[:, 1:])
    pooled = attention_pool(h, p, "clip_projector.", cfg.attn_pool_num_heads, cfg.ln_eps)   # P:690
    # CLIP branch P:693-720 (taps are in ascending block order, decoder k consumes the k-th tap)
    cpe = gather_rows(p["clip_pos_embed"], idx)
    x_clip = torch.stack([linear_decoder(t + cpe, p, f"clip_decoder.{k}.", cfg.ln_eps)
                          for k, t in enumerate(taps_clip)])
    if cfg.clip_teacher_final_dim > 0:
        x_align = linear_decoder(pooled, p, "final_clip_decoder.", cfg.ln_eps)
    else:
        x_align = pooled
    # MAE branch P:723-742 (cls dropped; mae_pos_embed has no cls row)
    mpe = gather_rows(p["mae_pos_embed"], idx[:, 1:] - 1)
    x_mae = torch.stack([mlp_decoder(t + mpe, p, f"mae_decoder.{k}.", cfg.ln_eps)
                         for k, t in enumerate(taps_mae)])
    if return_blocks:
        return (x_clip, x_align, x_mae), blocks_out
    return x_clip, x_align, x_mae


def distill_losses(outputs, targets, clip_loss_ratio=(1.0, 1.0), mae_loss_ratio=1.0):
    """E:131-148: (2 - 2 * <s, t>).mean() per head, weighted sum."""
    oc, of, om = outputs
    tc, tf, tm = targets
    l_mid = (2 - 2 * (oc * tc).sum(-1)).mean()
    l_fin = (2 - 2 * (of * tf).sum(-1)).mean()
    l_mae = (2 - 2 * (om * tm).sum(-1)).mean()
    total = l_mid * clip_loss_ratio[0] + l_fin * clip_loss_ratio[1] + l_mae * mae_loss_ratio
    return total, (l_mid, l_fin, l_mae)


def image_pos_table(p: Dict[str, torch.Tensor], name: str, cfg: StudentConfig) -> torch.Tensor:
    """V:592-607 / V:652-667: positional table of image mode (T = 1): the separate `img_` table, else the video table with
    the patch rows averaged over the frames (cls row kept)."""
    if cfg.sep_image_video_pos_embed:
        return p[name.replace("pos_embed", "img_pos_embed")]
    tab = p[name]
    T, h, w = cfg.grid
    img = tab[:, 1:, :].reshape(1, T, h * w, cfg.embed_dim).mean(dim=1)
    return torch.cat([tab[:, 0:1, :], img], dim=1)


def encoder_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, mask: Optional[np.ndarray], cfg: StudentConfig,
                    use_image: bool = False, x_vis_return_idx: int = -1) -> Dict[str, torch.Tensor]:
    """The distillation student D:612-697 (mask given, 2-tuple = x_clip_align, x_align) and the stage-2 vision encoder V:578-685
    (mask optional, image mode, early exit; 4-tuple = x_vis, x_pool_vis, x_clip_align, x_align) in one restatement: both are the
    trunk of P:629-720 without the MAE branch.  Returns every output by name."""
    dt = p["pos_embed"].dtype
    tok = patch_embed(x.to(dt), p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], cfg.tubelet_size, cfg.patch_size)
    B = tok.shape[0]
    pos = image_pos_table(p, "pos_embed", cfg) if use_image else p["pos_embed"]                 # V:592-607
    tok = torch.cat([p["cls_token"].expand(B, -1, -1), tok], dim=1) + pos                      # V:585-608
    if mask is not None:
        idx = visible_indices(mask)                                                           # V:611-612 / D:641
    else:
        idx = np.tile(np.arange(tok.shape[1], dtype=np.int32), (B, 1))       # V:613-614
    h = gather_rows(tok, idx)
    taps = []
    last = cfg.depth + x_vis_return_idx                                                       # V:633-635
    for i in range(cfg.depth):
        h = block(h, p, i, cfg)
        if i in cfg.clip_return_index:
            taps.append(h)
        if i == last:
            break
    out = {"x_vis": h}
    pooled = attention_pool(h, p, "clip_projector.", cfg.attn_pool_num_heads, cfg.ln_eps)       # V:646 / D:665
    out["x_pool_vis"] = pooled
    dec = mlp_decoder if cfg.clip_decoder_kind == "mlp" else linear_decoder                    # D:412-414
    out["x_align"] = dec(pooled, p, "final_clip_decoder.", cfg.ln_eps) if cfg.clip_teacher_final_dim > 0 else pooled
    cpos = image_pos_table(p, "clip_pos_embed", cfg) if use_image else p["clip_pos_embed"]      # V:652-669
    cpe = gather_rows(cpos, idx)
    out["x_clip_align"] = torch.stack([dec(t + cpe, p, f"clip_decoder.{k}.", cfg.ln_eps) for k, t in enumerate(taps)])
    return out


def finetune_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, cfg: StudentConfig) -> torch.Tensor:
    """single_modality/models/internvideo2.py:500-543 ("F:"): every token through the blocks, attention pool (F:538), `fc_norm`
    LayerNorm (default eps 1e-5, F:439), `head` Linear (F:441) -> logits (B, num_classes)."""
    tok = patch_embed(x.to(p["pos_embed"].dtype), p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], cfg.tubelet_size, cfg.patch_size)
    B = tok.shape[0]
    h = torch.cat([p["cls_token"].expand(B, -1, -1), tok], dim=1) + p["pos_embed"]
    for i in range(cfg.depth):
        h = block(h, p, i, cfg)
    pooled = attention_pool(h, p, "clip_projector.", cfg.attn_pool_num_heads, cfg.ln_eps)
    y = layernorm(pooled, p["fc_norm.weight"], p["fc_norm.bias"], 1e-5)
    return y @ p["head.weight"].t() + p["head.bias"]


def finetune_param_shapes(cfg: StudentConfig, num_classes: int) -> Dict[str, Tuple[int, ...]]:
    full = param_shapes(cfg)
    s = {k: v for k, v in full.items() if k.startswith(("blocks.", "clip_projector.", "patch_embed.")) or k in ("cls_token", "pos_embed")}
    s["fc_norm.weight"] = (cfg.clip_embed_dim,); s["fc_norm.bias"] = (cfg.clip_embed_dim,)
    s["head.weight"] = (num_classes, cfg.clip_embed_dim); s["head.bias"] = (num_classes,)
    return s


def synthetic_finetune_params(cfg: StudentConfig, num_classes: int, seed: int = 0) -> Dict[str, torch.Tensor]:
    base = synthetic_params(cfg, seed=seed)
    rng = np.random.Generator(np.random.PCG64(500 + seed))
    out = {k: base[k] for k in finetune_param_shapes(cfg, num_classes) if k in base}
    out["fc_norm.weight"] = torch.from_numpy(1.0 + 0.1 * rng.standard_normal(cfg.clip_embed_dim)).float()
    out["fc_norm.bias"] = torch.from_numpy(0.02 * rng.standard_normal(cfg.clip_embed_dim)).float()
    out["head.weight"] = torch.from_numpy(0.05 * rng.standard_normal((num_classes, cfg.clip_embed_dim))).float()
    out["head.bias"] = torch.from_numpy(0.02 * rng.standard_normal(num_classes)).float()
    return out


# --------------------------------------------------------------------------------------
# the frozen CLIP teacher                                            T: = single_modality/models/internvl_clip_vision.py
# --------------------------------------------------------------------------------------
def clip_teacher_forward(p: Dict[str, torch.Tensor], image: torch.Tensor, cfg: StudentConfig, return_index: List[int],
                         norm_type: str = "l2"):
    """T:411-465 (`InternVL_CLIP.forward`): every frame is its own sequence of 1 + H*W tokens through the student's block
    (T:157-300 == P:149-297); tapped features are merged over the frames (cls rows averaged) and l2-normalised, the pooled
    feature is averaged over the frames and l2-normalised, and the pooling query's head-averaged attention over the patch keys
    is returned for attention-guided masking.  -> (z (K,B,1+T*HW,C), x (B,Cf), attn (B*T, HW))"""
    dt = p["pos_embed"].dtype
    tok = patch_embed(image.to(dt), p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], 1, cfg.patch_size)   # T:412, (B, T*HW, C)
    B, T = image.shape[0], image.shape[2]
    HW, C = tok.shape[1] // T, tok.shape[2]
    x = tok.reshape(B * T, HW, C)                                                             # T:413-414
    x = torch.cat([p["cls_token"].expand(B * T, -1, -1), x], dim=1) + p["pos_embed"]           # T:416-418
    z = []
    for i in range(cfg.depth):                                                                # T:420-433
        x = block(x, p, i, cfg)
        if i in return_index:
            z.append(x)
    # T:440 attention pooling with return_attn: same arithmetic as attention_pool() plus attn.mean(1) (T:82-83)
    pre = "clip_projector."
    H = cfg.attn_pool_num_heads
    hd = C // H
    N = x.shape[1]
    q_in = layernorm(x.mean(1, keepdim=True), p[pre + "norm1_q.weight"], p[pre + "norm1_q.bias"], cfg.ln_eps)
    k_in = layernorm(x, p[pre + "norm1_k.weight"], p[pre + "norm1_k.bias"], cfg.ln_eps)
    v_in = layernorm(x, p[pre + "norm1_v.weight"], p[pre + "norm1_v.bias"], cfg.ln_eps)
    ca = pre + "cross_attn."
    q = (q_in @ p[ca + "q.weight"].t() + p[ca + "q_bias"]).reshape(B * T, 1, H, hd).permute(0, 2, 1, 3) * hd ** -0.5
    k = (k_in @ p[ca + "k.weight"].t() + p[ca + "k_bias"]).reshape(B * T, N, H, hd).permute(0, 2, 1, 3)
    v = (v_in @ p[ca + "v.weight"].t() + p[ca + "v_bias"]).reshape(B * T, N, H, hd).permute(0, 2, 1, 3)
    att = (q @ k.transpose(-2, -1)).softmax(dim=-1)                                            # (BT, H, 1, N)
    o = (att @ v).transpose(1, 2).reshape(B * T, 1, C)
    pooled = (o @ p[ca + "proj.weight"].t() + p[ca + "proj.bias"]).squeeze(1)                  # (BT, Cf)
    attn = att.mean(1)[:, 0, 1:]                                                              # T:83, T:463
    if norm_type == "l2":                                                                     # T:445-456
        zs = torch.stack(z)                                                                   # (K, BT, HW+1, C)
        K = zs.shape[0]
        cls, pat = zs[:, :, :1, :], zs[:, :, 1:, :]
        cls = cls.reshape(K, B, T, 1, C).mean(2)
        pat = pat.reshape(K, B, T * HW, C)
        zs = torch.cat([cls, pat], dim=2)
        zs = zs / zs.norm(dim=-1, keepdim=True)
        xf = pooled.reshape(B, T, -1).mean(1)
        xf = xf / xf.norm(dim=-1, keepdim=True)
        return zs, xf, attn
    return torch.stack(z), pooled, attn


def teacher_param_shapes(cfg: StudentConfig) -> Dict[str, Tuple[int, ...]]:
    """state_dict of InternVL_CLIP (T:385-405): per-frame pos_embed (1, HW+1, D), Conv3d kernel (D, 3, 1, p, p), blocks, projector."""
    full = param_shapes(cfg)
    g = cfg.img_size // cfg.patch_size
    s = {k: v for k, v in full.items() if k.startswith("blocks.") or k.startswith("clip_projector.") or k == "cls_token"}
    s["pos_embed"] = (1, g * g + 1, cfg.embed_dim)
    s["patch_embed.proj.weight"] = (cfg.embed_dim, cfg.in_chans, 1, cfg.patch_size, cfg.patch_size)
    s["patch_embed.proj.bias"] = (cfg.embed_dim,)
    return s


def synthetic_teacher_params(cfg: StudentConfig, seed: int = 0, gamma: float = 0.5) -> Dict[str, torch.Tensor]:
    """deterministic teacher weights (numpy PCG64): N(0, 0.02) matrices, LayerScale ~ gamma (T:347 init 0.1 for the real 6B),
    2-D sincos + noise positional table."""
    rng = np.random.Generator(np.random.PCG64(seed))
    shapes = teacher_param_shapes(cfg)
    g = cfg.img_size // cfg.patch_size
    pe = sincos_pos_embed_3d(cfg.embed_dim, g, 1, cls_token=True)
    out: Dict[str, torch.Tensor] = {}
    for k in sorted(shapes):
        shp = shapes[k]
        if k == "pos_embed":
            a = pe[None] + 0.01 * rng.standard_normal(shp)
        elif k.endswith("gamma"):
            a = gamma * (1.0 + 0.1 * rng.standard_normal(shp))
        elif k.endswith("weight") and ("norm" in k.split(".")[-2]):
            a = 1.0 + 0.1 * rng.standard_normal(shp)
        elif k.endswith("bias") or k.endswith("_bias"):
            a = 0.02 * rng.standard_normal(shp)
        else:
            a = 0.02 * rng.standard_normal(shp)
        out[k] = torch.from_numpy(np.ascontiguousarray(a)).float()
    return out


# --------------------------------------------------------------------------------------
# stage-2 contrastive logits                                          C:15-103, C:200-216
# --------------------------------------------------------------------------------------
def contrastive_sim(v: torch.Tensor, t: torch.Tensor, temp) -> Tuple[torch.Tensor, torch.Tensor]:
    """C:31-53 (2-D inputs): F.normalize(eps=1e-12) both; sim_v2t = v @ t^T / temp."""
    v = v / v.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    t = t / t.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    s = v @ t.t() / temp
    return s, s.t()


def vtc_loss(v: torch.Tensor, t: torch.Tensor, idx: Optional[torch.Tensor], temp) -> torch.Tensor:
    """C:65-103 with all_gather already applied by the caller (rank-order concatenation, U:198-202).
    Targets eq(idx, idx^T) row-normalised C:208-212; loss = (CE(v2t) + CE(t2v)) / 2."""
    s_v2t, s_t2v = contrastive_sim(v, t, temp)
    if idx is not None:
        idx = idx.view(-1, 1)
        tg = torch.eq(idx, idx.t()).to(s_v2t.dtype)
        tg = tg / tg.sum(1, keepdim=True)
    else:
        tg = torch.eye(s_v2t.shape[0], dtype=s_v2t.dtype)
    l1 = -(F.log_softmax(s_v2t, dim=1) * tg).sum(1).mean()
    l2 = -(F.log_softmax(s_t2v, dim=1) * tg).sum(1).mean()
    return (l1 + l2) / 2


def clamp_temperature(temp: torch.Tensor) -> torch.Tensor:
    """MM internvideo2_stage2_visual.py:291-294: clamp to [0.001, 0.5] at the start of every forward."""
    return temp.clamp(0.001, 0.5)


# --------------------------------------------------------------------------------------
# InternVideo1 VideoMAE pixel target                 IV1-MAE/engine_for_pretraining.py:66-98
# --------------------------------------------------------------------------------------
def videomae_pixel_target(videos: torch.Tensor, mask: np.ndarray, patch: int, tubelet: int = 2,
                          mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), normalize: bool = True) -> torch.Tensor:
    """Un-normalise, cut into (t 2)(h p)(w p) cubes, per-patch mean / unbiased-var normalise over the
    (2 p p) axis per channel, flatten to (n, 2 p p c), gather the masked tokens.
    videos (B,3,T,H,W) normalised; mask bool (B,N) True = masked -> (B, N_mask, tubelet*p*p*3)."""
    B, C, T, H, W = videos.shape
    m = torch.tensor(mean, dtype=videos.dtype)[None, :, None, None, None]
    s = torch.tensor(std, dtype=videos.dtype)[None, :, None, None, None]
    un = videos * s + m
    t, h, w = T // tubelet, H // patch, W // patch
    sq = un.reshape(B, C, t, tubelet, h, patch, w, patch).permute(0, 2, 4, 6, 3, 5, 7, 1)
    sq = sq.reshape(B, t * h * w, tubelet * patch * patch, C)
    if normalize:
        sq = (sq - sq.mean(dim=-2, keepdim=True)) / (sq.var(dim=-2, unbiased=True, keepdim=True).sqrt() + 1e-6)
    pt = sq.reshape(B, t * h * w, tubelet * patch * patch * C)
    mm = torch.from_numpy(np.asarray(mask).astype(bool))
    return pt[mm].reshape(B, -1, pt.shape[-1])


# --------------------------------------------------------------------------------------
# VideoMAE pixel-reconstruction model        MP: = InternVideo1/Pretrain/VideoMAE/modeling_pretrain.py, MF: = modeling_finetune.py
# --------------------------------------------------------------------------------------
@dataclass
class MaeConfig:
    img_size: int = 224
    patch_size: int = 16
    tubelet_size: int = 2
    num_frames: int = 16
    enc_dim: int = 768
    enc_depth: int = 12
    enc_heads: int = 12
    dec_dim: int = 384
    dec_depth: int = 4
    dec_heads: int = 6
    mlp_ratio: float = 4.0
    qkv_bias: bool = True
    init_values: float = 0.0
    ln_eps: float = 1e-6

    @property
    def num_patches(self) -> int:
        g = self.img_size // self.patch_size
        return (self.num_frames // self.tubelet_size) * g * g

    @property
    def num_classes(self) -> int:
        return 3 * self.tubelet_size * self.patch_size ** 2


def sinusoid_table(n_position: int, d_hid: int) -> torch.Tensor:
    """MF:224-241 (float64 numpy, then fp32)."""
    tab = np.array([[pos / np.power(10000, 2 * (j // 2) / d_hid) for j in range(d_hid)] for pos in range(n_position)])
    tab[:, 0::2] = np.sin(tab[:, 0::2])
    tab[:, 1::2] = np.cos(tab[:, 1::2])
    return torch.tensor(tab, dtype=torch.float).unsqueeze(0)


def mae_block(x: torch.Tensor, p: Dict[str, torch.Tensor], pre: str, heads: int, eps: float) -> torch.Tensor:
    """MF:170-181 with MF:104-129: LayerNorm -> qkv (bias = [q_bias, 0, v_bias]) -> softmax(q k^T hd^-0.5) v -> proj;
    optional gamma_1 / gamma_2; erf-GELU MLP."""
    B, N, C = x.shape
    hd = C // heads
    h = layernorm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"], eps)
    qkv = h @ p[pre + "attn.qkv.weight"].t()
    if (pre + "attn.q_bias") in p:
        qkv = qkv + torch.cat([p[pre + "attn.q_bias"], torch.zeros_like(p[pre + "attn.v_bias"]), p[pre + "attn.v_bias"]])
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
    a = (att @ v).transpose(1, 2).reshape(B, N, C)
    a = a @ p[pre + "attn.proj.weight"].t() + p[pre + "attn.proj.bias"]
    if (pre + "gamma_1") in p:
        a = p[pre + "gamma_1"] * a
    x = x + a
    h = layernorm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"], eps)
    m = gelu(h @ p[pre + "mlp.fc1.weight"].t() + p[pre + "mlp.fc1.bias"], "erf") @ p[pre + "mlp.fc2.weight"].t() + p[pre + "mlp.fc2.bias"]
    if (pre + "gamma_2") in p:
        m = p[pre + "gamma_2"] * m
    return x + m


def videomae_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, mask: np.ndarray, cfg: MaeConfig) -> torch.Tensor:
    """MP:375-392: encoder on the visible tokens (MP:125-139), encoder_to_decoder, mask-token scatter with the decoder's sinusoid
    table, decoder blocks on all tokens, head(norm(.)) on the last N_mask rows.  mask bool (B, N), True = masked."""
    mm = torch.from_numpy(np.asarray(mask).astype(bool))
    tok = patch_embed(x, p["encoder.patch_embed.proj.weight"], p["encoder.patch_embed.proj.bias"], cfg.tubelet_size, cfg.patch_size)
    B, N, Ce = tok.shape
    tok = tok + sinusoid_table(N, Ce).to(tok.dtype)                                            # MP:129
    h = tok[~mm].reshape(B, -1, Ce)                                                           # MP:133
    for i in range(cfg.enc_depth):
        h = mae_block(h, p, f"encoder.blocks.{i}.", cfg.enc_heads, cfg.ln_eps)
    h = layernorm(h, p["encoder.norm.weight"], p["encoder.norm.bias"], cfg.ln_eps)             # MP:138
    h = h @ p["encoder_to_decoder.weight"].t()                                                # MP:377
    Cd = h.shape[-1]
    pos = sinusoid_table(N, Cd).to(h.dtype).expand(B, -1, -1)
    pos_vis, pos_msk = pos[~mm].reshape(B, -1, Cd), pos[mm].reshape(B, -1, Cd)                  # MP:382-385
    full = torch.cat([h + pos_vis, p["mask_token"] + pos_msk], dim=1)                          # MP:387-389
    for i in range(cfg.dec_depth):
        full = mae_block(full, p, f"decoder.blocks.{i}.", cfg.dec_heads, cfg.ln_eps)
    tail = full[:, -pos_msk.shape[1]:]                                                        # MP:264
    tail = layernorm(tail, p["decoder.norm.weight"], p["decoder.norm.bias"], cfg.ln_eps)
    return tail @ p["decoder.head.weight"].t() + p["decoder.head.bias"]


def flash_attn_func_contract(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, softmax_scale: float) -> torch.Tensor:
    """flash_attn.flash_attn_func as documented (flash-attn 2.x README / docstring: "q, k, v: (batch_size, seqlen, nheads, headdim)",
    returns (batch_size, seqlen, nheads, headdim)); non-causal, no dropout.  Pinned third-party dependency of the reference
    (single_modality/requirements.txt: flash_attn==2.0.8), absent here: restated from its published contract."""
    a = torch.einsum("bshd,bthd->bhst", q * softmax_scale, k).softmax(dim=-1)
    return torch.einsum("bhst,bthd->bshd", a, v)


def videomae_teacher_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, mask: Optional[np.ndarray], heads: int, depth: int,
                             return_index: List[int], tubelet: int, patch: int, eps: float = 1e-6, as_coded: bool = True,
                             norm_type: str = "l2") -> torch.Tensor:
    """single_modality/models/videomae.py:285-312 ("VT:").  Blocks VT:99-132; attention VT:85-98 -- as coded, q/k/v of shape
    (B, H, N, hd) go to flash_attn_func (contract (B, S, Hh, d)) and the result is reshaped to (B, N, -1); as_coded=False computes
    the token-to-token attention of InternVideo1's VideoMAE (MF:104-129) instead.  p["pos_embed"]: the (1, N, C) table."""
    tok = patch_embed(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], tubelet, patch)
    B, N, C = tok.shape
    h = tok + p["pos_embed"].to(tok.dtype)                                                     # VT:289-290
    if mask is not None:
        mm = torch.from_numpy(np.asarray(mask).astype(bool))
        h = h[~mm].reshape(B, -1, C)                                                          # VT:293-294
    hd = C // heads
    z = []
    for i in range(depth):
        pre = f"blocks.{i}."
        n1 = layernorm(h, p[pre + "norm1.weight"], p[pre + "norm1.bias"], eps)
        qkv = n1 @ p[pre + "attn.qkv.weight"].t()
        if (pre + "attn.q_bias") in p:
            qkv = qkv + torch.cat([p[pre + "attn.q_bias"], torch.zeros_like(p[pre + "attn.v_bias"]), p[pre + "attn.v_bias"]])
        Nn = h.shape[1]
        qkv = qkv.reshape(B, Nn, 3, heads, hd).permute(2, 0, 3, 1, 4)                          # VT:93: (3, B, H, N, hd)
        q, k, v = qkv[0], qkv[1], qkv[2]
        if as_coded:
            a = flash_attn_func_contract(q, k, v, hd ** -0.5).reshape(B, Nn, -1)               # VT:96
        else:
            att = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(dim=-1)
            a = (att @ v).transpose(1, 2).reshape(B, Nn, C)
        a = a @ p[pre + "attn.proj.weight"].t() + p[pre + "attn.proj.bias"]
        if (pre + "gamma_1") in p:
            a = p[pre + "gamma_1"] * a
        h = h + a
        n2 = layernorm(h, p[pre + "norm2.weight"], p[pre + "norm2.bias"], eps)
        m = gelu(n2 @ p[pre + "mlp.fc1.weight"].t() + p[pre + "mlp.fc1.bias"], "erf") @ p[pre + "mlp.fc2.weight"].t() + p[pre + "mlp.fc2.bias"]
        if (pre + "gamma_2") in p:
            m = p[pre + "gamma_2"] * m
        h = h + m
        if i == depth - 1:
            h = layernorm(h, p["norm.weight"], p["norm.bias"], eps)                            # VT:300-301
        if i in return_index:
            z.append(h)
    out = torch.stack(z)
    if norm_type == "l2":
        out = out / out.norm(dim=-1, keepdim=True)                                             # VT:306-307
    return out


def mae_teacher_params(cfg: MaeConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """teacher weights = the `encoder.` sub-tree of synthetic_mae_params (what VT:315-326 keeps of a VideoMAE checkpoint)"""
    return {k[8:]: v for k, v in synthetic_mae_params(cfg, seed=seed).items() if k.startswith("encoder.")}


def mae_param_shapes(cfg: MaeConfig) -> Dict[str, Tuple[int, ...]]:
    s: Dict[str, Tuple[int, ...]] = {}
    s["mask_token"] = (1, 1, cfg.dec_dim)
    s["encoder.patch_embed.proj.weight"] = (cfg.enc_dim, 3, cfg.tubelet_size, cfg.patch_size, cfg.patch_size)
    s["encoder.patch_embed.proj.bias"] = (cfg.enc_dim,)
    for side, D, depth in (("encoder", cfg.enc_dim, cfg.enc_depth), ("decoder", cfg.dec_dim, cfg.dec_depth)):
        Hm = int(D * cfg.mlp_ratio)
        for i in range(depth):
            b = f"{side}.blocks.{i}."
            if cfg.init_values > 0:
                s[b + "gamma_1"] = (D,); s[b + "gamma_2"] = (D,)
            for n in ("norm1", "norm2"):
                s[b + n + ".weight"] = (D,); s[b + n + ".bias"] = (D,)
            if cfg.qkv_bias:
                s[b + "attn.q_bias"] = (D,); s[b + "attn.v_bias"] = (D,)
            s[b + "attn.qkv.weight"] = (3 * D, D)
            s[b + "attn.proj.weight"] = (D, D); s[b + "attn.proj.bias"] = (D,)
            s[b + "mlp.fc1.weight"] = (Hm, D); s[b + "mlp.fc1.bias"] = (Hm,)
            s[b + "mlp.fc2.weight"] = (D, Hm); s[b + "mlp.fc2.bias"] = (D,)
        s[f"{side}.norm.weight"] = (D,); s[f"{side}.norm.bias"] = (D,)
    s["decoder.head.weight"] = (cfg.num_classes, cfg.dec_dim)
    s["decoder.head.bias"] = (cfg.num_classes,)
    s["encoder_to_decoder.weight"] = (cfg.dec_dim, cfg.enc_dim)
    return s


def synthetic_mae_params(cfg: MaeConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    rng = np.random.Generator(np.random.PCG64(seed))
    out: Dict[str, torch.Tensor] = {}
    shapes = mae_param_shapes(cfg)
    for k in sorted(shapes):
        shp = shapes[k]
        if "gamma_" in k:
            a = 0.5 * (1.0 + 0.1 * rng.standard_normal(shp))
        elif k.endswith("weight") and "norm" in k.split(".")[-2]:
            a = 1.0 + 0.1 * rng.standard_normal(shp)
        elif k.endswith("bias") or k.endswith("_bias"):
            a = 0.02 * rng.standard_normal(shp)
        elif k == "mask_token":
            a = 0.02 * rng.standard_normal(shp)
        else:
            fan = shp[1] if len(shp) == 2 else int(np.prod(shp[1:]))
            a = rng.standard_normal(shp) / math.sqrt(fan)                      # xavier-like scale (MP:100-103)
        out[k] = torch.from_numpy(np.ascontiguousarray(a)).float()
    return out


def synthetic_mae_batch(cfg: MaeConfig, B: int, n_mask: int, seed: int = 0):
    """ImageNet-normalised random-pixel clips + a random mask with n_mask masked tokens per clip"""
    rng = np.random.Generator(np.random.PCG64(2000 + seed))
    video = rng.random((B, 3, cfg.num_frames, cfg.img_size, cfg.img_size), dtype=np.float32)
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)[None, :, None, None, None]
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)[None, :, None, None, None]
    video = (video - mean) / std
    mask = np.zeros((B, cfg.num_patches), dtype=bool)
    for b in range(B):
        mask[b, rng.permutation(cfg.num_patches)[:n_mask]] = True
    return torch.from_numpy(video), mask


def named_mae_config(name: str) -> MaeConfig:
    if name == "mae_tiny":       # hd 32 / 16, gamma on, q/v bias on
        return MaeConfig(img_size=32, patch_size=8, tubelet_size=2, num_frames=4, enc_dim=64, enc_depth=2, enc_heads=2,
                         dec_dim=32, dec_depth=2, dec_heads=2, mlp_ratio=4.0, qkv_bias=True, init_values=0.1)
    if name == "mae_tiny88":     # hd 88 encoder like ViT-g (1408 / 16), no gamma (the shipped recipes: init_values = 0)
        return MaeConfig(img_size=28, patch_size=14, tubelet_size=2, num_frames=8, enc_dim=176, enc_depth=2, enc_heads=2,
                         dec_dim=64, dec_depth=1, dec_heads=2, mlp_ratio=48 / 11, qkv_bias=True, init_values=0.0)
    if name == "mae_teach":      # teacher flavour: 16 frames, 4x4 grid of 8-pixel patches (positional table resized from the 8x14x14 one)
        return MaeConfig(img_size=32, patch_size=8, tubelet_size=2, num_frames=16, enc_dim=96, enc_depth=3, enc_heads=4,
                         dec_dim=32, dec_depth=1, dec_heads=2, mlp_ratio=4.0, qkv_bias=True, init_values=0.0)
    if name == "mae_base":       # pretrain_mae_base_patch16_224 (MP:416-434)
        return MaeConfig()
    raise KeyError(name)


# --------------------------------------------------------------------------------------
# deterministic synthetic parameters / inputs (shared by golden generation and the tests)
# --------------------------------------------------------------------------------------
def param_shapes(cfg: StudentConfig) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys and shapes of PretrainInternVideo2 with sep_pos_embed=False (P:471-553)."""
    D, Hm, N = cfg.embed_dim, cfg.mlp_hidden, cfg.num_patches
    s: Dict[str, Tuple[int, ...]] = {}
    s["cls_token"] = (1, 1, D)
    s["pos_embed"] = (1, N + 1, D)
    s["clip_pos_embed"] = (1, N + 1, D)
    if cfg.has_mae:
        s["mae_pos_embed"] = (1, N, D)
    if cfg.sep_image_video_pos_embed:
        s["img_pos_embed"] = (1, cfg.grid[1] * cfg.grid[2] + 1, D)
        s["clip_img_pos_embed"] = (1, cfg.grid[1] * cfg.grid[2] + 1, D)
    s["patch_embed.proj.weight"] = (D, cfg.in_chans, cfg.tubelet_size, cfg.patch_size, cfg.patch_size)
    s["patch_embed.proj.bias"] = (D,)
    for i in range(cfg.depth):
        b = f"blocks.{i}."
        s[b + "norm1.weight"] = (D,)
        s[b + "attn.qkv.weight"] = (3 * D, D)
        s[b + "attn.proj.weight"] = (D, D)
        s[b + "attn.proj.bias"] = (D,)
        s[b + "attn.q_norm.weight"] = (D,)
        s[b + "attn.k_norm.weight"] = (D,)
        s[b + "ls1.gamma"] = (D,)
        s[b + "norm2.weight"] = (D,)
        s[b + "mlp.fc1.weight"] = (Hm, D)
        s[b + "mlp.fc1.bias"] = (Hm,)
        s[b + "mlp.fc2.weight"] = (D, Hm)
        s[b + "mlp.fc2.bias"] = (D,)
        s[b + "ls2.gamma"] = (D,)
    cp = "clip_projector."
    for n in ("q", "k", "v"):
        s[cp + f"norm1_{n}.weight"] = (D,)
        s[cp + f"norm1_{n}.bias"] = (D,)
        s[cp + f"cross_attn.{n}_bias"] = (D,)
        s[cp + f"cross_attn.{n}.weight"] = (D, D)
    s[cp + "cross_attn.proj.weight"] = (cfg.clip_embed_dim, D)
    s[cp + "cross_attn.proj.bias"] = (cfg.clip_embed_dim,)
    def _decoder(d, cin, cout):
        if cfg.clip_decoder_kind == "mlp":                     # MLP_Decoder (D:374-409)
            s[d + "head.0.weight"] = (cin, cin)
            s[d + "head.0.bias"] = (cin,)
            s[d + "head.2.weight"] = (cout, cin)
            s[d + "head.2.bias"] = (cout,)
        else:
            s[d + "head.weight"] = (cout, cin)
            s[d + "head.bias"] = (cout,)
        s[d + "norm.weight"] = (cout,)
        s[d + "norm.bias"] = (cout,)

    for k in range(len(cfg.clip_return_index)):
        _decoder(f"clip_decoder.{k}.", D, cfg.clip_teacher_embed_dim)
    if cfg.clip_teacher_final_dim > 0:
        _decoder("final_clip_decoder.", cfg.clip_embed_dim, cfg.clip_teacher_final_dim)
    for k in range(cfg.mae_return_layer if cfg.has_mae else 0):
        d = f"mae_decoder.{k}."
        s[d + "head.0.weight"] = (D, D)
        s[d + "head.0.bias"] = (D,)
        s[d + "head.2.weight"] = (cfg.mae_teacher_embed_dim, D)
        s[d + "head.2.bias"] = (cfg.mae_teacher_embed_dim,)
        s[d + "norm.weight"] = (cfg.mae_teacher_embed_dim,)
        s[d + "norm.bias"] = (cfg.mae_teacher_embed_dim,)
    return s


def iter_synthetic_params(cfg: StudentConfig, seed: int = 0, gamma: float = 1.0, dtype=torch.float32):
    """synthetic_params as a stream of (name, tensor) in generation order (sorted names, ONE PCG64 stream): the 6B model's 5.9 G parameters are
    filled one tensor at a time instead of through a 24 GB dictionary"""
    rng = np.random.Generator(np.random.PCG64(seed))
    shapes = param_shapes(cfg)
    pe = sincos_pos_embed_3d(cfg.embed_dim, cfg.grid[1], cfg.grid[0], cls_token=True)
    for k in sorted(shapes):
        shp = shapes[k]
        if k == "pos_embed" or k == "clip_pos_embed":
            a = pe[None] + 0.01 * rng.standard_normal(shp)
        elif k == "mae_pos_embed":
            a = pe[None, 1:] + 0.01 * rng.standard_normal(shp)
        elif k in ("img_pos_embed", "clip_img_pos_embed"):
            a = sincos_pos_embed_3d(cfg.embed_dim, cfg.grid[1], 1, cls_token=True)[None] + 0.01 * rng.standard_normal(shp)
        elif k.endswith("gamma"):
            a = gamma * (1.0 + 0.1 * rng.standard_normal(shp))
        elif k.endswith("weight") and ("norm" in k.split(".")[-2]):
            a = 1.0 + 0.1 * rng.standard_normal(shp)
        elif k.endswith("bias") or k.endswith("_bias"):
            a = 0.02 * rng.standard_normal(shp)
        else:
            a = 0.02 * rng.standard_normal(shp)
            if k.startswith("blocks.") and (k.endswith("attn.proj.weight") or k.endswith("mlp.fc2.weight")):
                i = int(k.split(".")[1])
                a = a / math.sqrt(2.0 * (i + 1))
        yield k, torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def synthetic_params(cfg: StudentConfig, seed: int = 0, gamma: float = 1.0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Deterministic, platform-independent parameter fill (numpy PCG64, keys in sorted order) used by
    the golden fixtures and by every parity test.  Statistics follow the reference init (P:588-603:
    N(0, 0.02) matrices, proj/fc2 scaled by 1/sqrt(2(i+1)), sincos position tables) except that
    LayerScale gamma is O(1) so that block errors are visible (SURVEY.md 7 "hard parts"), norm
    weights are perturbed around 1 and biases are non-zero so that every term is exercised."""
    return dict(iter_synthetic_params(cfg, seed, gamma, dtype))


def synthetic_batch(cfg: StudentConfig, B: int, n_vis_per_frame: int, seed: int = 0, dtype=torch.float32):
    """uniform [0,1) random-pixel clips, per-frame randperm visible set (cls visible),
    l2-normalised gaussian targets.  numpy PCG64 so that it is identical on every box."""
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    T, h, w = cfg.grid
    video = rng.random((B, cfg.in_chans, cfg.num_frames, cfg.img_size, cfg.img_size), dtype=np.float32)
    mask = np.ones((B, T, h * w), dtype=bool)
    for b in range(B):
        for t in range(T):
            mask[b, t, rng.permutation(h * w)[:n_vis_per_frame]] = False
    mask = np.concatenate([np.zeros((B, 1), dtype=bool), mask.reshape(B, -1)], axis=1)
    L = 1 + T * n_vis_per_frame

    def unit(shape):
        a = rng.standard_normal(shape).astype(np.float32)
        return a / np.linalg.norm(a, axis=-1, keepdims=True)

    tg_clip = unit((cfg.clip_return_layer, B, L, cfg.clip_teacher_embed_dim))
    tg_final = unit((B, cfg.clip_teacher_final_dim if cfg.clip_teacher_final_dim > 0 else cfg.clip_embed_dim))
    tg_mae = unit((cfg.mae_return_layer, B, L - 1, cfg.mae_teacher_embed_dim))
    return (torch.from_numpy(video).to(dtype), mask,
            (torch.from_numpy(tg_clip).to(dtype), torch.from_numpy(tg_final).to(dtype), torch.from_numpy(tg_mae).to(dtype)))


# named configurations used across tests / bench (BASELINE.json configs)
def named_config(name: str) -> StudentConfig:
    if name == "tiny64":      # hd = 64, fixture-sized
        return StudentConfig(img_size=56, embed_dim=128, depth=3, num_heads=2, mlp_ratio=4.0, num_frames=4,
                             attn_pool_num_heads=2, clip_embed_dim=64, clip_teacher_embed_dim=96,
                             clip_teacher_final_dim=64, clip_return_layer=2, mae_teacher_embed_dim=128,
                             mae_return_layer=2)
    if name == "tiny88":      # hd = 88 like the 1B model, mlp_ratio 48/11
        return StudentConfig(img_size=56, embed_dim=176, depth=3, num_heads=2, mlp_ratio=48 / 11, num_frames=4,
                             attn_pool_num_heads=2, clip_embed_dim=64, clip_teacher_embed_dim=96,
                             clip_teacher_final_dim=64, clip_return_layer=3, mae_teacher_embed_dim=176,
                             mae_return_layer=2)
    if name == "dist64":      # DistInternVideo2 flavour: MLP decoders, explicit return index, no MAE branch
        return StudentConfig(img_size=56, embed_dim=128, depth=3, num_heads=2, mlp_ratio=4.0, num_frames=4,
                             attn_pool_num_heads=2, clip_embed_dim=64, clip_teacher_embed_dim=96,
                             clip_teacher_final_dim=64, clip_return_layer=2, has_mae=False, clip_decoder_kind="mlp",
                             clip_return_index_override=(2, 0))
    if name == "mm88":        # stage-2 vision encoder flavour: hd 88, separate image tables
        return StudentConfig(img_size=56, embed_dim=176, depth=4, num_heads=2, mlp_ratio=48 / 11, num_frames=4,
                             attn_pool_num_heads=2, clip_embed_dim=64, clip_teacher_embed_dim=96,
                             clip_teacher_final_dim=64, clip_return_layer=3, has_mae=False, sep_image_video_pos_embed=True)
    if name == "mm64":        # stage-2 vision encoder flavour: shared tables (image mode averages over frames)
        return StudentConfig(img_size=56, embed_dim=128, depth=3, num_heads=2, mlp_ratio=4.0, num_frames=4,
                             attn_pool_num_heads=2, clip_embed_dim=64, clip_teacher_embed_dim=96,
                             clip_teacher_final_dim=64, clip_return_layer=2, has_mae=False)
    if name == "teach128":    # CLIP-teacher flavour: hd = 128 like InternVL-6B (3200 / 25), per-frame sequences of 17 tokens
        return StudentConfig(img_size=56, embed_dim=256, depth=3, num_heads=2, mlp_ratio=4.0, num_frames=4,
                             attn_pool_num_heads=4, clip_embed_dim=64, clip_return_layer=2, has_mae=False)
    if name == "S14":         # BASELINE configs[0]: ViT-S/14, 4 x 112^2
        return StudentConfig(img_size=112, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4.0, num_frames=4,
                             clip_return_layer=1, mae_return_layer=1)
    if name == "B14":         # configs[1]: ViT-B/14, 8 x 224^2
        return StudentConfig(embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, num_frames=8,
                             clip_teacher_embed_dim=1408, clip_return_layer=6, mae_return_layer=4)
    if name == "1B":          # configs[2]: the headline model
        return StudentConfig(embed_dim=1408, depth=40, num_heads=16, mlp_ratio=48 / 11, num_frames=8,
                             clip_return_layer=6, mae_return_layer=4)
    raise KeyError(name)


# =================================================================================================================================
# Stage-2 text / fusion tower: post-LN BERT with cross-attention to the vision tokens in the layers
# >= fusion_layer, the MLM head, and the VTM / MLM losses.
#   "XB:" = /root/reference/InternVideo2/multi_modality/models/backbones/bert/xbert.py
#   "BB:" = /root/reference/InternVideo2/multi_modality/models/backbones/bert/builder.py
#   "C:"  = /root/reference/InternVideo2/multi_modality/models/criterions.py
#   "S2:" = /root/reference/InternVideo2/multi_modality/models/internvideo2_stage2_visual.py
# Pinned by tests/golden/bert_tiny.npz (tests/golden/make_golden_bert.py runs the reference's own BertForMaskedLM, vtm_loss and MLMLoss
# on CPU; dropout 0 -- the stage-2 configs keep BERT's 0.1 dropout, which is a random mask and has no fixed-point to pin).
# =================================================================================================================================
@dataclass
class BertTowerConfig:
    """configs/config_bert_large.json + BB:18-24 (encoder_width = vision d_model, fusion_layer from the model config)"""
    vocab_size: int = 30522
    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    pad_token_id: int = 0
    cls_token_id: int = 101
    mask_token_id: int = 103
    fusion_layer: int = 19
    encoder_width: int = 1408


def named_bert_config(name: str) -> BertTowerConfig:
    if name == "bert_tiny":       # fixture-sized: hd 64, 4 layers of which the last 2 cross-attend to 176-wide vision tokens (mm88)
        return BertTowerConfig(vocab_size=210, hidden_size=128, num_hidden_layers=4, num_attention_heads=2, intermediate_size=256,
                               max_position_embeddings=40, fusion_layer=2, encoder_width=176, cls_token_id=5, mask_token_id=7)
    if name == "bert_large_1B":   # scripts/pretraining/stage2/1B/config.py: bert_large, fusion_layer 19, vision d_model 1408
        return BertTowerConfig()
    raise KeyError(name)


def bert_param_shapes(cfg: BertTowerConfig) -> Dict[str, Tuple[int, ...]]:
    """state_dict of the reference's BertForMaskedLM (parameters only; `bert.embeddings.position_ids` is an arange buffer, and
    `cls.predictions.decoder.weight` / `.decoder.bias` alias the word embeddings / `cls.predictions.bias`, XB:846-858,1599-1614)."""
    D, I, W = cfg.hidden_size, cfg.intermediate_size, cfg.encoder_width
    s: Dict[str, Tuple[int, ...]] = {
        "bert.embeddings.word_embeddings.weight": (cfg.vocab_size, D),
        "bert.embeddings.position_embeddings.weight": (cfg.max_position_embeddings, D),
        "bert.embeddings.token_type_embeddings.weight": (cfg.type_vocab_size, D),
        "bert.embeddings.LayerNorm.weight": (D,), "bert.embeddings.LayerNorm.bias": (D,),
    }

    def attn(pre: str, kv_in: int):
        s[pre + "self.query.weight"] = (D, D); s[pre + "self.query.bias"] = (D,)
        s[pre + "self.key.weight"] = (D, kv_in); s[pre + "self.key.bias"] = (D,)
        s[pre + "self.value.weight"] = (D, kv_in); s[pre + "self.value.bias"] = (D,)
        s[pre + "output.dense.weight"] = (D, D); s[pre + "output.dense.bias"] = (D,)
        s[pre + "output.LayerNorm.weight"] = (D,); s[pre + "output.LayerNorm.bias"] = (D,)
    for i in range(cfg.num_hidden_layers):
        pre = f"bert.encoder.layer.{i}."
        attn(pre + "attention.", D)
        if i >= cfg.fusion_layer:                                                     # XB:608-610
            attn(pre + "crossattention.", W)                                          # XB:354-356
        s[pre + "intermediate.dense.weight"] = (I, D); s[pre + "intermediate.dense.bias"] = (I,)
        s[pre + "output.dense.weight"] = (D, I); s[pre + "output.dense.bias"] = (D,)
        s[pre + "output.LayerNorm.weight"] = (D,); s[pre + "output.LayerNorm.bias"] = (D,)
    s["cls.predictions.bias"] = (cfg.vocab_size,)
    s["cls.predictions.transform.dense.weight"] = (D, D); s["cls.predictions.transform.dense.bias"] = (D,)
    s["cls.predictions.transform.LayerNorm.weight"] = (D,); s["cls.predictions.transform.LayerNorm.bias"] = (D,)
    return s


def synthetic_bert_params(cfg: BertTowerConfig, seed: int = 0, std: float = 0.05) -> Dict[str, torch.Tensor]:
    """random-init tower: weights N(0, std) (XB:905-907 uses initializer_range 0.02; a wider spread makes every term visible in the
    parity checks), LayerNorm weights around 1, small non-zero biases; the padding row of the word table is zero (nn.Embedding)."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for k, shp in bert_param_shapes(cfg).items():
        if k.endswith("LayerNorm.weight"):
            p[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            p[k] = 0.02 * torch.randn(shp, generator=g)
        else:
            p[k] = std * torch.randn(shp, generator=g)
    p["bert.embeddings.word_embeddings.weight"][cfg.pad_token_id] = 0
    return p


def additive_mask(attention_mask: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """XB:1091-1120 get_extended_attention_mask / invert_attention_mask (encoder, 2-D mask): (1 - mask) * -10000 as [B, 1, 1, L]"""
    return (1.0 - attention_mask[:, None, None, :].to(dtype)) * -10000.0


def bert_embeddings(p: Dict[str, torch.Tensor], ids: torch.Tensor, cfg: BertTowerConfig) -> torch.Tensor:
    """XB:298-334: (word[ids] + token_type[0]) + position[0..L-1] -> LayerNorm (dropout p = 0)"""
    L = ids.shape[1]
    e = p["bert.embeddings.word_embeddings.weight"][ids] + p["bert.embeddings.token_type_embeddings.weight"][torch.zeros_like(ids)]
    e = e + p["bert.embeddings.position_embeddings.weight"][:L][None]
    return layernorm(e, p["bert.embeddings.LayerNorm.weight"], p["bert.embeddings.LayerNorm.bias"], cfg.layer_norm_eps)


def bert_attention(h: torch.Tensor, p: Dict[str, torch.Tensor], pre: str, cfg: BertTowerConfig, mask_add: Optional[torch.Tensor],
                   enc: Optional[torch.Tensor] = None) -> torch.Tensor:
    """BertAttention = BertSelfAttention (XB:390-498) + BertSelfOutput (XB:508-512).  Self: keys / values from h; cross: from the vision
    tokens `enc` (width encoder_width) with the encoder mask.  scores / sqrt(hd) + additive mask -> softmax -> context -> dense ->
    LayerNorm(dense + h)."""
    B, L, D = h.shape
    H = cfg.num_attention_heads
    hd = D // H
    src = h if enc is None else enc
    q = torch.nn.functional.linear(h, p[pre + "self.query.weight"], p[pre + "self.query.bias"]).view(B, L, H, hd).permute(0, 2, 1, 3)
    k = torch.nn.functional.linear(src, p[pre + "self.key.weight"], p[pre + "self.key.bias"]).view(B, -1, H, hd).permute(0, 2, 1, 3)
    v = torch.nn.functional.linear(src, p[pre + "self.value.weight"], p[pre + "self.value.bias"]).view(B, -1, H, hd).permute(0, 2, 1, 3)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)                                   # XB:417, 458
    if mask_add is not None:
        s = s + mask_add                                                                       # XB:459-461
    ctx = torch.matmul(torch.softmax(s, dim=-1), v).permute(0, 2, 1, 3).reshape(B, L, D)       # XB:464-482
    o = torch.nn.functional.linear(ctx, p[pre + "output.dense.weight"], p[pre + "output.dense.bias"])
    return layernorm(o + h, p[pre + "output.LayerNorm.weight"], p[pre + "output.LayerNorm.bias"], cfg.layer_norm_eps)


def bert_layer(h: torch.Tensor, p: Dict[str, torch.Tensor], i: int, cfg: BertTowerConfig, mask_add, enc=None, enc_mask_add=None):
    """BertLayer.forward XB:613-688: self-attention, cross-attention when the layer has one (i >= fusion_layer; the encoder states must
    then be given, XB:637-640), GELU(erf) feed-forward with its own post-LN (XB:579-596)."""
    pre = f"bert.encoder.layer.{i}."
    a = bert_attention(h, p, pre + "attention.", cfg, mask_add)
    if i >= cfg.fusion_layer:
        assert enc is not None, "encoder_hidden_states must be given for cross-attention layers"
        a = bert_attention(a, p, pre + "crossattention.", cfg, enc_mask_add, enc=enc)
    u = gelu(torch.nn.functional.linear(a, p[pre + "intermediate.dense.weight"], p[pre + "intermediate.dense.bias"]), "erf")
    o = torch.nn.functional.linear(u, p[pre + "output.dense.weight"], p[pre + "output.dense.bias"])
    return layernorm(o + a, p[pre + "output.LayerNorm.weight"], p[pre + "output.LayerNorm.bias"], cfg.layer_norm_eps)


def bert_layer_range(cfg: BertTowerConfig, mode: str) -> Tuple[int, int]:
    """XB:721-733"""
    if mode == "text":
        return 0, cfg.fusion_layer
    if mode == "fusion":
        return cfg.fusion_layer, cfg.num_hidden_layers
    if mode == "multi_modal":
        return 0, cfg.num_hidden_layers
    raise ValueError(mode)


def bert_model(p: Dict[str, torch.Tensor], cfg: BertTowerConfig, input_ids: Optional[torch.Tensor] = None,
               attention_mask: Optional[torch.Tensor] = None, encoder_embeds: Optional[torch.Tensor] = None,
               encoder_hidden_states: Optional[torch.Tensor] = None, encoder_attention_mask: Optional[torch.Tensor] = None,
               mode: str = "multi_modal") -> torch.Tensor:
    """BertModel.forward XB:1122-1296 (no pooler: BB:47-54 / BertForMaskedLM XB:1599) -> last_hidden_state [B, L, D].
    `encoder_embeds` (already-embedded text states) replaces the embedding lookup (XB:1258-1267); a missing attention mask is all ones
    (XB:1209-1212); a missing encoder mask is all ones (XB:1241-1245)."""
    if encoder_embeds is None:
        h = bert_embeddings(p, input_ids, cfg)
    else:
        h = encoder_embeds
    B, L = h.shape[:2]
    if attention_mask is None:
        attention_mask = torch.ones(B, L)
    mask_add = additive_mask(attention_mask, h.dtype)
    enc_add = None
    if encoder_hidden_states is not None:
        if encoder_attention_mask is None:
            encoder_attention_mask = torch.ones(encoder_hidden_states.shape[:2])
        enc_add = additive_mask(encoder_attention_mask, h.dtype)
    lo, hi = bert_layer_range(cfg, mode)
    for i in range(lo, hi):
        h = bert_layer(h, p, i, cfg, mask_add, encoder_hidden_states, enc_add)
    return h


def bert_mlm_head(h: torch.Tensor, p: Dict[str, torch.Tensor], cfg: BertTowerConfig) -> torch.Tensor:
    """BertOnlyMLMHead XB:829-874: dense -> GELU(erf) -> LayerNorm -> decoder tied to the word embeddings + output-only bias"""
    t = gelu(torch.nn.functional.linear(h, p["cls.predictions.transform.dense.weight"], p["cls.predictions.transform.dense.bias"]), "erf")
    t = layernorm(t, p["cls.predictions.transform.LayerNorm.weight"], p["cls.predictions.transform.LayerNorm.bias"], cfg.layer_norm_eps)
    return torch.nn.functional.linear(t, p["bert.embeddings.word_embeddings.weight"], p["cls.predictions.bias"])


def mlm_mask_tokens(input_ids: np.ndarray, draw_mask: np.ndarray, draw_replace: np.ndarray, draw_random: np.ndarray,
                    random_words: np.ndarray, cfg: BertTowerConfig) -> Tuple[np.ndarray, np.ndarray]:
    """MLMLoss.mask C:297-342 with the three Bernoulli draws (p = masking_prob, 0.8, 0.5) and the random-word table made explicit:
    never mask [PAD] / [CLS]; labels = -100 off the masked positions; 80 % -> [MASK], half of the rest -> a random word, the rest keep
    their token.  Integer work: bit-exact.  -> (masked input_ids, labels)"""
    ids = input_ids.copy()
    masked = draw_mask.astype(bool).copy()
    masked[input_ids == cfg.pad_token_id] = False
    masked[input_ids == cfg.cls_token_id] = False
    labels = input_ids.copy()
    labels[~masked] = -100
    replaced = draw_replace.astype(bool) & masked
    ids[replaced] = cfg.mask_token_id
    rnd = draw_random.astype(bool) & masked & ~replaced
    ids[rnd] = random_words[rnd]
    return ids, labels


def mlm_loss(p: Dict[str, torch.Tensor], cfg: BertTowerConfig, masked_ids: torch.Tensor, labels: torch.Tensor,
             attention_mask: torch.Tensor, vision_embeds: torch.Tensor) -> torch.Tensor:
    """MLMLoss.mlm_loss C:235-274 after the masking: text-mode pass over the masked ids, fusion-mode pass cross-attending to every
    vision token (vision_atts = None, S2:153-156), MLM head, CrossEntropyLoss with ignore_index -100 (XB:1677-1682)."""
    text = bert_model(p, cfg, input_ids=masked_ids, attention_mask=attention_mask, mode="text")
    fused = bert_model(p, cfg, encoder_embeds=text, attention_mask=attention_mask, encoder_hidden_states=vision_embeds, mode="fusion")
    logits = bert_mlm_head(fused, p, cfg)
    return torch.nn.functional.cross_entropy(logits.view(-1, cfg.vocab_size), labels.view(-1), ignore_index=-100)


def vtm_negative_weights(vision_proj: torch.Tensor, text_proj: torch.Tensor, idx: Optional[torch.Tensor], temp) -> Tuple[torch.Tensor, torch.Tensor]:
    """C:133-146: sampling weights of the hard negatives.  softmax(sim + 1e-4) per row, zero where idx matches (same example; the
    diagonal when idx is None, C:200-216), non-finite entries -> 1e-2.  -> (weights_v2t, weights_t2v)"""
    s_v2t, s_t2v = contrastive_sim(vision_proj, text_proj, temp)
    w_v2t = torch.softmax(s_v2t + 1e-4, dim=1)
    w_t2v = torch.softmax(s_t2v + 1e-4, dim=1)
    if idx is not None:
        same = idx.view(-1, 1) == idx.view(1, -1)
    else:
        same = torch.eye(s_v2t.shape[0], dtype=torch.bool)
    w_v2t = torch.nan_to_num(w_v2t.masked_fill(same, 0), nan=1e-2, posinf=1e-2, neginf=1e-2)
    w_t2v = torch.nan_to_num(w_t2v.masked_fill(same, 0), nan=1e-2, posinf=1e-2, neginf=1e-2)
    return w_v2t, w_t2v


def vtm_loss_given_negatives(p: Dict[str, torch.Tensor], cfg: BertTowerConfig, itm_w: torch.Tensor, itm_b: torch.Tensor,
                             vision_embeds: torch.Tensor, text_embeds: torch.Tensor, text_atts: torch.Tensor,
                             vision_neg: torch.Tensor, text_neg: torch.Tensor) -> torch.Tensor:
    """C:148-181 after the multinomial draws (vision_neg[b] = the video paired with text b as a negative, text_neg[b] = the text paired
    with video b): fusion pass over [pos | (neg video, text) | (video, neg text)] = 3B rows, itm head on the [CLS] state, cross entropy
    with labels [1] * B + [0] * 2B."""
    B = vision_embeds.shape[0]
    v_all = torch.cat([vision_embeds, vision_embeds[vision_neg], vision_embeds], dim=0)
    t_all = torch.cat([text_embeds, text_embeds, text_embeds[text_neg]], dim=0)
    a_all = torch.cat([text_atts, text_atts, text_atts[text_neg]], dim=0)
    out = bert_model(p, cfg, encoder_embeds=t_all, attention_mask=a_all, encoder_hidden_states=v_all, mode="fusion")
    logits = torch.nn.functional.linear(out[:, 0], itm_w, itm_b)
    labels = torch.cat([torch.ones(B, dtype=torch.long), torch.zeros(2 * B, dtype=torch.long)])
    return torch.nn.functional.cross_entropy(logits, labels)


def synthetic_text_batch(cfg: BertTowerConfig, B: int, L: int, seed: int = 0):
    """right-padded token ids [B, L] ([CLS] first, lengths 3..L, no special tokens inside) and their attention mask"""
    rng = np.random.RandomState(seed)
    lens = rng.randint(3, L + 1, size=B)
    lens[0] = L
    ids = np.full((B, L), cfg.pad_token_id, dtype=np.int64)
    mask = np.zeros((B, L), dtype=np.int64)
    lo = max(cfg.cls_token_id, cfg.mask_token_id, cfg.pad_token_id) + 1
    for b in range(B):
        ids[b, 0] = cfg.cls_token_id
        ids[b, 1:lens[b]] = rng.randint(lo, cfg.vocab_size, size=lens[b] - 1)
        mask[b, :lens[b]] = 1
    return ids, mask
