"""MI355X-native mirror of InternVideo2/single_modality/models/internvideo2_pretrain.py ("P:").

Same class names, constructor kwargs, parameter names/shapes (state_dict contract), forward
signature and return tuple as the reference -- `from internvideo_amd.internvideo2_pretrain import
pretrain_internvideo2_1B_patch14_224` is the only line a user of run_pretraining.py changes -- but every forward /
backward FLOP runs in the hand-written gfx950 kernels of csrc/ (no flash_attn / apex / cuDNN, no torch compute op).

Differences from the reference that are visible to a caller:
  * the three `use_flash_attn / use_fused_rmsnorm / use_fused_mlp` flags are accepted and must be consistent
    (P:446-447) but do not select code paths: there is one (fused) path.  `fused_mlp_act` = "erf" (parity with the
    unfused Mlp / the CPU oracle, default) or "tanh" (what flash_attn's FusedMLP computes; use it for checkpoints
    trained on the fused reference path);
  * `use_checkpoint` / `checkpoint_num` (P:294-296, 323-327) are honoured: the first `checkpoint_num` blocks keep only their outputs
    and are recomputed in backward, bit-identically (functional.BlockStackFn).  Off by default: 288 GB of HBM holds all block
    activations at the reference batch size, so the shipped recipes need no recomputation here;
  * compute is bf16 MFMA; the residual stream is fp32 by default (attribute `model.residual_dtype = "fp32"`, the parity setting) or
    bf16 (`model.residual_dtype = "bf16"`: what the reference's own bf16 recipe carries, `residual_in_fp32=False`, P:283-286, 467)
    whatever the parameter dtype; outputs are bf16;
  * forward on CPU tensors raises: there is no CPU path (host-side logic -- construction, state_dict, mask/index
    helpers -- works without a GPU).
"""
from __future__ import annotations

import math
from functools import partial
from typing import List, Optional

import numpy as np
import torch
from torch import nn

from . import functional as Fn
from . import ops
from .lib import InternVideoHipError
from .pos_embed import get_1d_sincos_pos_embed, get_2d_sincos_pos_embed, get_3d_sincos_pos_embed

_registry = {}


def register_model(fn):
    """timm-style registry (the reference registers with timm.models.registry.register_model, P:747,758)."""
    _registry[fn.__name__] = fn
    try:                                       # also register with timm when it is installed
        from timm.models.registry import register_model as _rm
        return _rm(fn)
    except Exception:
        return fn


def create_model(name: str, **kwargs):
    return _registry[name](**kwargs)


def _trunc_normal_(t, std=.02):
    return nn.init.trunc_normal_(t, mean=0., std=std, a=-2., b=2.)          # timm 0.5.4 defaults (absolute bounds)


# ---------------------------------------------------------------------------------------------------------------
# parameter containers with the reference's names
# ---------------------------------------------------------------------------------------------------------------
class RMSNorm(nn.Module):
    """P:117-128 / flash_attn DropoutAddRMSNorm(prenorm=True) P:466-467.  forward(x, residual=None) -> (y, new_residual)
    following the fused protocol P:283-286; standalone use computes through the same kernel."""

    def __init__(self, hidden_size, eps=1e-6, prenorm=True):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps
        self.prenorm = prenorm

    def forward(self, x, residual=None):
        return _RMSNormAddFn.apply(x, residual, self.weight, self.variance_epsilon)


DropoutAddRMSNorm = RMSNorm


class _RMSNormAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, w, eps):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        xb = x2 if x2.dtype == torch.bfloat16 else x2.to(torch.bfloat16)
        r2 = residual.reshape(-1, shp[-1]).float().contiguous() if residual is not None else None
        if r2 is None:
            r_in = x2.float()
            res_out, y, rstd = ops.rmsnorm_add_fwd(r_in, None, None, None, 1, Fn.vec(w), eps, want_res_out=False)
            res_out = r_in
        else:
            res_out, y, rstd = ops.rmsnorm_add_fwd(r2, xb, None, None, 1, Fn.vec(w), eps)
        ctx.save_for_backward(res_out, rstd)
        ctx.w, ctx.has_res, ctx.xdtype = w, residual is not None, x.dtype
        return y.reshape(shp), res_out.reshape(shp)

    @staticmethod
    def backward(ctx, dy, dres):
        res_out, rstd = ctx.saved_tensors
        w = ctx.w
        D = res_out.shape[-1]
        dy2 = dy.reshape(-1, D).contiguous().to(torch.bfloat16)
        dr = dres.reshape(-1, D).float().contiguous().clone() if dres is not None else None
        dres_in, dbranch, dw, _ = ops.rmsnorm_add_bwd(dy2, dr, res_out, rstd, Fn.vec(w), None, None, None, 1, want_dbranch=ctx.has_res)
        if ctx.has_res:
            return dbranch.reshape(dy.shape).to(ctx.xdtype), dres_in.reshape(dy.shape), Fn._ret_grad(w, dw), None
        return dres_in.reshape(dy.shape).to(ctx.xdtype), None, Fn._ret_grad(w, dw), None


class LayerScale(nn.Module):
    """P:131-146.  The multiply is fused into the residual/RMSNorm kernel in fp32 (== force_fp32 semantics)."""

    def __init__(self, dim, init_values=1e-5, inplace=False, force_fp32=False):
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(init_values * torch.ones(dim))
        self.force_fp32 = force_fp32


class Attention(nn.Module):
    """P:149-217 parameter container: qkv (no bias), proj, q_norm / k_norm over the full dim."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0., use_flash_attn=False,
                 causal=False, norm_layer=nn.LayerNorm, qk_normalization=False, use_fused_rmsnorm=False):
        super().__init__()
        assert dim % num_heads == 0, 'dim should be divisible by num_heads'
        if qkv_bias or attn_drop or proj_drop or causal or not qk_normalization:
            raise InternVideoHipError("the MI355X path implements the InternVideo2 configuration: qkv_bias=False, "
                                      "qk_normalization=True, no dropout, non-causal (P:412-418,515)")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        self.proj = nn.Linear(dim, dim)
        self.q_norm = RMSNorm(dim)
        self.k_norm = RMSNorm(dim)


class Mlp(nn.Module):
    """P:220-244 parameter container (fc1, fc2)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, bias=True, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)


FusedMLP = Mlp


class Block(nn.Module):
    """P:247-297 parameter container; the arithmetic of all blocks runs in functional.BlockStackFn."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0., init_values=None,
                 drop_path=0., norm_layer=None, use_flash_attn=False, use_fused_mlp=False, fused_mlp_heuristic=1,
                 with_cp=False, qk_normalization=False, layerscale_no_force_fp32=False, use_fused_rmsnorm=False):
        super().__init__()
        self.norm1 = RMSNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop,
                              qk_normalization=qk_normalization)
        self.ls1 = LayerScale(dim, init_values=init_values, force_fp32=(not layerscale_no_force_fp32)) if init_values else None
        self.norm2 = RMSNorm(dim, eps=1e-6)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), drop=drop)
        self.ls2 = LayerScale(dim, init_values=init_values, force_fp32=(not layerscale_no_force_fp32)) if init_values else None
        self.drop_path = float(drop_path)
        self.with_cp = with_cp

    def flat_params(self):
        return [self.norm1.weight, self.attn.qkv.weight, self.attn.q_norm.weight, self.attn.k_norm.weight,
                self.attn.proj.weight, self.attn.proj.bias, self.ls1.gamma if self.ls1 is not None else None,
                self.norm2.weight, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias,
                self.ls2.gamma if self.ls2 is not None else None]


class PatchEmbed(nn.Module):
    """P:300-331: Conv3d parameters (weight (D,3,t,p,p), bias); arithmetic in functional.PatchEmbedGatherFn."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, num_frames=8, tubelet_size=1, norm_layer=None):
        super().__init__()
        img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.img_size, self.patch_size, self.tubelet_size = img_size, patch_size, tubelet_size
        self.grid_size = (num_frames // tubelet_size, img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1] * self.grid_size[2]
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=(tubelet_size, patch_size[0], patch_size[1]),
                              stride=(tubelet_size, patch_size[0], patch_size[1]))
        self.norm = nn.Identity()


class CrossAttention(nn.Module):
    """P:18-80 parameter container."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, out_dim=None):
        super().__init__()
        out_dim = out_dim or dim
        self.num_heads = num_heads
        self.q = nn.Linear(dim, dim, bias=False)
        self.k = nn.Linear(dim, dim, bias=False)
        self.v = nn.Linear(dim, dim, bias=False)
        self.q_bias = nn.Parameter(torch.zeros(dim))
        self.k_bias = nn.Parameter(torch.zeros(dim))
        self.v_bias = nn.Parameter(torch.zeros(dim))
        self.proj = nn.Linear(dim, out_dim)


class AttentionPoolingBlock(nn.Module):
    """P:83-114."""

    def __init__(self, dim, num_heads, qkv_bias=True, qk_scale=None, drop=0., attn_drop=0., norm_layer=None, out_dim=None):
        super().__init__()
        ln = norm_layer or partial(nn.LayerNorm, eps=1e-5)
        self.norm1_q, self.norm1_k, self.norm1_v = ln(dim), ln(dim), ln(dim)
        self.cross_attn = CrossAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, out_dim=out_dim)
        self.num_heads = num_heads

    def forward(self, x, B, L):
        """x: fp32 residual-stream rows [B*L, D] -> [B, out_dim] bf16"""
        ca = self.cross_attn
        return Fn.AttnPoolFn.apply(x, B, L, self.num_heads, self.norm1_q.eps,
                                   self.norm1_q.weight, self.norm1_q.bias, self.norm1_k.weight, self.norm1_k.bias,
                                   self.norm1_v.weight, self.norm1_v.bias, ca.q.weight, ca.q_bias, ca.k.weight, ca.k_bias,
                                   ca.v.weight, ca.v_bias, ca.proj.weight, ca.proj.bias)


class Linear_Decoder(nn.Module):
    """P:334-365."""

    def __init__(self, in_channels=1408, out_channels=3200, norm_layer=nn.LayerNorm, norm_type='l2'):
        super().__init__()
        if norm_type not in ('l2', 'none'):                    # P:358-363: anything else raises in the reference's forward
            raise NotImplementedError(f"norm_type {norm_type!r}: the reference implements 'l2' and 'none'")
        self.norm_type = norm_type
        self.head = nn.Linear(in_channels, out_channels)
        self.norm = norm_layer(out_channels)

    def forward(self, x):
        """standalone use (final_clip_decoder, P:720): x bf16 [..., in] -> bf16 [..., out], l2-normalised unless norm_type 'none'"""
        y = Fn.LinearFn.apply(x, self.head.weight, self.head.bias)
        return Fn.LnL2Fn.apply(y, self.norm.weight, self.norm.bias, self.norm.eps, None, self.norm_type == 'none')


class MLP_Decoder(nn.Module):
    """P:368-403."""

    def __init__(self, in_channels=768, out_channels=768, norm_layer=nn.LayerNorm, norm_type='l2'):
        super().__init__()
        if norm_type not in ('l2', 'none'):                    # P:396-401
            raise NotImplementedError(f"norm_type {norm_type!r}: the reference implements 'l2' and 'none'")
        self.norm_type = norm_type
        self.head = nn.Sequential(nn.Linear(in_channels, in_channels), nn.GELU(), nn.Linear(in_channels, out_channels))
        self.norm = norm_layer(out_channels)

    def forward(self, x):
        y = Fn.MlpFn.apply(x, self.head[0].weight, self.head[0].bias, self.head[2].weight, self.head[2].bias, "gelu_erf")
        return Fn.LnL2Fn.apply(y, self.norm.weight, self.norm.bias, self.norm.eps, None, self.norm_type == 'none')


# ---------------------------------------------------------------------------------------------------------------
def build_gather_indices(mask: torch.Tensor, device, L: Optional[int] = None, check: bool = True) -> tuple:
    """mask (B, 1+N) bool/uint8, True = masked (P:659).  -> (vis_idx int32 [B,L], inv_idx int32 [B,1+N]) on `device`.
    A CPU mask (what engines/engine_for_pretraining.py:110-116 builds) is compacted on the host, bit-exactly like
    `x[~mask]`, and raises on ragged rows like the reference's reshape; a device mask is compacted by the HIP kernel."""
    if mask.dim() != 2:
        raise ValueError("mask must be (B, 1+N)")
    if mask.dtype not in (torch.bool, torch.uint8):
        mask = mask.to(torch.bool)
    if not mask.is_cuda:
        m = mask.to(torch.bool).numpy()
        keep = ~m
        cnt = keep.sum(1)
        if not (cnt == cnt[0]).all():
            raise RuntimeError(f"mask keeps a different number of tokens per clip ({cnt.tolist()}): x[~mask].reshape(B,-1,C) is ill-defined")
        L = int(cnt[0])
        vis = np.nonzero(keep)[1].reshape(m.shape[0], L).astype(np.int32)
        inv = np.full(m.shape, -1, dtype=np.int32)
        np.put_along_axis(inv, vis, np.broadcast_to(np.arange(L, dtype=np.int32), vis.shape), axis=1)
        return (torch.from_numpy(vis).to(device, non_blocking=True), torch.from_numpy(inv).to(device, non_blocking=True))
    if L is None:                                   # one host sync to learn the kept count (pass L to avoid it)
        L = int((~mask[0].to(torch.bool)).sum().item())
    vis, inv, cnt = ops.mask_to_indices(mask, L)
    if check and not bool((cnt == L).all().item()):
        raise RuntimeError("mask keeps a different number of tokens per clip: x[~mask].reshape(B,-1,C) is ill-defined")
    return vis, inv


def full_gather_indices(B: int, N1: int, device) -> tuple:
    """`mask=None` (multi_modality/models/backbones/internvideo2/internvideo2.py:611-614): every token is kept."""
    idx = torch.arange(N1, dtype=torch.int32, device=device).unsqueeze(0).expand(B, N1).contiguous()
    return idx, idx


def _residual_is_bf16(v) -> bool:
    if v in ("bf16", "bfloat16", torch.bfloat16):
        return True
    if v in ("fp32", "float32", torch.float32, None):
        return False
    raise ValueError(f"residual_dtype must be 'fp32' or 'bf16', got {v!r}")


class PretrainInternVideo2(nn.Module):
    """P:406-744."""

    def __init__(
            self, in_chans: int = 3, patch_size: int = 14, img_size: int = 224, qkv_bias: bool = False,
            drop_path_rate: float = 0.25, embed_dim: int = 1408, num_heads: int = 16, mlp_ratio: float = 4.3637,
            init_values: float = 1e-5, qk_normalization: bool = True, depth: int = 40,
            use_flash_attn: bool = True, use_fused_rmsnorm: bool = True, use_fused_mlp: bool = True,
            fused_mlp_heuristic: int = 1, attn_pool_num_heads: int = 16, clip_embed_dim: int = 768,
            layerscale_no_force_fp32: bool = False, num_frames: int = 8, tubelet_size: int = 1,
            sep_pos_embed: bool = False, use_checkpoint: bool = False, checkpoint_num: int = 0,
            clip_teacher_embed_dim: int = 3200, clip_teacher_final_dim: int = 768, clip_norm_type: str = 'l2',
            clip_return_layer: int = 1, clip_student_return_interval: int = 1,
            mae_teacher_embed_dim: int = 1408, mae_norm_type: str = 'l2', mae_return_layer: int = 1,
            mae_student_return_interval: int = 1, fused_mlp_act: str = "erf", verbose: bool = False,
    ):
        super().__init__()
        assert use_flash_attn == use_fused_rmsnorm == use_fused_mlp, \
            'use_flash_attn, use_fused_rmsnorm and use_fused_mlp should be consistent'
        self.use_flash_attn = use_flash_attn
        self.embed_dim, self.depth, self.num_heads = embed_dim, depth, num_heads
        self.fused_mlp_act = {"erf": "gelu_erf", "tanh": "gelu_tanh"}[fused_mlp_act]
        self.clip_norm_type, self.mae_norm_type = clip_norm_type, mae_norm_type
        self.clip_return_index = [depth - int(i * clip_student_return_interval) - 1 for i in range(clip_return_layer)]
        self.mae_return_index = [depth - int(i * mae_student_return_interval) - 1 for i in range(mae_return_layer)]
        self.norm_layer_for_blocks = partial(RMSNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, num_frames=num_frames, tubelet_size=tubelet_size)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.sep_pos_embed = bool(sep_pos_embed)
        if self.sep_pos_embed:                                 # P:479-495: spatial + temporal (+ cls) tables, joined on the fly in forward
            grid = self.patch_embed.grid_size
            self.grid_size = grid
            for pre in ("", "clip_", "mae_"):
                setattr(self, pre + "pos_embed_spatial", nn.Parameter(torch.zeros(1, grid[1] * grid[2], embed_dim)))
                setattr(self, pre + "pos_embed_temporal", nn.Parameter(torch.zeros(1, grid[0], embed_dim)))
                if pre != "mae_":
                    setattr(self, pre + "pos_embed_cls", nn.Parameter(torch.zeros(1, 1, embed_dim)))
        else:
            self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
            self.clip_pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
            self.mae_pos_embed = nn.Parameter(torch.zeros(1, num_patches, embed_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth, device="cpu")]
        self.drop_path_rates = dpr
        with_cp_list = [use_checkpoint and idx < checkpoint_num for idx in range(depth)]
        self.blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias=qkv_bias, drop_path=dpr[i], init_values=init_values, attn_drop=0.,
                  use_flash_attn=use_flash_attn, use_fused_mlp=use_fused_mlp, fused_mlp_heuristic=fused_mlp_heuristic,
                  with_cp=with_cp_list[i], qk_normalization=qk_normalization,
                  layerscale_no_force_fp32=layerscale_no_force_fp32, use_fused_rmsnorm=use_fused_rmsnorm)
            for i in range(depth)])
        self.clip_projector = AttentionPoolingBlock(dim=embed_dim, num_heads=attn_pool_num_heads, qkv_bias=True,
                                                    norm_layer=partial(nn.LayerNorm, eps=1e-5), out_dim=clip_embed_dim)
        self.clip_decoder = nn.ModuleList([
            Linear_Decoder(in_channels=embed_dim, out_channels=clip_teacher_embed_dim,
                           norm_layer=partial(nn.LayerNorm, eps=1e-5), norm_type=clip_norm_type)
            for _ in range(clip_return_layer)])
        self.final_clip_decoder = nn.Identity()
        if clip_teacher_final_dim > 0:
            self.final_clip_decoder = Linear_Decoder(in_channels=clip_embed_dim, out_channels=clip_teacher_final_dim,
                                                     norm_layer=partial(nn.LayerNorm, eps=1e-5), norm_type=clip_norm_type)
        self.mae_decoder = nn.ModuleList([
            MLP_Decoder(in_channels=embed_dim, out_channels=mae_teacher_embed_dim,
                        norm_layer=partial(nn.LayerNorm, eps=1e-5), norm_type=mae_norm_type)
            for _ in range(mae_return_layer)])
        self.init_pos_embed()
        _trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)
        self.fix_init_weight()
        self.grad_ready_hook = None          # set by the training engine: called with the block index during backward
        # BASELINE configs[4] (the 6B encoder on fp8 MFMA): True routes the four GEMMs of every block (forward, dgrad, wgrad) through
        # per-tensor-scaled e4m3 operands (gemm_fp8.hip); norms, attention, residual stream and optimizer state keep their precisions.
        # The reference has no such switch: it is an attribute, not a constructor argument, so the constructor signature stays P:296-320.
        self.fp8_gemm = False
        # "current": every fp8 quantisation takes its scale from the tensor's own max|x| (two passes over it); "delayed": from the amax the
        # same call site saw on the last steps (functional.Fp8History: one pass, saturating if the range grew)
        self.fp8_scaling = "current"
        # "tensor": one scale per weight matrix; "channel": one scale per output feature for the forward GEMM and one per input feature for
        # the transposed copy the dgrad GEMM reads (ops.fp8_quantize_weight) -- activations and gradients stay per-tensor either way
        self.fp8_weight_scales = "tensor"
        # Type of the residual stream between the blocks.  "fp32" (default) is the parity setting: block outputs are compared with the
        # reference's fp32 CPU forward.  "bf16" is what the reference's bf16 recipe itself carries (DropoutAddRMSNorm(prenorm=True) with
        # residual_in_fp32 left False, P:283-286, 467; `model.bfloat16()` on the unfused path): the residual kernels then move 8 instead
        # of 12 bytes per element forward and 10 instead of 16 backward.  An attribute for the same reason as fp8_gemm.
        self.residual_dtype = "fp32"

    # ---- initialisation (P:560-603) -------------------------------------------------------------------------
    def init_pos_embed(self):
        if self.sep_pos_embed:                                 # P:562-577 (the cls tables stay zero)
            D = self.pos_embed_spatial.shape[-1]
            sp = torch.from_numpy(get_2d_sincos_pos_embed(D, self.patch_embed.grid_size[1])).float().unsqueeze(0)
            tm = torch.from_numpy(get_1d_sincos_pos_embed(D, self.patch_embed.grid_size[0])).float().unsqueeze(0)
            for pre in ("", "clip_", "mae_"):
                getattr(self, pre + "pos_embed_spatial").data.copy_(sp)
                getattr(self, pre + "pos_embed_temporal").data.copy_(tm)
            return
        pe = get_3d_sincos_pos_embed(self.pos_embed.shape[-1], self.patch_embed.grid_size[1], self.patch_embed.grid_size[0], cls_token=True)
        self.pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        self.clip_pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        self.mae_pos_embed.data.copy_(torch.from_numpy(pe[1:]).float().unsqueeze(0))

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def fix_init_weight(self):
        for layer_id, layer in enumerate(self.blocks):
            layer.attn.proj.weight.data.div_(math.sqrt(2.0 * (layer_id + 1)))
            layer.mlp.fc2.weight.data.div_(math.sqrt(2.0 * (layer_id + 1)))

    @property
    def dtype(self):
        return self.patch_embed.proj.weight.dtype

    def get_num_layers(self):
        return len(self.blocks)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'pos_embed_spatial', 'pos_embed_temporal', 'pos_embed_cls', 'cls_token',
                'clip_pos_embed', 'clip_pos_embed_spatial', 'clip_pos_embed_temporal', 'clip_pos_embed_cls',
                'mae_pos_embed', 'mae_pos_embed_spatial', 'mae_pos_embed_temporal'}

    # ---- forward --------------------------------------------------------------------------------------------------
    def _pos_table(self, which: str):
        """the (1, [1 +] N, D) positional table `which` in ("", "clip_", "mae_"): the joint parameter, or -- sep_pos_embed (P:639-655,
        696-712, 726-734) -- spatial.repeat(T) + temporal.repeat_interleave(H W) behind the cls row, composed with torch ops so that
        autograd carries the table's gradient back to the separable parameters"""
        if not self.sep_pos_embed:
            return getattr(self, which + "pos_embed")
        sp, tm = getattr(self, which + "pos_embed_spatial"), getattr(self, which + "pos_embed_temporal")
        pos = sp.repeat(1, self.grid_size[0], 1) + torch.repeat_interleave(tm, self.grid_size[1] * self.grid_size[2], dim=1)
        if which != "mae_":
            pos = torch.cat([getattr(self, which + "pos_embed_cls").expand(pos.shape[0], -1, -1), pos], 1)
        return pos

    def _fp8_history(self, device):
        """the amax history of the delayed-scaling fp8 path (None unless fp8_gemm and fp8_scaling == "delayed" while training)"""
        if not (getattr(self, "fp8_gemm", False) and getattr(self, "fp8_scaling", "current") == "delayed" and self.training):
            return None
        h = getattr(self, "_fp8_hist", None)
        if h is None or h.cur.device != torch.device(device):
            h = Fn.Fp8History(device)
            self._fp8_hist = h
        return h

    def _drop_path_scales(self, B, device):
        """per-(block, branch, sample) keep/(1-p) factors: timm DropPath `x.div(keep) * floor(keep + U)` (P:264,274)."""
        if not self.training or max(self.drop_path_rates) == 0.0:
            return None
        keep = getattr(self, "_dp_keep", None)           # cached on the device: no host-to-device copy per step (graph capture)
        if keep is None or keep.device != torch.device(device):
            keep = 1.0 - torch.tensor(self.drop_path_rates, dtype=torch.float32, device=device).view(-1, 1, 1)
            self._dp_keep = keep
        u = getattr(self, "_dp_uniform", None)           # tests: the uniform draws of a reference run, (depth, 2, B) -- instead of fresh ones
        if u is None:
            u = torch.rand((self.depth, 2, B), dtype=torch.float32, device=device)
        else:
            u = u.to(device=device, dtype=torch.float32)
        return (torch.floor(keep + u) / keep).contiguous()

    def forward_features(self, x, mask, vis_inv=None, pos_embed=None, n_blocks=None, extra_taps=(), bf16_taps=False):
        """-> (taps dict {block index: fp32 [B*L, D] residual-stream value}, vis_idx, inv_idx, B, L)
        mask None keeps every token; `pos_embed` overrides self.pos_embed (image mode of the stage-2 encoder); `n_blocks` runs
        only the first n blocks (x_vis_return_idx of the stage-2 encoder) -- the last one run is always tapped."""
        if not x.is_cuda:
            raise InternVideoHipError("PretrainInternVideo2.forward needs HBM-resident inputs: there is no CPU path")
        if x.dim() != 5:
            raise ValueError(f"expected a (B, C, T, H, W) clip tensor, got {tuple(x.shape)}")
        B = x.shape[0]
        pe = self.patch_embed
        if vis_inv is not None:
            vis_idx, inv_idx = vis_inv
        elif mask is None:
            n_tok = (x.shape[2] // pe.tubelet_size) * (x.shape[3] // pe.patch_size[0]) * (x.shape[4] // pe.patch_size[1])
            vis_idx, inv_idx = full_gather_indices(B, n_tok + 1, x.device)
        else:
            # `static_visible_tokens` (attribute, optional): the kept-token count of every clip, known to the caller (fixed mask ratio) --
            # skips the two host reads of build_gather_indices so that the forward can be captured into a HIP graph
            Ls = getattr(self, "static_visible_tokens", None)
            vis_idx, inv_idx = build_gather_indices(mask, x.device, L=Ls, check=Ls is None)
        L = vis_idx.shape[1]
        pos = self._pos_table("") if pos_embed is None else pos_embed
        if inv_idx.shape[1] != pos.shape[-2]:
            raise ValueError(f"mask / clip describe {inv_idx.shape[1] - 1} tokens but the positional table has {pos.shape[-2] - 1}")
        x0 = Fn.PatchEmbedGatherFn.apply(x, vis_idx, inv_idx, pe.proj.weight, pe.proj.bias, self.cls_token, pos,
                                         pe.tubelet_size, pe.patch_size[0])
        n_run = self.depth if n_blocks is None else int(n_blocks)
        if not 1 <= n_run <= self.depth:
            raise ValueError(f"n_blocks={n_blocks} outside 1..{self.depth}")
        taps = sorted({t for t in (set(self.clip_return_index) | set(self.mae_return_index) | set(extra_taps)) if t < n_run} | {n_run - 1})
        n_cp = 0                                             # leading blocks built with with_cp (use_checkpoint / checkpoint_num, P:323)
        for blk in self.blocks[:n_run]:
            if not getattr(blk, "with_cp", False):
                break
            n_cp += 1
        meta = dict(B=B, L=L, H=self.num_heads, eps=1e-6, act=self.fused_mlp_act, taps=taps, grad_ready_hook=self.grad_ready_hook,
                    checkpoint_num=n_cp if torch.is_grad_enabled() else 0, fp8=bool(getattr(self, "fp8_gemm", False)),
                    fp8_hist=self._fp8_history(x0.device), fp8_wchan=(getattr(self, "fp8_weight_scales", "tensor") == "channel"),
                    res_bf16=_residual_is_bf16(getattr(self, "residual_dtype", "fp32")),
                    taps_bf16=bool(bf16_taps),           # callers whose tap consumers read bf16 rows (the decoders, the attention pool)
                    # DropPath skipping: "auto" (default) = the dropped (sample, branch) pairs are not computed wherever the kernels with
                    # device-side counts apply and the problem is large enough to gain; True = wherever they apply; False = never
                    dp_skip=getattr(self, "drop_path_skip", "auto"), dp_count_acc=getattr(self, "dp_count_acc", None))
        params = [p for blk in self.blocks[:n_run] for p in blk.flat_params()]
        outs = Fn.BlockStackFn.apply(x0, self._drop_path_scales(B, x.device), meta, *params)
        return dict(zip(taps, outs)), vis_idx, inv_idx, B, L

    def _clip_branch(self, taps, vis_idx, inv_idx, clip_pos_embed=None, targets=None):
        """CLIP branch (P:700-719): tap + clip_pos_embed[~mask] -> decoder k (ascending block order, P:669-675).
        targets None -> stacked l2-normalised features (K,B,L,Cc); else sum over decoders of sum_rows(2 - 2<s,t>) (1-element fp32)."""
        pos = self._pos_table("clip_") if clip_pos_embed is None else clip_pos_embed
        none_ = self.clip_norm_type == 'none'
        outs = []
        for k, (t, dec) in enumerate(zip(sorted(i for i in self.clip_return_index if i in taps), self.clip_decoder)):
            tg = None if targets is None else targets[k]
            if isinstance(dec, MLP_Decoder):
                outs.append(Fn.PosDecoderFn.apply(taps[t], pos, vis_idx, inv_idx, 0, dec.norm.eps, 1 + 2 * none_, tg,
                                                  dec.head[0].weight, dec.head[0].bias, dec.head[2].weight, dec.head[2].bias,
                                                  dec.norm.weight, dec.norm.bias))
            else:
                outs.append(Fn.PosDecoderFn.apply(taps[t], pos, vis_idx, inv_idx, 0, dec.norm.eps, 2 * none_, tg,
                                                  dec.head.weight, dec.head.bias, dec.norm.weight, dec.norm.bias))
        return torch.stack(outs) if targets is None else sum(outs)

    def _final_branch(self, pooled, target=None):
        """final_clip_decoder (P:720) on the pooled token; with target: sum_rows(2 - 2<s,t>) fused (1-element fp32)."""
        fd = self.final_clip_decoder
        if isinstance(fd, nn.Identity):
            if target is not None:
                raise InternVideoHipError("clip_teacher_final_dim=0: there is no final decoder to distill")
            return pooled
        if isinstance(fd, MLP_Decoder):
            y = Fn.MlpFn.apply(pooled, fd.head[0].weight, fd.head[0].bias, fd.head[2].weight, fd.head[2].bias, "gelu_erf")
        else:
            y = Fn.LinearFn.apply(pooled, fd.head.weight, fd.head.bias)
        return Fn.LnL2Fn.apply(y, fd.norm.weight, fd.norm.bias, fd.norm.eps, target, fd.norm_type == 'none')

    def forward(self, x, mask):
        taps, vis_idx, inv_idx, B, L = self.forward_features(x, mask, bf16_taps=True)
        x_final = taps[self.depth - 1]
        pooled = self.clip_projector(x_final, B, L)                                             # P:690
        x_clip_align = self._clip_branch(taps, vis_idx, inv_idx)                                 # P:700-719
        x_align = self._final_branch(pooled)                                                     # P:720
        mae_pos = self._pos_table("mae_")
        x_mae_align = torch.stack([
            Fn.PosDecoderFn.apply(taps[t], mae_pos, vis_idx, inv_idx, 1, dec.norm.eps, 1 + 2 * (self.mae_norm_type == 'none'), None,
                                  dec.head[0].weight, dec.head[0].bias, dec.head[2].weight, dec.head[2].bias,
                                  dec.norm.weight, dec.norm.bias)
            for t, dec in zip(sorted(self.mae_return_index), self.mae_decoder)])
        return x_clip_align, x_align, x_mae_align

    def forward_loss(self, x, mask, targets, clip_loss_ratio=(1.0, 1.0), mae_loss_ratio=1.0, vis_inv=None):
        """Student forward + the distillation loss of engines/engine_for_pretraining.py:131-148 with the decoder tails
        (LayerNorm, l2, <s,t>, mean) fused: the (K,B,L,3200) student features are never written to HBM.
        targets = (clip_middle (K,B,L,Cc), clip_final (B,Cf), mae (K',B,L-1,Cm)), l2-normalised, bf16 or fp32.
        -> (loss, (loss_clip_middle, loss_clip_final, loss_mae)) as fp32 device scalars."""
        tg_clip, tg_final, tg_mae = targets
        taps, vis_idx, inv_idx, B, L = self.forward_features(x, mask, vis_inv, bf16_taps=True)
        pooled = self.clip_projector(taps[self.depth - 1], B, L)
        n_clip = float(tg_clip.shape[0] * B * L)
        n_mae = float(tg_mae.shape[0] * B * (L - 1))
        l_clip = self._clip_branch(taps, vis_idx, inv_idx, targets=tg_clip) / n_clip
        if tg_final is not None and clip_loss_ratio[1] > 0 and not isinstance(self.final_clip_decoder, nn.Identity):
            l_final = self._final_branch(pooled, tg_final) / float(B)
        else:                                                  # engine_for_pretraining.py:135-138: zeros when the final feature is not distilled
            l_final = torch.zeros(1, dtype=torch.float32, device=l_clip.device)     # (clip_teacher_final_dim = 0 / ratio 0 / no target)
        mae_pos = self._pos_table("mae_")
        l_mae = sum(
            Fn.PosDecoderFn.apply(taps[t], mae_pos, vis_idx, inv_idx, 1, dec.norm.eps, 1 + 2 * (self.mae_norm_type == 'none'), tg_mae[k],
                                  dec.head[0].weight, dec.head[0].bias, dec.head[2].weight, dec.head[2].bias,
                                  dec.norm.weight, dec.norm.bias)
            for k, (t, dec) in enumerate(zip(sorted(self.mae_return_index), self.mae_decoder))) / n_mae
        loss = l_clip * clip_loss_ratio[0] + l_final * clip_loss_ratio[1] + l_mae * mae_loss_ratio
        return loss.reshape(()), (l_clip.reshape(()), l_final.reshape(()), l_mae.reshape(()))


@register_model
def pretrain_internvideo2_1B_patch14_224(pretrained=False, **kwargs):
    """P:747-755"""
    return PretrainInternVideo2(img_size=224, patch_size=14, embed_dim=1408, depth=40, num_heads=16, mlp_ratio=48 / 11,
                                attn_pool_num_heads=16, clip_embed_dim=768, **kwargs)


@register_model
def pretrain_internvideo2_6B_patch14_224(pretrained=False, **kwargs):
    """P:758-766"""
    return PretrainInternVideo2(img_size=224, patch_size=14, embed_dim=3200, depth=48, num_heads=25, mlp_ratio=4,
                                attn_pool_num_heads=16, clip_embed_dim=768, **kwargs)
