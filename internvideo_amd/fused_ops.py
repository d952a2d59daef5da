"""The fused-op seam: drop-in `nn.Module`s for the three flash_attn pieces the reference's model files import,

    from flash_attn.modules.mlp import FusedMLP                  (models/internvideo2_pretrain.py:13, used :268-269)
    from flash_attn.ops.rms_norm import DropoutAddRMSNorm        (:14, used :466-467, :200-202, :283-286)
    from .flash_attention_class import FlashAttention            (:11, used :166, :208-210)

with the same constructor / forward contracts, backed by the gfx950 kernels.  A maintainer who wants to keep the reference's own
`Block` / `Attention` Python and only swap the fused ops imports these three names from here; the whole-model
mirrors in this package (internvideo2_pretrain.py ...) fuse further (residual protocol, LayerScale, DropPath inside the norm kernels).

Each module is an autograd node: forward and backward run in libinternvideo_hip.so; CPU tensors raise (no fallback).
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import nn

from . import functional as Fn
from . import torch_ops  # noqa: F401  (registers torch.ops.internvideo_hip.*: the three modules below dispatch through it)
from .internvideo2_pretrain import RMSNorm
from .lib import InternVideoHipError

_K = torch.ops.internvideo_hip

BF16 = torch.bfloat16


class _FlashQKVPackedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, scale, kv_len, pad):
        """kv_len int32 [B] | None; pad bool (B, S) | None = True at padded positions (their output rows are zeros, like pad_input's)"""
        B, S, three, H, hd = qkv.shape
        q2 = qkv.reshape(B * S, 3 * H * hd)
        if q2.dtype != BF16:
            q2 = q2.to(BF16)
        q2 = q2.contiguous()
        out, lse = _K.flash_attn_fwd(q2, B, S, H, scale, kv_len)
        if pad is not None:
            out.view(B, S, H * hd).masked_fill_(pad.unsqueeze(-1), 0)
        ctx.save_for_backward(q2, out, lse, kv_len, pad)
        ctx.meta = (B, S, H, hd, scale, qkv.dtype)
        return out.view(B, S, H, hd).to(qkv.dtype)

    @staticmethod
    def backward(ctx, dout):
        q2, out, lse, kv_len, pad = ctx.saved_tensors
        B, S, H, hd, scale, dt = ctx.meta
        do = dout.reshape(B * S, H * hd).to(BF16).contiguous()
        if pad is not None:                                    # pad_input's backward drops the gradient of padded rows
            do = do.view(B, S, H * hd).masked_fill(pad.unsqueeze(-1), 0).view(B * S, H * hd)
        dqkv = _K.flash_attn_bwd(q2, out, do, lse, B, S, H, scale, kv_len)
        if pad is not None:                                    # padded tokens receive no gradient at all (unpad_input's backward)
            dqkv.view(B, S, 3 * H * hd).masked_fill_(pad.unsqueeze(-1), 0)
        return dqkv.view(B, S, 3, H, hd).to(dt), None, None, None


class FlashAttention(nn.Module):
    """models/flash_attention_class.py:10-70: `forward(qkv (B,S,3,H,hd) bf16|fp16 on the GPU, key_padding_mask=None, causal=False,
    cu_seqlens=None, max_s=None, need_weights=False) -> (out (B,S,H,hd), None)`.  Equal-length, non-causal, dropout-free attention
    is what every InternVideo2 vision tower uses (attn_drop 0, P:515).  `key_padding_mask` (B, S) bool, True = keep (flash_attn's
    convention, :51-62) is supported for RIGHT-padded batches -- each row a prefix of ones, as tokenised text is: keys beyond a
    sequence's length are excluded in the kernel and padded positions get zero outputs / zero gradients, which is what the
    reference's unpad -> varlen kernel -> pad round trip produces.  Masks with holes, the pre-unpadded `cu_seqlens` form and causal
    attention raise.  fp16 inputs are computed in bf16 (gfx950 MFMA path) and cast back."""

    def __init__(self, softmax_scale=None, attention_dropout=0.0, device=None, dtype=None):
        super().__init__()
        self.softmax_scale = softmax_scale
        self.dropout_p = attention_dropout

    def forward(self, qkv, key_padding_mask=None, causal=False, cu_seqlens=None, max_s=None, need_weights=False):
        assert not need_weights
        assert qkv.dtype in [torch.float16, torch.bfloat16]
        assert qkv.is_cuda
        if cu_seqlens is not None or causal:
            raise InternVideoHipError("FlashAttention (MI355X): the pre-unpadded (cu_seqlens) form and causal attention are not implemented "
                                      "(the InternVideo2 vision towers never use them)")
        if self.training and self.dropout_p:
            raise InternVideoHipError("FlashAttention (MI355X): attention dropout is not implemented (InternVideo2 uses 0)")
        if qkv.dim() != 5 or qkv.shape[2] != 3:
            raise InternVideoHipError(f"qkv must be (B, S, 3, H, D), got {tuple(qkv.shape)}")
        kv_len = pad = None
        if key_padding_mask is not None:
            keep = key_padding_mask.to(torch.bool)
            if tuple(keep.shape) != tuple(qkv.shape[:2]):
                raise InternVideoHipError(f"key_padding_mask must be (B, S) = {tuple(qkv.shape[:2])}")
            kv_len = keep.sum(1, dtype=torch.int32)
            prefix = torch.arange(keep.shape[1], device=keep.device).unsqueeze(0) < kv_len.unsqueeze(1)
            if not bool(torch.equal(keep, prefix)):                        # one host read, like the reference's unpad_input (nonzero)
                raise InternVideoHipError("FlashAttention (MI355X): key_padding_mask must be right-padded (a prefix of ones per row)")
            if bool((kv_len == 0).any()):
                raise InternVideoHipError("FlashAttention (MI355X): every sequence needs at least one valid token")
            pad = ~keep
        return _FlashQKVPackedFn.apply(qkv, self.softmax_scale, kv_len, pad), None


class FusedMLP(nn.Module):
    """flash_attn.modules.mlp.FusedMLP as the reference uses it (P:268-269: `FusedMLP(in_features, hidden_features, heuristic)`):
    fc2(gelu(fc1(x))) with bias, `.fc1` / `.fc2` Linear parameters (the checkpoint keys `mlp.fc1.*`, `mlp.fc2.*`).
    activation: 'gelu_approx' (tanh; flash_attn's default and what the released checkpoints were trained with) or 'gelu' (erf,
    == the unfused `Mlp`, P:220-244).  fc1's epilogue applies the GELU, fc2's dgrad epilogue multiplies by gelu'."""

    def __init__(self, in_features, hidden_features=None, out_features=None, bias1=True, bias2=True, activation='gelu_approx',
                 return_residual=False, checkpoint_lvl=0, heuristic='auto', device=None, dtype=None):
        super().__init__()
        if activation not in ('gelu_approx', 'gelu'):
            raise InternVideoHipError("FusedMLP (MI355X): activation must be 'gelu_approx' or 'gelu'")
        if not (bias1 and bias2) or return_residual:
            raise InternVideoHipError("FusedMLP (MI355X): bias-free / return_residual variants are not used by InternVideo2")
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features * 4
        self.activation = activation
        self.fc1 = nn.Linear(in_features, hidden_features, device=device, dtype=dtype)
        self.fc2 = nn.Linear(hidden_features, out_features, device=device, dtype=dtype)

    def forward(self, x):
        act = "gelu_tanh" if self.activation == 'gelu_approx' else "gelu_erf"
        if not x.is_cuda:
            raise InternVideoHipError("FusedMLP needs HBM-resident inputs: there is no CPU path")
        y, _, _ = _K.fused_mlp(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, act)     # differentiable registered operator
        return y.to(x.dtype)


class DropoutAddRMSNorm(RMSNorm):
    """flash_attn.ops.rms_norm.DropoutAddRMSNorm as the reference uses it (P:466-467 `partial(DropoutAddRMSNorm, eps=1e-6,
    prenorm=True)`): forward(x, residual=None) -> (rmsnorm(x + residual) * weight, x + residual) with fp32 statistics and the
    weight multiply before the down-cast; p = 0 (no dropout).  prenorm=False returns only the normalised tensor."""

    def __init__(self, hidden_size, prenorm=False, p=0.0, eps=1e-5, residual_in_fp32=False, device=None, dtype=None):
        if p:
            raise InternVideoHipError("DropoutAddRMSNorm (MI355X): dropout is not implemented (InternVideo2 uses p = 0)")
        super().__init__(hidden_size, eps=eps, prenorm=prenorm)
        self.residual_in_fp32 = residual_in_fp32

    def forward(self, x, residual=None):
        if not x.is_cuda:
            raise InternVideoHipError("DropoutAddRMSNorm needs HBM-resident inputs: there is no CPU path")
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if residual is None:                                   # first block: the stream starts as x itself
            res, y, _ = _K.rmsnorm_add(x2.float().contiguous(), None, None, None, 1, self.weight, self.variance_epsilon)
        else:
            xb = (x2 if x2.dtype == BF16 else x2.to(BF16)).contiguous()
            res, y, _ = _K.rmsnorm_add(residual.reshape(-1, shp[-1]).float().contiguous(), xb, None, None, 1, self.weight, self.variance_epsilon)
        y = y.reshape(shp).to(x.dtype)
        if not self.prenorm:
            return y
        res = res.reshape(shp)
        return y, (res if self.residual_in_fp32 else res.to(x.dtype))


class Linear(nn.Linear):
    """nn.Linear (the qkv / proj / decoder-head Linears of the reference's Attention and decoders, P:158-160, 341) on the gfx950 GEMM:
    forward, dgrad and wgrad through the differentiable registered operator `torch.ops.internvideo_hip.linear`.  Parameters, state_dict
    keys and forward signature are nn.Linear's; the result is bf16 (what `model.bfloat16()` produces in the reference's recipe)."""

    def forward(self, x):
        if not x.is_cuda:
            raise InternVideoHipError("Linear needs HBM-resident inputs: there is no CPU path")
        return _K.linear(x, self.weight, self.bias)


class Fp8Linear(nn.Linear):
    """nn.Linear whose forward, dgrad and wgrad GEMMs run on the fp8 (OCP e4m3) MFMA path of gfx950 (per-tensor scaling, fp32
    accumulation) -- the option BASELINE configs[4] names for the InternVideo2-6B encoder (P:758-766).  Parameters, state_dict keys and
    forward signature are nn.Linear's; in/out features must be multiples of 16."""

    def forward(self, x):
        from . import functional as Fn
        if self.in_features % 16 or self.out_features % 16:
            raise InternVideoHipError("Fp8Linear: in / out features must be multiples of 16")
        if not x.is_cuda:
            raise InternVideoHipError("Fp8Linear needs HBM-resident inputs: there is no CPU path")
        return Fn.Fp8LinearFn.apply(x, self.weight, self.bias)
