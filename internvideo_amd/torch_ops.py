"""`torch.ops.internvideo_hip.*`: the hot-path kernels as registered PyTorch operators.

The C ABI (include/internvideo_hip.h) is the product boundary; this module registers its main entry points with the PyTorch dispatcher so
that code written against `torch.ops` (custom-op call sites, `torch.library.opcheck`, fake-tensor tracing / export of a model that uses
the kernels) sees ordinary operators:

    import internvideo_amd.torch_ops                      # registers the library once
    y = torch.ops.internvideo_hip.gemm(a, w, bias, "gelu_erf")
    o, lse = torch.ops.internvideo_hip.flash_attn_fwd(qkv, B, L, H)

Each operator has a CUDA (= HIP on ROCm) implementation that calls the C ABI through internvideo_amd.ops -- there is no CPU
implementation: dispatching one on CPU tensors raises the dispatcher's NotImplementedError -- and a Meta implementation (output shapes
/ dtypes only) for fake-tensor tracing.  Autograd stays where it is (internvideo_amd.functional's Function classes call the same
wrappers); these operators are the forward / backward building blocks, registered without autograd formulas.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32
_LIB = torch.library.Library("internvideo_hip", "DEF")

_LIB.define("gemm(Tensor a, Tensor b, Tensor? bias=None, str act='none', bool a_kc=True, bool b_kc=True, float alpha=1.0, bool out_fp32=False) -> Tensor")
_LIB.define("gemm_dact(Tensor dy, Tensor w, Tensor dact_in, str act='gelu_erf_d') -> Tensor")
_LIB.define("flash_attn_fwd(Tensor qkv, int B, int L, int H, float? scale=None, Tensor? kv_len=None) -> (Tensor, Tensor)")
_LIB.define("flash_attn_bwd(Tensor qkv, Tensor out, Tensor dout, Tensor lse, int B, int L, int H, float? scale=None, Tensor? kv_len=None) -> Tensor")
_LIB.define("rmsnorm_add_fwd(Tensor? res_in, Tensor? branch, Tensor? gamma, Tensor? rowscale, int rows_per_sample, Tensor? w, float eps) -> (Tensor, Tensor, Tensor)")
_LIB.define("layernorm_fwd(Tensor x, Tensor w, Tensor b, float eps) -> (Tensor, Tensor)")
_LIB.define("add_layernorm_fwd(Tensor a, Tensor? r, Tensor w, Tensor b, float eps, bool gelu=False) -> (Tensor, Tensor)")
_LIB.define("cross_entropy_rows(Tensor logits, Tensor labels, int V, int ignore_index=-100) -> Tensor")
_LIB.define("adamw_step_(Tensor(a!) master, Tensor(b!) exp_avg, Tensor(c!) exp_avg_sq, Tensor grad, Tensor(d!)? shadow, float lr, float beta1, "
            "float beta2, float eps, float weight_decay, int step) -> ()")


def _gemm_shape(a, b, a_kc, b_kc):
    M = a.shape[-2] if a_kc else a.shape[-1]
    N = b.shape[-2] if b_kc else b.shape[-1]
    return (*a.shape[:-2], M, N)


def _gemm(a, b, bias=None, act="none", a_kc=True, b_kc=True, alpha=1.0, out_fp32=False):
    return ops.gemm(a, b, a_kc=a_kc, b_kc=b_kc, bias=bias, act=act, alpha=alpha, out_fp32=out_fp32)


def _gemm_meta(a, b, bias=None, act="none", a_kc=True, b_kc=True, alpha=1.0, out_fp32=False):
    return a.new_empty(_gemm_shape(a, b, a_kc, b_kc), dtype=F32 if out_fp32 else BF16)


def _gemm_dact(dy, w, dact_in, act="gelu_erf_d"):
    return ops.gemm(dy, w, a_kc=True, b_kc=False, dact_in=dact_in, act=act)


def _gemm_dact_meta(dy, w, dact_in, act="gelu_erf_d"):
    return dy.new_empty((dy.shape[0], w.shape[1]), dtype=BF16)


def _fa_fwd(qkv, B, L, H, scale=None, kv_len=None):
    return ops.flash_attn_fwd_packed(qkv, B, L, H, scale, kv_len=kv_len)


def _fa_fwd_meta(qkv, B, L, H, scale=None, kv_len=None):
    return qkv.new_empty((B * L, qkv.shape[1] // 3)), qkv.new_empty((B, H, L), dtype=F32)


def _fa_bwd(qkv, out, dout, lse, B, L, H, scale=None, kv_len=None):
    return ops.flash_attn_bwd_packed(qkv, out, dout, lse, B, L, H, scale, kv_len=kv_len)


def _fa_bwd_meta(qkv, out, dout, lse, B, L, H, scale=None, kv_len=None):
    return torch.empty_like(qkv)


def _rms(res_in, branch, gamma, rowscale, rows_per_sample, w, eps):
    res_out, y, rstd = ops.rmsnorm_add_fwd(res_in, branch, gamma, rowscale, rows_per_sample, w, eps)
    src = res_in if res_in is not None else branch
    if y is None:
        y = src.new_empty((0,), dtype=BF16)
    if rstd is None:
        rstd = src.new_empty((0,), dtype=F32)
    return res_out, y, rstd


def _rms_meta(res_in, branch, gamma, rowscale, rows_per_sample, w, eps):
    src = res_in if res_in is not None else branch
    M, D = src.shape
    if w is None:
        return src.new_empty((M, D), dtype=F32), src.new_empty((0,), dtype=BF16), src.new_empty((0,), dtype=F32)
    return src.new_empty((M, D), dtype=F32), src.new_empty((M, D), dtype=BF16), src.new_empty((M,), dtype=F32)


def _ln(x, w, b, eps):
    y, _, stats = ops.layernorm_fwd(x, w, b, eps)
    return y, stats


def _ln_meta(x, w, b, eps):
    return x.new_empty(x.shape, dtype=BF16), x.new_empty((x.shape[0], 2), dtype=F32)


def _aln(a, r, w, b, eps, gelu=False):
    return ops.add_layernorm_fwd(a, r, w, b, eps, gelu=gelu)


def _aln_meta(a, r, w, b, eps, gelu=False):
    return torch.empty_like(a), a.new_empty((a.shape[0], 2), dtype=F32)


def _ce(logits, labels, V, ignore_index=-100):
    return ops.ce_rows(logits, labels, V=V, ignore_index=ignore_index, want_grad=False)[0]


def _ce_meta(logits, labels, V, ignore_index=-100):
    return logits.new_empty((1,), dtype=F32)


def _adamw(master, exp_avg, exp_avg_sq, grad, shadow, lr, beta1, beta2, eps, weight_decay, step):
    ops.adamw_step(master, exp_avg, exp_avg_sq, grad, shadow, lr, beta1, beta2, eps, weight_decay, step)


def _adamw_meta(master, exp_avg, exp_avg_sq, grad, shadow, lr, beta1, beta2, eps, weight_decay, step):
    return None


for _name, _impl, _meta in (("gemm", _gemm, _gemm_meta), ("gemm_dact", _gemm_dact, _gemm_dact_meta), ("flash_attn_fwd", _fa_fwd, _fa_fwd_meta),
                            ("flash_attn_bwd", _fa_bwd, _fa_bwd_meta), ("rmsnorm_add_fwd", _rms, _rms_meta), ("layernorm_fwd", _ln, _ln_meta),
                            ("add_layernorm_fwd", _aln, _aln_meta), ("cross_entropy_rows", _ce, _ce_meta), ("adamw_step_", _adamw, _adamw_meta)):
    _LIB.impl(_name, _impl, "CUDA")
    _LIB.impl(_name, _meta, "Meta")

OPERATORS: Tuple[str, ...] = ("gemm", "gemm_dact", "flash_attn_fwd", "flash_attn_bwd", "rmsnorm_add_fwd", "layernorm_fwd", "add_layernorm_fwd",
                              "cross_entropy_rows", "adamw_step_")
