"""Stage-2 text / fusion tower on the MI355X kernels.

Mirrors InternVideo2/multi_modality/models/backbones/bert/xbert.py for the path stage 2 runs (builder.py:31-45 ->
`BertForMaskedLM`; internvideo2_stage2_visual.py:271-289 `encode_text`; criterions.py:105-182,235-274): the same module tree and
parameter names -- `bert.embeddings.{word,position,token_type}_embeddings`, `bert.encoder.layer.{i}.attention.self.{query,key,value}`,
`.attention.output.{dense,LayerNorm}`, `.crossattention.*` in the layers >= `fusion_layer`, `.intermediate.dense`,
`.output.{dense,LayerNorm}`, `cls.predictions.{bias,transform.dense,transform.LayerNorm,decoder}` -- so a reference checkpoint's text
tower loads key for key, and the same call signature for what stage 2 uses of it (`input_ids` / `encoder_embeds`, `attention_mask`,
`encoder_hidden_states`, `encoder_attention_mask`, `mode` in {"text", "fusion", "multi_modal"}, `labels`, `return_logits`).

All arithmetic is in libinternvideo_hip.so:
  embeddings         ivh_bert_embed_fwd / _bwd     gather of the three fp32 tables + LayerNorm, one pass; backward scatters with fp32 atomics
  q, k, v            one MFMA GEMM on the concatenated weights -> packed [B*L, 3*D] rows, read in place by the attention kernels
  attention          ivh_flash_attn_fwd / _bwd with kv_len (right-padded text; the reference adds -10000 to the padded keys' scores,
                     xbert.py:1118-1120, which removes them from an fp32 softmax exactly) -- self (Lq = Lk = L) and cross (keys / values
                     from the vision tokens, width `encoder_width`)
  dense + LayerNorm  GEMM (+ bias) then ivh_add_layernorm_fwd / _bwd: LayerNorm(dense + residual) on bf16 rows, fp32 statistics
  feed-forward       fc1 GEMM with erf-GELU epilogue (gelu' as by-product), fc2 GEMM whose dgrad epilogue multiplies by gelu'
  MLM / VTM heads    decoder GEMM on the (tied) word-embedding matrix padded to a multiple of 8 columns, ivh_ce_rows for the loss and,
                     in backward, the logits' gradient written over the logits with the upstream gradient as a device scalar
Dropout (config_bert_large.json keeps BERT's 0.1 / 0.1) runs inside the kernels in training mode: hidden dropout in the embedding and
add + LayerNorm kernels, attention-probability dropout in the flash kernels (head dims <= 64), both from a counter-based mask
hash(seed, element index) that the backward kernels regenerate; the seed of every call site comes from `next_dropout_seed()`
(torch.initial_seed() and a host-side call counter: reproducible under torch.manual_seed, no device round trip; the random stream
differs from torch's Philox stream, as any fused dropout's does).
Unsupported (raises): attention masks with holes (text is right-padded), relative position embeddings, head masks, decoder (causal)
mode, past key values.
"""
from __future__ import annotations

import json
import math
from types import SimpleNamespace
from typing import Optional

import torch
from torch import nn

from . import functional as Fn
from . import ops
from .lib import InternVideoHipError

BF16, F32 = torch.bfloat16, torch.float32


class BertConfig:
    """the fields of xbert.py:83-200 that the stage-2 path reads, with the defaults of configs/config_bert_large.json"""

    def __init__(self, vocab_size=30522, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                 hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=512,
                 type_vocab_size=2, initializer_range=0.02, layer_norm_eps=1e-12, pad_token_id=0, position_embedding_type="absolute",
                 fusion_layer=19, encoder_width=1408, **extra):
        self.vocab_size, self.hidden_size, self.num_hidden_layers = vocab_size, hidden_size, num_hidden_layers
        self.num_attention_heads, self.intermediate_size, self.hidden_act = num_attention_heads, intermediate_size, hidden_act
        self.hidden_dropout_prob, self.attention_probs_dropout_prob = hidden_dropout_prob, attention_probs_dropout_prob
        self.max_position_embeddings, self.type_vocab_size = max_position_embeddings, type_vocab_size
        self.initializer_range, self.layer_norm_eps, self.pad_token_id = initializer_range, layer_norm_eps, pad_token_id
        self.position_embedding_type = position_embedding_type
        self.fusion_layer, self.encoder_width = fusion_layer, encoder_width
        for k, v in extra.items():
            setattr(self, k, v)
        if hidden_act != "gelu":
            raise InternVideoHipError(f"hidden_act {hidden_act!r}: the MI355X text tower implements BERT's erf-GELU")
        if position_embedding_type != "absolute":
            raise InternVideoHipError("only absolute position embeddings are implemented (config_bert_large.json)")
        if hidden_size % num_attention_heads:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)" % (hidden_size, num_attention_heads))

    @classmethod
    def from_json_file(cls, path: str) -> "BertConfig":
        with open(path) as f:
            return cls(**json.load(f))


_DROP_CALLS = 0


def next_dropout_seed() -> int:
    """32-bit seed of one dropout call site: a function of torch.initial_seed() and the number of seeds drawn so far in this process"""
    global _DROP_CALLS
    _DROP_CALLS += 1
    return (torch.initial_seed() * 0x9E3779B1 + _DROP_CALLS * 0x85EBCA6B) & 0xFFFFFFFF


_DROP_EPOCH = [None]                                   # device int32 [1] registered with the library, or None


def set_dropout_epoch(t: Optional[torch.Tensor]):
    """register (or, with None, unregister) the device-side dropout epoch: a uint32 / int32 [1] tensor in HBM that the training engine
    advances once per step (IVTrainEngine.dropout_epoch).  While registered, every dropout mask is hash(call-site seed + epoch * K,
    element), so a step captured into a HIP graph draws fresh masks on every replay (include/internvideo_hip.h ivh_set_dropout_epoch)."""
    from .lib import call
    if t is not None and not (t.is_cuda and t.numel() == 1 and t.element_size() == 4):
        raise InternVideoHipError("dropout epoch must be one 32-bit integer in HBM")
    call("ivh_set_dropout_epoch", t.data_ptr() if t is not None else None)
    _DROP_EPOCH[0] = t


# ---- autograd functions over the C ABI ---------------------------------------------------------------------------------------------
class BertEmbedFn(torch.autograd.Function):
    """dropout(LayerNorm((word[ids] + type[0]) + pos[0..L-1])) (xbert.py:298-334) -> bf16 [B*L, D]"""

    @staticmethod
    def forward(ctx, ids, L, word, pos, type_, lnw, lnb, eps, pad_id, drop_p=0.0, seed=0):
        y, stats = ops.bert_embed_fwd(ids, L, Fn.vec(word), Fn.vec(pos), Fn.vec(type_), Fn.vec(lnw), Fn.vec(lnb), eps, drop_p, seed)
        ctx.save_for_backward(ids, stats)
        ctx.p, ctx.L, ctx.pad_id, ctx.drop = (word, pos, type_, lnw, lnb), L, pad_id, (drop_p, seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        ids, stats = ctx.saved_tensors
        word, pos, type_, lnw, lnb = ctx.p
        dword = torch.zeros(word.shape, dtype=F32, device=dy.device)
        dpos = torch.zeros(pos.shape, dtype=F32, device=dy.device)
        dtype_ = torch.zeros(type_.shape, dtype=F32, device=dy.device)
        dw, db = ops.bert_embed_bwd(ids, ctx.L, Fn.vec(word), Fn.vec(pos), Fn.vec(type_), Fn.vec(lnw), stats, dy.contiguous(), ctx.pad_id,
                                    dword, dpos, dtype_, ctx.drop[0], ctx.drop[1])
        # Fn._ret_grad: into the engine's flat buffers when it manages the parameter (then None for autograd), else the plain gradient
        return (None, None, Fn._ret_grad(word, dword), Fn._ret_grad(pos, dpos), Fn._ret_grad(type_, dtype_), Fn._ret_grad(lnw, dw), Fn._ret_grad(lnb, db),
                None, None, None, None)


class AddLayerNormFn(torch.autograd.Function):
    """LayerNorm(dropout(a) + r) on bf16 rows (xbert.py:508-512, 592-596); gelu=True: LayerNorm(gelu(a)) (xbert.py:839-843)"""

    @staticmethod
    def forward(ctx, a, r, w, b, eps, gelu, drop_p=0.0, seed=0):
        a2 = a.reshape(-1, a.shape[-1]).contiguous()
        r2 = r.reshape(-1, r.shape[-1]).contiguous() if r is not None else None
        y, stats = ops.add_layernorm_fwd(a2, r2, Fn.vec(w), Fn.vec(b), eps, gelu=gelu, drop_p=drop_p, seed=seed)
        ctx.save_for_backward(a2, r2, stats)
        ctx.p, ctx.gelu, ctx.shape, ctx.drop = (w, b), gelu, a.shape, (drop_p, seed)
        return y.reshape(a.shape)

    @staticmethod
    def backward(ctx, dy):
        a2, r2, stats = ctx.saved_tensors
        w, b = ctx.p
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        dx, dw, db = ops.add_layernorm_bwd(a2, r2, Fn.vec(w), stats, dy2, gelu=ctx.gelu, drop_p=ctx.drop[0], seed=ctx.drop[1])
        if ctx.drop[0] > 0:
            dx_r, dx_a = dx[0].reshape(ctx.shape), dx[1].reshape(ctx.shape)
        else:
            dx_r = dx_a = dx.reshape(ctx.shape)
        return dx_a, (dx_r if r2 is not None else None), Fn._ret_grad(w, dw), Fn._ret_grad(b, db), None, None, None, None


def _rows8(dy: torch.Tensor, x: torch.Tensor):
    """the transposing weight-gradient GEMM reads its operands rows-contiguous in 8-row groups: pad a ragged row count with zeros"""
    M = dy.shape[0]
    if M % 8 == 0:
        return dy, x
    pad = 8 - M % 8
    return torch.nn.functional.pad(dy, (0, 0, 0, pad)), torch.nn.functional.pad(x, (0, 0, 0, pad))


class CatLinearFn(torch.autograd.Function):
    """[x W_0^T + b_0 | x W_1^T + b_1 | ...] as ONE GEMM on the row-concatenated weights (query / key / value of BertSelfAttention,
    xbert.py:353-359, 400-415): the result is the packed [M, 3*D] layout the attention kernels read in place."""

    @staticmethod
    def forward(ctx, x, *wb):
        ws, bs = wb[0::2], wb[1::2]
        x2 = x.reshape(-1, x.shape[-1])
        if x2.dtype != BF16:
            x2 = x2.to(BF16)
        W = torch.cat([Fn.mat(w) for w in ws], dim=0)
        bias = torch.cat([Fn.vec(b) for b in bs], dim=0)
        y = ops.gemm(x2.contiguous(), W, bias=bias)
        ctx.save_for_backward(x2, W)
        ctx.p, ctx.xshape = (ws, bs), x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, W = ctx.saved_tensors
        ws, bs = ctx.p
        dy2 = dy.contiguous()
        dx = ops.gemm(dy2, W, a_kc=True, b_kc=False).reshape(ctx.xshape) if ctx.needs_input_grad[0] else None
        need_w = [Fn._wants_grad(ctx, 1 + 2 * i, w) for i, w in enumerate(ws)]        # frozen towers (freeze_text): no GEMM, no .grad
        need_b = [Fn._wants_grad(ctx, 2 + 2 * i, b) for i, b in enumerate(bs)]
        dB = ops.colsum_bf16(dy2) if any(need_b) else None
        offs = [0]
        for w in ws:
            offs.append(offs[-1] + w.shape[0])
        deferred = any(need_w) and Fn._defer_to_end(dy2, x2, [(w, offs[i], w.shape[0]) for i, w in enumerate(ws)])    # inside Fn.grouped_weight_grads()
        dW = None
        if any(need_w) and not deferred:
            dyp, xp = _rows8(dy2, x2)
            dW = ops.gemm(dyp, xp, a_kc=False, b_kc=False)
        out = []
        for i, (w, b) in enumerate(zip(ws, bs)):
            out += [Fn._ret_grad(w, dW[offs[i]:offs[i + 1]]) if (dW is not None and need_w[i]) else None,
                    Fn._ret_grad(b, dB[offs[i]:offs[i + 1]]) if need_b[i] else None]
        return (dx, *out)


class SelfAttnFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(hd) + mask) v over packed rows [B*L, 3*D] (xbert.py:417-482); kv_len int32 [B] | None"""

    @staticmethod
    def forward(ctx, qkv, B, L, H, kv_len, drop_p=0.0, seed=0):
        out, lse = ops.flash_attn_fwd_packed(qkv, B, L, H, kv_len=kv_len, drop_p=drop_p, seed=seed)
        ctx.save_for_backward(qkv, out, lse, kv_len)
        ctx.meta = (B, L, H, drop_p, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, kv_len = ctx.saved_tensors
        B, L, H, drop_p, seed = ctx.meta
        return (ops.flash_attn_bwd_packed(qkv, out, dout.contiguous(), lse, B, L, H, kv_len=kv_len, drop_p=drop_p, seed=seed),
                None, None, None, None, None, None)


class CrossAttnFn(torch.autograd.Function):
    """text queries [B, Lq, H, hd] over vision keys / values [B, Lk, H, hd] (xbert.py:404-408: `is_cross_attention`)"""

    @staticmethod
    def forward(ctx, q, k, v, kv_len, drop_p=0.0, seed=0):
        out, lse = ops.flash_attn_fwd(q, k, v, kv_len=kv_len, drop_p=drop_p, seed=seed)
        ctx.save_for_backward(q, k, v, out, lse, kv_len)
        ctx.drop = (drop_p, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse, kv_len = ctx.saved_tensors
        dq, dkv = ops.flash_attn_bwd(q, k, v, out, dout.contiguous(), lse, kv_len=kv_len, drop_p=ctx.drop[0], seed=ctx.drop[1])
        return dq, dkv[0], dkv[1], None, None, None


class LinearCrossEntropyFn(torch.autograd.Function):
    """CrossEntropyLoss(ignore_index)(x W^T + b, labels) for a head whose width V need not be a multiple of 8 (the MLM decoder,
    V = 30522, xbert.py:846-864,1677-1682; the VTM head, V = 2, criterions.py:173-181).  The weight is padded with zero rows to a
    multiple of 8 columns of logits; the loss kernel ignores the padding.  Backward: the logits' gradient (times the upstream scalar,
    read on the device) overwrites the saved logits, then dgrad / wgrad / bias-sum."""

    @staticmethod
    def forward(ctx, x, w, b, labels, ignore_index):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.dtype != BF16:
            x2 = x2.to(BF16)
        x2 = x2.contiguous()
        V = w.shape[0]
        Vp = (V + 7) // 8 * 8
        W = Fn.mat(w)
        bias = Fn.vec(b) if b is not None else None
        if Vp != V:
            W = torch.nn.functional.pad(W, (0, 0, 0, Vp - V))
            bias = torch.nn.functional.pad(bias, (0, Vp - V)) if bias is not None else None
        logits = ops.gemm(x2, W, bias=bias)
        loss, _ = ops.ce_rows(logits, labels, V=V, ignore_index=ignore_index, want_grad=False)
        ctx.save_for_backward(x2, W, logits, labels)
        ctx.p, ctx.V, ctx.ignore, ctx.xshape = (w, b), V, ignore_index, x.shape
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        x2, W, logits, labels = ctx.saved_tensors
        w, b = ctx.p
        V = ctx.V
        if getattr(ctx, "consumed", False):
            raise InternVideoHipError("LinearCrossEntropyFn: the saved logits were overwritten by their gradient in the first backward pass; "
                                      "a second backward through the same graph (retain_graph=True) is not supported")
        ctx.consumed = True
        _, dl = ops.ce_rows(logits, labels, V=V, ignore_index=ctx.ignore, want_grad=True, dscale_dev=g.reshape(1).float().contiguous(),
                            inplace=True)
        dx = ops.gemm(dl, W, a_kc=True, b_kc=False).reshape(ctx.xshape) if ctx.needs_input_grad[0] else None
        dlp, xp = _rows8(dl, x2)
        dW = Fn._ret_grad(w, ops.gemm(dlp, xp, a_kc=False, b_kc=False)[:V])
        db = Fn._ret_grad(b, ops.colsum_bf16(dl)[:V]) if b is not None else None
        return dx, dW, db, None, None


# ---- modules (same tree / names as xbert.py) -----------------------------------------------------------------------------------------
def _drop(module: nn.Module, p: float):
    """(drop probability, seed) of one dropout call site: (0, 0) outside training"""
    if not p or not module.training:
        return 0.0, 0
    if _DROP_EPOCH[0] is None and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        # seeds are scalar launch arguments drawn from a host counter: without the device-side epoch (set_dropout_epoch) a captured
        # step would replay the SAME masks on every step
        raise InternVideoHipError("dropout > 0 while the step is being captured into a HIP graph: every replay would reuse one set of dropout "
                                  "masks.  Register a device-side epoch first (xbert.set_dropout_epoch / IVTrainEngine(dropout_epoch=True)), capture with "
                                  "hidden_dropout_prob = attention_probs_dropout_prob = 0, or run the text tower eagerly")
    return float(p), next_dropout_seed()


class BertEmbeddings(nn.Module):
    """xbert.py:272-334"""

    def __init__(self, config: BertConfig):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=config.pad_token_id)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.register_buffer("position_ids", torch.arange(config.max_position_embeddings).expand((1, -1)))
        self.config = config

    def forward(self, input_ids=None, token_type_ids=None, position_ids=None, inputs_embeds=None, past_key_values_length=0):
        if inputs_embeds is not None or position_ids is not None or past_key_values_length:
            raise InternVideoHipError("BertEmbeddings (MI355X): only the input_ids path with default positions is implemented")
        if token_type_ids is not None and bool((token_type_ids != 0).any()):
            raise InternVideoHipError("BertEmbeddings (MI355X): token_type_ids are all zero on the stage-2 path")
        B, L = input_ids.shape
        y = BertEmbedFn.apply(input_ids, L, self.word_embeddings.weight, self.position_embeddings.weight, self.token_type_embeddings.weight,
                              self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps, self.config.pad_token_id,
                              *_drop(self, self.config.hidden_dropout_prob))
        return y.view(B, L, -1)


class BertSelfAttention(nn.Module):
    """xbert.py:337-498: the three projections (key / value read `encoder_width` features when cross-attending)"""

    def __init__(self, config: BertConfig, is_cross_attention: bool):
        super().__init__()
        self.config = config
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = config.hidden_size // config.num_attention_heads
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        kv_in = config.encoder_width if is_cross_attention else config.hidden_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(kv_in, self.all_head_size)
        self.value = nn.Linear(kv_in, self.all_head_size)
        self.is_cross_attention = is_cross_attention

    def forward(self, hidden_states, kv_len=None, encoder_hidden_states=None, encoder_kv_len=None):
        B, L, D = hidden_states.shape
        H, hd = self.num_attention_heads, self.attention_head_size
        drop = _drop(self, self.config.attention_probs_dropout_prob)
        if drop[0] and hd > 64:
            raise InternVideoHipError(f"attention dropout is built for head dims <= 64 (BERT: 64), got {hd}")
        if encoder_hidden_states is None:
            qkv = CatLinearFn.apply(hidden_states, self.query.weight, self.query.bias, self.key.weight, self.key.bias,
                                    self.value.weight, self.value.bias)
            return SelfAttnFn.apply(qkv, B, L, H, kv_len, *drop).view(B, L, D)
        Lk = encoder_hidden_states.shape[1]
        q = Fn.LinearFn.apply(hidden_states, self.query.weight, self.query.bias).view(B, L, H, hd)
        k = Fn.LinearFn.apply(encoder_hidden_states, self.key.weight, self.key.bias).view(B, Lk, H, hd)
        v = Fn.LinearFn.apply(encoder_hidden_states, self.value.weight, self.value.bias).view(B, Lk, H, hd)
        return CrossAttnFn.apply(q, k, v, encoder_kv_len, *drop).view(B, L, D)


class BertSelfOutput(nn.Module):
    """xbert.py:501-512"""

    def __init__(self, config: BertConfig):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.config = config

    def forward(self, hidden_states, input_tensor):
        h = Fn.LinearFn.apply(hidden_states, self.dense.weight, self.dense.bias)
        return AddLayerNormFn.apply(h, input_tensor, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps, False,
                                    *_drop(self, self.config.hidden_dropout_prob))


class BertAttention(nn.Module):
    """xbert.py:515-567"""

    def __init__(self, config: BertConfig, is_cross_attention: bool = False):
        super().__init__()
        self.self = BertSelfAttention(config, is_cross_attention)
        self.output = BertSelfOutput(config)

    def forward(self, hidden_states, kv_len=None, encoder_hidden_states=None, encoder_kv_len=None):
        ctx = self.self(hidden_states, kv_len, encoder_hidden_states, encoder_kv_len)
        return self.output(ctx, hidden_states)


class BertIntermediate(nn.Module):
    """xbert.py:570-582 (its GELU runs in the epilogue of the GEMM issued by BertLayer)"""

    def __init__(self, config: BertConfig):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)


class BertOutput(nn.Module):
    """xbert.py:585-596"""

    def __init__(self, config: BertConfig):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)


class BertLayer(nn.Module):
    """xbert.py:599-688"""

    def __init__(self, config: BertConfig, layer_num: int):
        super().__init__()
        self.config = config
        self.layer_num = layer_num
        self.attention = BertAttention(config)
        self.has_cross_attention = layer_num >= config.fusion_layer
        if self.has_cross_attention:
            self.crossattention = BertAttention(config, is_cross_attention=True)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)

    def forward(self, hidden_states, kv_len=None, encoder_hidden_states=None, encoder_kv_len=None):
        a = self.attention(hidden_states, kv_len)
        if self.has_cross_attention:
            assert encoder_hidden_states is not None, "encoder_hidden_states must be given for cross-attention layers"
            a = self.crossattention(a, None, encoder_hidden_states, encoder_kv_len)
        f = Fn.MlpFn.apply(a, self.intermediate.dense.weight, self.intermediate.dense.bias, self.output.dense.weight, self.output.dense.bias,
                           "gelu")
        return AddLayerNormFn.apply(f, a, self.output.LayerNorm.weight, self.output.LayerNorm.bias, self.output.LayerNorm.eps, False,
                                    *_drop(self, self.config.hidden_dropout_prob))


class BertEncoder(nn.Module):
    """xbert.py:690-812: `mode` selects the layer range -- "text": [0, fusion_layer), "fusion": [fusion_layer, N), "multi_modal": all"""

    def __init__(self, config: BertConfig):
        super().__init__()
        self.config = config
        self.layer = nn.ModuleList([BertLayer(config, i) for i in range(config.num_hidden_layers)])

    def forward(self, hidden_states, kv_len=None, encoder_hidden_states=None, encoder_kv_len=None, mode="multi_modal"):
        if mode in ("text", "temporal"):
            lo, hi = 0, self.config.fusion_layer
        elif mode == "fusion":
            lo, hi = self.config.fusion_layer, self.config.num_hidden_layers
        elif mode == "multi_modal":
            lo, hi = 0, self.config.num_hidden_layers
        else:
            raise ValueError(f"unknown mode {mode!r}")
        recompute = bool(getattr(self.config, "gradient_checkpointing", False)) and self.training and torch.is_grad_enabled()
        for i in range(lo, hi):
            if recompute:                              # xbert.py:743-765: torch.utils.checkpoint around every layer while training
                hidden_states = _checkpointed_layer(self.layer[i], hidden_states, kv_len, encoder_hidden_states, encoder_kv_len)
            else:
                hidden_states = self.layer[i](hidden_states, kv_len, encoder_hidden_states, encoder_kv_len)
        return hidden_states


def _checkpointed_layer(layer, hidden_states, kv_len, encoder_hidden_states, encoder_kv_len):
    """`config.gradient_checkpointing` (builder.py:23 <- the stage-2 configs' `gradient_checkpointing = True # for text encoder`): keep only
    the layer's input, run the layer again inside backward.  The dropout masks of the kernels are functions of (seed, element index) with
    seeds drawn from a host counter, so the second run rewinds the counter to where the first one started: same masks, and the
    recomputed layer is bit-identical to the stored one (tests/test_bert_gpu.py)."""
    from torch.utils.checkpoint import checkpoint
    state = {"start": None}

    def run(h, enc):
        global _DROP_CALLS
        if state["start"] is None:
            state["start"] = _DROP_CALLS
            return layer(h, kv_len, enc, encoder_kv_len)
        keep, _DROP_CALLS = _DROP_CALLS, state["start"]
        try:
            return layer(h, kv_len, enc, encoder_kv_len)
        finally:
            _DROP_CALLS = keep

    return checkpoint(run, hidden_states, encoder_hidden_states, use_reentrant=False, preserve_rng_state=False)


def right_padded_lengths(mask: Optional[torch.Tensor], what: str) -> Optional[torch.Tensor]:
    """attention mask (B, L), 1 = attend -> int32 lengths [B], or None when nothing is masked.  The kernels exclude a suffix of the
    keys; masks with holes raise (one host read per call, on the (B, L) mask)."""
    if mask is None:
        return None
    cached = getattr(mask, "_ivh_kv_len", False)
    if cached is not False:                            # set below / by callers that derive a mask from checked ones (stage2.vtm_loss).  The cache
        return cached                                  # lives on the tensor object: a mask edited in place needs `del mask._ivh_kv_len`
    keep = mask.to(torch.bool)
    n = keep.sum(1, dtype=torch.int32)
    prefix = torch.arange(keep.shape[1], device=keep.device).unsqueeze(0) < n.unsqueeze(1)
    ok = torch.stack([(keep == prefix).all(), (n > 0).all(), (n == keep.shape[1]).all()]).tolist()
    if not ok[0]:
        raise InternVideoHipError(f"{what} must be right-padded (a prefix of ones per row)")
    if not ok[1]:
        raise InternVideoHipError(f"{what}: every sequence needs at least one valid token")
    mask._ivh_kv_len = None if ok[2] else n.contiguous()
    return mask._ivh_kv_len


class BertModel(nn.Module):
    """xbert.py:1013-1296 without the pooler (builder.py:47-54 / BertForMaskedLM build it with add_pooling_layer=False)"""

    def __init__(self, config: BertConfig, add_pooling_layer: bool = False):
        super().__init__()
        if add_pooling_layer:
            raise InternVideoHipError("BertModel (MI355X): stage 2 builds the tower without the pooler")
        self.config = config
        self.embeddings = BertEmbeddings(config)
        self.encoder = BertEncoder(config)
        self.pooler = None
        self.apply(_init_weights(config))

    def get_input_embeddings(self):
        return self.embeddings.word_embeddings

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None, inputs_embeds=None,
                encoder_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None, past_key_values=None, use_cache=None,
                output_attentions=None, output_hidden_states=None, return_dict=None, is_decoder=False, mode="multi_modal",
                normalize_attention=True):
        if head_mask is not None or past_key_values is not None or use_cache or output_attentions or output_hidden_states or is_decoder \
                or inputs_embeds is not None or isinstance(encoder_hidden_states, (list, tuple)):
            raise InternVideoHipError("BertModel (MI355X): head masks, caches, attention / hidden-state outputs, decoder mode, inputs_embeds "
                                      "and lists of encoder states are outside the stage-2 training path")
        if encoder_embeds is None:
            if input_ids is None:
                raise ValueError("You have to specify either input_ids or inputs_embeds or encoder_embeds")
            if not input_ids.is_cuda:
                raise InternVideoHipError("input_ids must live in HBM; there is no CPU path")
            h = self.embeddings(input_ids=input_ids, token_type_ids=token_type_ids, position_ids=position_ids)
        else:
            h = encoder_embeds if encoder_embeds.dtype == BF16 else encoder_embeds.to(BF16)
        kv_len = right_padded_lengths(attention_mask, "attention_mask")
        enc, enc_len = None, None
        if encoder_hidden_states is not None:
            enc = encoder_hidden_states if encoder_hidden_states.dtype == BF16 else encoder_hidden_states.to(BF16)
            enc_len = right_padded_lengths(encoder_attention_mask, "encoder_attention_mask")
        out = self.encoder(h, kv_len, enc, enc_len, mode=mode)
        if return_dict is False:
            return (out, None)
        return SimpleNamespace(last_hidden_state=out, pooler_output=None, past_key_values=None, hidden_states=None, attentions=None,
                               cross_attentions=None)


class BertPredictionHeadTransform(nn.Module):
    """xbert.py:829-843"""

    def __init__(self, config: BertConfig):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = nn.LayerNorm(config.hidden_size, eps=config.layer_norm_eps)

    def forward(self, hidden_states):
        h = Fn.LinearFn.apply(hidden_states, self.dense.weight, self.dense.bias)
        return AddLayerNormFn.apply(h, None, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps, True)


class BertLMPredictionHead(nn.Module):
    """xbert.py:846-864: decoder weight tied to the word embeddings (done by BertForMaskedLM), output-only bias shared with `decoder.bias`"""

    def __init__(self, config: BertConfig):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.vocab_size))
        self.decoder.bias = self.bias

    def forward(self, hidden_states):
        t = self.transform(hidden_states)
        V = self.decoder.weight.shape[0]
        if V % 8:
            raise InternVideoHipError("materialised MLM logits need a vocabulary that is a multiple of 8; use `labels=` (fused loss)")
        return Fn.LinearFn.apply(t, self.decoder.weight, self.bias)


class BertOnlyMLMHead(nn.Module):
    """xbert.py:866-873"""

    def __init__(self, config: BertConfig):
        super().__init__()
        self.predictions = BertLMPredictionHead(config)

    def forward(self, sequence_output):
        return self.predictions(sequence_output)


class BertForMaskedLM(nn.Module):
    """xbert.py:1592-1698: `bert` (no pooler) + `cls`; forward(..., labels) -> .loss = CrossEntropyLoss over the labelled tokens."""

    def __init__(self, config: BertConfig):
        super().__init__()
        self.config = config
        self.bert = BertModel(config, add_pooling_layer=False)
        self.cls = BertOnlyMLMHead(config)
        self.cls.apply(_init_weights(config))
        self.cls.predictions.decoder.weight = self.bert.embeddings.word_embeddings.weight          # tie_word_embeddings (xbert.py:1606-1614)

    def get_output_embeddings(self):
        return self.cls.predictions.decoder

    def forward(self, input_ids=None, attention_mask=None, token_type_ids=None, position_ids=None, head_mask=None, inputs_embeds=None,
                encoder_embeds=None, encoder_hidden_states=None, encoder_attention_mask=None, labels=None, output_attentions=None,
                output_hidden_states=None, return_dict=None, is_decoder=False, mode="multi_modal", normalize_attention=True,
                soft_labels=None, alpha=0, return_logits=False):
        if soft_labels is not None:
            raise InternVideoHipError("BertForMaskedLM (MI355X): soft-label distillation is not on the stage-2 path (criterions.py:262)")
        out = self.bert(input_ids, attention_mask=attention_mask, token_type_ids=token_type_ids, position_ids=position_ids,
                        head_mask=head_mask, inputs_embeds=inputs_embeds, encoder_embeds=encoder_embeds,
                        encoder_hidden_states=encoder_hidden_states, encoder_attention_mask=encoder_attention_mask,
                        output_attentions=output_attentions, output_hidden_states=output_hidden_states, return_dict=True,
                        is_decoder=is_decoder, mode=mode)
        seq = out.last_hidden_state
        if return_logits or labels is None:
            logits = self.cls(seq)
            if return_logits:
                return logits
            return SimpleNamespace(loss=None, loss_aux=0.0, logits=logits, hidden_states=None, attentions=None)
        pred = self.cls.predictions
        t = pred.transform(seq)
        loss = LinearCrossEntropyFn.apply(t, pred.decoder.weight, pred.bias, labels.reshape(-1), -100)
        return SimpleNamespace(loss=loss, loss_aux=0.0, logits=None, hidden_states=None, attentions=None)


def _init_weights(config: BertConfig):
    """xbert.py:905-917"""
    def init(module):
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=config.initializer_range)
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()
    return init


def build_bert(model_config, pretrain: bool, checkpoint: bool = False, encoder_width: Optional[int] = None):
    """builder.py:9-68 `build_bert`: text-encoder config from the JSON file (or a dict / BertConfig under `text_encoder.config`),
    `encoder_width` = the vision tower's d_model, `fusion_layer` from the model config (all layers text-only when the multimodal part is
    disabled); `BertForMaskedLM` for pre-training, `BertModel` otherwise.  Random init: there is no network for `from_pretrained`
    (load the reference checkpoint's `text_encoder.*` keys with load_state_dict)."""
    te = model_config["text_encoder"] if isinstance(model_config, dict) else model_config.text_encoder
    get = (lambda o, k, d=None: o.get(k, d)) if isinstance(te, dict) else (lambda o, k, d=None: getattr(o, k, d))
    src = get(te, "config")
    if isinstance(src, BertConfig):
        cfg = src
    elif isinstance(src, dict):
        cfg = BertConfig(**src)
    else:
        cfg = BertConfig.from_json_file(src)
    ve = model_config["vision_encoder"] if isinstance(model_config, dict) else model_config.vision_encoder
    cfg.encoder_width = encoder_width if encoder_width is not None else get(ve, "d_model")
    cfg.gradient_checkpointing = bool(checkpoint)          # builder.py:23; honoured by BertEncoder.forward while training
    cfg.fusion_layer = get(te, "fusion_layer", cfg.fusion_layer)
    mm = model_config["multimodal"] if isinstance(model_config, dict) else getattr(model_config, "multimodal", None)
    if mm is not None and not get(mm, "enable", True):
        cfg.fusion_layer = cfg.num_hidden_layers
    return BertForMaskedLM(cfg) if pretrain else BertModel(cfg, add_pooling_layer=False)
