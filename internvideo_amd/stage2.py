"""Stage-2 video<->text training seam (VTM / MLM over the BERT fusion tower).

Mirrors InternVideo2/multi_modality/models/criterions.py:15-103,200-216 (`get_sim`, `VTC_VTM_Loss.vtc_loss`, `get_mask`) and
multi_modality/models/utils.py:193-212 (`AllGather` / `allgather_wgrad`): same names, argument meaning and return values.
The arithmetic (F.normalize, v t^T / temp, soft-target symmetric cross entropy, and its whole backward) is ONE C-ABI call
(ivh_vtc_loss_fwd_bwd); the feature exchange is ONE RCCL all-gather of a packed [vision | text | idx] row block instead of the
reference's three list-all-gathers + cat (collective C2 of SURVEY.md 2.4), whose backward is the local row slice, as in the reference.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch
import torch.distributed as dist
from torch import nn

from . import ops


class AllGather(torch.autograd.Function):
    """utils.py:193-212.  forward: rows of every rank concatenated in rank order; backward: this rank's slice of the gradient
    (no cross-rank reduction, exactly like the reference)."""

    @staticmethod
    def forward(ctx, tensor, args):
        ctx.rank, ctx.batch_size = int(args.rank), tensor.shape[0]
        t = tensor.contiguous()
        out = torch.empty((int(args.world_size) * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=getattr(args, "group", None))
        return out

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output[ctx.batch_size * ctx.rank: ctx.batch_size * (ctx.rank + 1)], None


allgather_wgrad = AllGather.apply


def _gather_args(group=None):
    if dist.is_available() and dist.is_initialized():
        return SimpleNamespace(world_size=dist.get_world_size(group), rank=dist.get_rank(group), group=group)
    return SimpleNamespace(world_size=1, rank=0, group=group)


class _VTCFn(torch.autograd.Function):
    """loss = vtc(v_all, t_all, idx_all, temp) with gradients for v_all, t_all and temp from the same kernel pass."""

    @staticmethod
    def forward(ctx, v_all, t_all, idx_all, temp):
        # criterions.py passes the clamped nn.Parameter: a device-resident temperature is read by the kernels (no host synchronisation)
        tval = temp if (isinstance(temp, torch.Tensor) and temp.is_cuda) else float(temp)
        loss, sim, dv, dt, dtemp = ops.vtc_loss_fwd_bwd(v_all.float(), t_all.float(), idx_all, tval, want_grad=True)
        ctx.save_for_backward(dv, dt, dtemp)
        ctx.temp_is_tensor = isinstance(temp, torch.Tensor)
        ctx.dtypes = (v_all.dtype, t_all.dtype)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        dv, dt, dtemp = ctx.saved_tensors
        gt = (dtemp.reshape(()) * g) if ctx.temp_is_tensor else None
        return (dv * g).to(ctx.dtypes[0]), (dt * g).to(ctx.dtypes[1]), None, gt


class _AbtFn(torch.autograd.Function):
    """a [ni, K] x b [nj, K] -> a b^T, fp32 (ivh_vtc_abt), with its backward on the same kernel"""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.float().contiguous(), b.float().contiguous()
        ctx.save_for_backward(a, b)
        return ops.abt_f32(a, b)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.float().contiguous()
        da = ops.abt_f32(g, b.t().contiguous()) if ctx.needs_input_grad[0] else None          # da[i, k] = sum_j g[i, j] b[j, k]
        db = ops.abt_f32(g.t().contiguous(), a.t().contiguous()) if ctx.needs_input_grad[1] else None
        return da, db


def _agg(sim, agg_method):
    """criterions.py:34-39 / 43-48: the frame axis (dim 1) is averaged or maximised; any other value leaves the logits 3-D, as the reference does"""
    if agg_method == "mean":
        return sim.mean(1)
    if agg_method == "max":
        return sim.max(1)[0]
    return sim


def get_sim(vision_proj: torch.Tensor, text_proj: torch.Tensor, temp=1.0, agg_method="mean"):
    """criterions.py:15-55 -> (sim_v2t, sim_t2v).  Pooled 2-D features (what InternVideo2 stage 2 trains on) are one fused kernel call; FRAME-LEVEL
    features -- vision_proj [B, L, C] against text_proj [B, C] (criterions.py:31-39), or text_proj [B, L, C] against vision_proj [B, C]
    (:40-48) -- are normalised, multiplied on the fp32 logits kernel and aggregated over the frame axis by `agg_method` ("mean" | "max")."""
    if vision_proj.ndim == 2 and text_proj.ndim == 2:
        tval = temp if (isinstance(temp, torch.Tensor) and temp.is_cuda) else float(temp)
        _, sim, _, _, _ = ops.vtc_loss_fwd_bwd(vision_proj.float(), text_proj.float(), None, tval, want_grad=False)
        return sim, sim.T
    vn = torch.nn.functional.normalize(vision_proj.float(), dim=-1)
    tn = torch.nn.functional.normalize(text_proj.float(), dim=-1)
    if vision_proj.ndim == 3 and text_proj.ndim == 2:
        Bv, Lf, C = vn.shape
        s = _AbtFn.apply(vn.reshape(Bv * Lf, C), tn).reshape(Bv, Lf, tn.shape[0]) / temp        # "mld,nd->mln"
        return _agg(s, agg_method), _agg(s.permute(2, 1, 0), agg_method)                        # "nd,mld->nlm" is its (n, l, m) view
    if text_proj.ndim == 3 and vision_proj.ndim == 2:
        Bt, Lf, C = tn.shape
        s = _AbtFn.apply(tn.reshape(Bt * Lf, C), vn).reshape(Bt, Lf, vn.shape[0]) / temp        # "nld,md->nlm": [text n, l, vision m]
        return _agg(s.permute(2, 1, 0), agg_method), _agg(s, agg_method)                        # "nd,mld->nlm": [vision n, l, text m]
    raise ValueError(f"get_sim: vision_proj {tuple(vision_proj.shape)} / text_proj {tuple(text_proj.shape)}: one of them may carry a frame axis")


class VTC_VTM_Loss(nn.Module):
    """criterions.py:58-216: video-text contrastive (VTC) and video-text matching (VTM) losses."""

    def __init__(self, vtm_hard_neg: bool = True, process_group=None):
        super().__init__()
        self.vtm_hard_neg = vtm_hard_neg
        self.process_group = process_group

    def get_gather_args(self):
        return _gather_args(self.process_group)

    @torch.no_grad()
    def get_mask(self, sim, idx=None, normalize=False):
        """criterions.py:200-216"""
        if idx is not None:
            idx = idx.view(-1, 1)
            mask = torch.eq(idx, idx.T).to(sim.dtype)
            if normalize:
                mask = mask / mask.sum(1, keepdim=True)
        else:
            mask = torch.zeros_like(sim)
            mask.fill_diagonal_(1)
        return mask

    def vtc_loss(self, vision_proj: torch.Tensor, text_proj: torch.Tensor, idx: Optional[torch.Tensor], temp=1.0,
                 all_gather: bool = True, agg_method: str = "mean") -> torch.Tensor:
        args = self.get_gather_args()
        if vision_proj.ndim != 2 or text_proj.ndim != 2:     # frame-level features (criterions.py:31-50): gathered tensor by tensor as :84-89
            if all_gather and args.world_size > 1:
                vision_proj = allgather_wgrad(vision_proj, args)
                text_proj = allgather_wgrad(text_proj, args)
                if idx is not None:
                    idx = allgather_wgrad(idx, args)
            sim_v2t, sim_t2v = get_sim(vision_proj, text_proj, temp, agg_method=agg_method)
            with torch.no_grad():
                targets = self.get_mask(sim_v2t, idx=idx, normalize=True)
            loss_i2t = -torch.sum(torch.nn.functional.log_softmax(sim_v2t, dim=1) * targets, dim=1).mean()
            loss_t2i = -torch.sum(torch.nn.functional.log_softmax(sim_t2v, dim=1) * targets, dim=1).mean()
            return (loss_i2t + loss_t2i) / 2
        if all_gather and args.world_size > 1:
            C = vision_proj.shape[1]
            cols = [vision_proj.float(), text_proj.float()]
            if idx is not None:                              # idx rides in the same packet (exact for |idx| < 2^24; larger ids use 2 floats)
                i64 = idx.to(torch.int64)
                cols += [(i64 >> 24).float().unsqueeze(1), (i64 & 0xFFFFFF).float().unsqueeze(1)]
            packed = allgather_wgrad(torch.cat(cols, dim=1), args)
            v_all, t_all = packed[:, :C], packed[:, C:2 * C]
            idx_all = None
            if idx is not None:
                idx_all = (packed[:, 2 * C].detach().to(torch.int64) << 24) | packed[:, 2 * C + 1].detach().to(torch.int64)
        else:
            v_all, t_all, idx_all = vision_proj, text_proj, idx
        return _VTCFn.apply(v_all.contiguous(), t_all.contiguous(), idx_all, temp)


    @torch.no_grad()
    def vtm_negative_weights(self, vision_proj, text_proj, temp, idx=None):
        """criterions.py:133-146 -> (weights_v2t, weights_t2v): softmax(sim + 1e-4) with the same-example pairs zeroed"""
        sim_v2t, sim_t2v = get_sim(vision_proj, text_proj, temp)
        weights_v2t = torch.softmax(sim_v2t + 1e-4, dim=1)
        weights_t2v = torch.softmax(sim_t2v + 1e-4, dim=1)
        mask = self.get_mask(sim_v2t, idx=idx).bool()
        weights_v2t.masked_fill_(mask, 0)
        weights_t2v.masked_fill_(mask, 0)
        weights_v2t = torch.nan_to_num_(weights_v2t, nan=1e-2, posinf=1e-2, neginf=1e-2)
        weights_t2v = torch.nan_to_num_(weights_t2v, nan=1e-2, posinf=1e-2, neginf=1e-2)
        return weights_v2t, weights_t2v, mask

    def get_rand_indices(self, mask, k):
        """criterions.py:184-198"""
        mask = mask.float()
        mask = mask - 10000 * mask
        mask += torch.randn_like(mask)
        _, indices = torch.sort(mask, dim=1, descending=True)
        return indices[:, :k].contiguous()

    def vtm_loss(self, multimodal_encoder, vtm_head: nn.Module, temp, vision_embeds: torch.Tensor, text_embeds: torch.Tensor,
                 vision_proj: torch.Tensor, text_proj: torch.Tensor, text_atts: torch.Tensor, idx: Optional[torch.Tensor],
                 neg_indices=None) -> torch.Tensor:
        """criterions.py:105-182.  One hard negative video per text and one hard negative text per video (multinomial over the
        contrastive weights, or uniform over the other examples without `vtm_hard_neg`), a fusion pass over the 3B pairs
        [positives | (negative video, text) | (video, negative text)], the 2-way head on the [CLS] state and its cross entropy.
        neg_indices = (vision_neg, text_neg) overrides the draw (parity tests); everything else follows the reference call for call."""
        weights_v2t, weights_t2v, mask = self.vtm_negative_weights(vision_proj.detach(), text_proj.detach(), temp, idx)
        if neg_indices is not None:
            vision_neg_indices, txt_neg_indices = neg_indices
        elif self.vtm_hard_neg:
            vision_neg_indices = torch.multinomial(weights_t2v, 1).squeeze(1)
            txt_neg_indices = torch.multinomial(weights_v2t, 1).squeeze(1)
        else:
            vision_neg_indices = self.get_rand_indices(mask, 1).squeeze(1)
            txt_neg_indices = self.get_rand_indices(mask, 1).squeeze(1)
        vision_embeds_neg = vision_embeds[vision_neg_indices]
        text_embeds_neg = text_embeds[txt_neg_indices]
        text_atts_neg = text_atts[txt_neg_indices]
        vision_embeds_all = torch.cat([vision_embeds, vision_embeds_neg, vision_embeds], dim=0)
        text_embeds_all = torch.cat([text_embeds, text_embeds, text_embeds_neg], dim=0)
        text_atts_all = torch.cat([text_atts, text_atts, text_atts_neg], dim=0)
        from .xbert import right_padded_lengths
        n = right_padded_lengths(text_atts, "attention_mask")          # checked once per batch; the 3B mask is a gather of checked rows
        if n is None:
            text_atts_all._ivh_kv_len = None
        else:
            text_atts_all._ivh_kv_len = torch.cat([n, n, n[txt_neg_indices]]).contiguous()
        output = multimodal_encoder(encoder_embeds=text_embeds_all, attention_mask=text_atts_all, encoder_hidden_states=vision_embeds_all,
                                    encoder_attention_mask=None, return_dict=True, mode="fusion")     # vision_atts are all ones (:134-136)
        vtm_embeds = output.last_hidden_state[:, 0]
        bs = vtm_embeds.shape[0] // 3
        vtm_labels = torch.ones(3 * bs, dtype=torch.int32, device=vtm_embeds.device)
        vtm_labels[bs:] = 0
        from .xbert import LinearCrossEntropyFn
        return LinearCrossEntropyFn.apply(vtm_embeds, vtm_head.weight, vtm_head.bias, vtm_labels, -100)          # :173-181


class MLMLoss(nn.Module):
    """criterions.py:227-342: masked language modelling over the fusion tower."""

    def __init__(self, masking_prob, tokenizer):
        super().__init__()
        self.tokenizer = tokenizer
        self.masking_prob = masking_prob

    def mlm_loss(self, text_encoder, text, vision_embeds, vision_atts, draws=None):
        """:235-274.  draws = (masked_indices, indices_replaced, indices_random, random_words) overrides the random draws (parity tests)."""
        input_ids = text.input_ids.clone()
        labels = input_ids.clone()
        input_ids, labels = self.mask(input_ids, text_encoder.config.vocab_size, input_ids.device, targets=labels,
                                      probability_matrix=torch.full(labels.shape, self.masking_prob, device=input_ids.device), draws=draws)
        intermediate_mlm_output = text_encoder.bert(input_ids, attention_mask=text.attention_mask, encoder_hidden_states=vision_embeds,
                                                    encoder_attention_mask=vision_atts, return_dict=True, mode="text")
        text_embeds = intermediate_mlm_output.last_hidden_state
        return self.simple_mlm_loss(text_encoder, text, text_embeds, vision_embeds, vision_atts, labels)

    def simple_mlm_loss(self, text_encoder, text, text_embeds, vision_embeds, vision_atts, labels):
        """:276-295"""
        mlm_output = text_encoder(encoder_embeds=text_embeds, attention_mask=text.attention_mask, encoder_hidden_states=vision_embeds,
                                  encoder_attention_mask=vision_atts, return_dict=True, labels=labels, soft_labels=None, mode="fusion")
        return mlm_output.loss

    def mask(self, input_ids, vocab_size, device, targets=None, masked_indices=None, probability_matrix=None, draws=None):
        """:297-342 on the device of the ids (the reference draws on the CPU): never [PAD] / [CLS]; 80 % [MASK], 10 % random word, 10 % kept"""
        if draws is not None:
            masked_indices, d_replace, d_random, random_words = [t.to(device) for t in draws]
            masked_indices = masked_indices.bool().clone()
        else:
            if masked_indices is None:
                masked_indices = torch.bernoulli(probability_matrix.to(device)).bool()
            d_replace = torch.bernoulli(torch.full(input_ids.shape, 0.8, device=device))
            d_random = torch.bernoulli(torch.full(input_ids.shape, 0.5, device=device))
            random_words = torch.randint(vocab_size, input_ids.shape, dtype=torch.long, device=device)
        # boolean-mask assignments of the reference written as masked_fill / where: the same result element for element, without the
        # nonzero() (a host synchronisation) behind `x[mask] = y[mask]` -- the whole loss can be captured into a HIP graph
        masked_indices = masked_indices & (input_ids != self.tokenizer.pad_token_id) & (input_ids != self.tokenizer.cls_token_id)
        if targets is not None:
            targets.masked_fill_(~masked_indices, -100)
        indices_replaced = d_replace.bool() & masked_indices
        input_ids.masked_fill_(indices_replaced, self.tokenizer.mask_token_id)
        indices_random = d_random.bool() & masked_indices & ~indices_replaced
        input_ids.copy_(torch.where(indices_random, random_words.to(input_ids.dtype), input_ids))
        if targets is not None:
            return input_ids, targets
        return input_ids


class new_UTA_Loss(nn.Module):
    """criterions.py:458-486: the unmasked-teacher alignment loss of stage 2 on materialised student / teacher features."""

    def __init__(self, distill_final_features=True, clip_loss_ratio=(1., 1.)):
        super().__init__()
        self.distill_final_features = distill_final_features
        self.clip_loss_ratio = clip_loss_ratio

    def uta_loss(self, student_output, student_output_final, targets_clip_middle_vis, targets_clip_final_vis):
        from . import functional as Fn
        loss_clip_middle = Fn.CosineAlignLossFn.apply(student_output, targets_clip_middle_vis)
        if self.distill_final_features and self.clip_loss_ratio[1] > 0:
            loss_clip_final = Fn.CosineAlignLossFn.apply(student_output_final, targets_clip_final_vis)
        else:
            loss_clip_final = torch.zeros(1, dtype=loss_clip_middle.dtype, device=loss_clip_middle.device)
        return loss_clip_middle * self.clip_loss_ratio[0] + loss_clip_final * self.clip_loss_ratio[1]


class Stage2VisionTextHeads(nn.Module):
    """The vision<->text contrastive head of `InternVideo2_Stage2_visual` (multi_modality/models/internvideo2_stage2_visual.py:40-44,
    103-104,117-120,291-294) without the towers: `vision_proj` Linear(clip_embed_dim -> embed_dim), `text_proj`
    Linear(text_width -> embed_dim), the learnable temperature clamped to [0.001, 0.5] at the start of every forward, the UTA
    alignment loss and the VTC loss (all-gathered over the data-parallel group).  Same parameter names as the reference model
    (`vision_proj.*`, `text_proj.*`, `temp`), so its checkpoints' head weights load.  The BERT text / fusion tower (VTM, MLM) is
    SURVEY.md 8(f) row 2: the caller supplies `pooled_text_embeds`."""

    def __init__(self, vision_width: int = 768, text_width: int = 1024, embed_dim: int = 512, temp: float = 0.07,
                 distill_final_features: bool = True, clip_loss_ratio=(1., 1.), loss_weight=None, process_group=None):
        super().__init__()
        self.vision_proj = nn.Linear(vision_width, embed_dim)
        self.text_proj = nn.Linear(text_width, embed_dim)
        self.temp = nn.parameter.Parameter(torch.ones([]) * temp)
        self.criterion_uta = new_UTA_Loss(distill_final_features, clip_loss_ratio)
        self.criterion_vtc_vtm = VTC_VTM_Loss(False, process_group=process_group)
        self.loss_weight = dict(uta=1.0, vtc=1.0) if loss_weight is None else dict(loss_weight)

    @torch.no_grad()
    def clip_contrastive_temperature(self, min_val=0.001, max_val=0.5):
        """internvideo2_stage2_visual.py:291-294"""
        self.temp.clamp_(min_val, max_val)

    def forward(self, pooled_vision_embeds, pooled_text_embeds, idx, student_output=None, student_output_final=None,
                targets_clip_middle_vis=None, targets_clip_final_vis=None):
        """-> dict(loss_uta=..., loss_vtc=...) weighted like internvideo2_stage2_visual.py:162-170"""
        from . import functional as Fn
        self.clip_contrastive_temperature()
        vision_proj = Fn.LinearFn.apply(pooled_vision_embeds, self.vision_proj.weight, self.vision_proj.bias)     # :103
        text_proj = Fn.LinearFn.apply(pooled_text_embeds, self.text_proj.weight, self.text_proj.bias)            # :104
        out = {}
        if self.loss_weight.get("uta", 0) != 0 and student_output is not None:
            out["loss_uta"] = self.criterion_uta.uta_loss(student_output, student_output_final, targets_clip_middle_vis,
                                                          targets_clip_final_vis) * self.loss_weight["uta"]
        if self.loss_weight.get("vtc", 0) != 0:
            out["loss_vtc"] = self.criterion_vtc_vtm.vtc_loss(vision_proj, text_proj, idx, self.temp, all_gather=True) * self.loss_weight["vtc"]
        return out


def _cfg(obj, path: str, default=None):
    """nested lookup `a.b.c` through dicts / attribute containers (the reference's EasyDict configs)"""
    for key in path.split("."):
        if obj is None:
            return default
        if isinstance(obj, dict):
            obj = obj.get(key, None)
        else:
            obj = getattr(obj, key, None)
    return default if obj is None else obj


class InternVideo2_Stage2_visual(nn.Module):
    """multi_modality/models/internvideo2_stage2_visual.py:17-360: the stage-2 model -- masked video encoder (+ frozen CLIP teacher for
    the unmasked-teacher alignment), BERT text / fusion tower, the two projection heads, the learnable temperature and the 2-way
    matching head; `forward(image, text, idx)` -> dict(loss_uta, loss_vtc, loss_vtm, loss_mlm), each already multiplied by its weight.
    Same attribute / parameter names as the reference (`vision_encoder.*`, `text_encoder.*`, `vision_proj.*`, `text_proj.*`, `temp`,
    `itm_head.*`), so a stage-2 checkpoint maps key for key.  `config` is the reference's nested config (dict or attribute container:
    `model.vision_encoder.*`, `model.text_encoder.*`, `model.embed_dim`, `model.temp`, `criterion.*`, `gradient_checkpointing`);
    the three towers may also be passed in prebuilt (tests, or towers restored elsewhere)."""

    def __init__(self, config, tokenizer, is_pretrain: bool = True, vision_encoder: Optional[nn.Module] = None,
                 text_encoder: Optional[nn.Module] = None, clip_teacher: Optional[nn.Module] = None):
        super().__init__()
        self.config, self.tokenizer, self.is_pretrain = config, tokenizer, is_pretrain
        self.vision_width = _cfg(config, "model.vision_encoder.clip_embed_dim")
        self.text_width = _cfg(config, "model.text_encoder.d_model")
        self.embed_dim = _cfg(config, "model.embed_dim")
        self.clip_teacher = clip_teacher
        self.vision_encoder = vision_encoder if vision_encoder is not None else self.build_vision_encoder()
        self._mask_parameters()
        if _cfg(config, "model.freeze_vision", False):
            self.freeze_vision()
        self.text_encoder = text_encoder if text_encoder is not None else self.build_text_encoder()
        if _cfg(config, "model.freeze_text", False):
            self.freeze_text()
        self.vision_proj = nn.Linear(self.vision_width, self.embed_dim)
        self.text_proj = nn.Linear(self.text_width, self.embed_dim)
        self.temp = nn.parameter.Parameter(torch.ones([]) * _cfg(config, "model.temp", 0.07))
        self.itm_head = nn.Linear(self.text_width, 2)
        lw = _cfg(config, "criterion.loss_weight", {})
        self.loss_weight = SimpleNamespace(uta=_cfg(lw, "uta", 0.0), vtc=_cfg(lw, "vtc", 1.0), vtm=_cfg(lw, "vtm", 1.0), mlm=_cfg(lw, "mlm", 1.0))
        self.criterion_uta = new_UTA_Loss(_cfg(config, "criterion.distill_final_features", True), _cfg(config, "criterion.clip_loss_ratio", (1., 1.)))
        self.criterion_vtc_vtm = VTC_VTM_Loss(_cfg(config, "criterion.vtm_hard_neg", True))
        self.criterion_mlm = MLMLoss(_cfg(config, "criterion.mlm_masking_prob", 0.5), tokenizer)
        self.uta_image_only = _cfg(config, "criterion.uta_image_only", False)

    # ---- construction ----------------------------------------------------------------------------------------------------------
    def build_vision_encoder(self):
        """:296-340"""
        from . import mm_internvideo2 as mm
        name = _cfg(self.config, "model.vision_encoder.name")
        model_cfg = _cfg(self.config, "model")
        if name == "pretrain_internvideo2_1b_patch14_224":
            enc = mm.pretrain_internvideo2_1b_patch14_224(model_cfg)
        elif name == "pretrain_internvideo2_6b_patch14_224":
            enc = mm.pretrain_internvideo2_6b_patch14_224(model_cfg)
        else:
            raise ValueError(f"Not implemented: {name}")
        teacher = _cfg(self.config, "model.vision_encoder.clip_teacher")
        if teacher is not None and self.clip_teacher is None:
            assert teacher == "internvl_clip_6b"
            from .internvl_clip_vision import internvl_clip_6b
            ve = "model.vision_encoder."
            self.clip_teacher = internvl_clip_6b(img_size=_cfg(self.config, ve + "clip_input_resolution"),
                                                 clip_norm_type=_cfg(self.config, ve + "clip_norm_type"), return_attn=True,
                                                 clip_return_layer=_cfg(self.config, ve + "clip_return_layer"),
                                                 clip_return_interval=_cfg(self.config, ve + "clip_teacher_return_interval"))
            for p in self.clip_teacher.parameters():
                p.requires_grad = False
        return enc

    def _mask_parameters(self):
        """:322-338"""
        ve = "model.vision_encoder."
        img_size, num_frames = _cfg(self.config, ve + "img_size", 224), _cfg(self.config, ve + "num_frames", 8)
        tubelet, patch = _cfg(self.config, ve + "tubelet_size", 1), _cfg(self.config, ve + "patch_size", 14)
        self.clip_img_size = _cfg(self.config, ve + "clip_input_resolution", img_size)
        self.video_mask_type = _cfg(self.config, ve + "video_mask_type", "random")
        self.video_window_size = (num_frames // tubelet, img_size // patch, img_size // patch)
        self.video_mask_ratio = _cfg(self.config, ve + "video_mask_ratio", 0.8)
        self.image_mask_type = _cfg(self.config, ve + "image_mask_type", "random")
        self.image_window_size = (1, img_size // patch, img_size // patch)
        self.image_mask_ratio = _cfg(self.config, ve + "image_mask_ratio", 0.5)

    def build_text_encoder(self):
        """:342-360"""
        from .xbert import build_bert
        name = _cfg(self.config, "model.text_encoder.name", "bert_large")
        if "bert" not in name:
            raise ValueError(f"Not implemented: {name}")
        return build_bert(_cfg(self.config, "model"), self.is_pretrain, _cfg(self.config, "gradient_checkpointing", False))

    def get_text_encoder(self):
        """:362-365"""
        encoder = self.text_encoder
        return encoder.bert if hasattr(encoder, "bert") else encoder

    def freeze_vision(self):
        for p in self.vision_encoder.parameters():
            p.requires_grad = False

    def freeze_text(self):
        for p in self.text_encoder.parameters():
            p.requires_grad = False

    def no_weight_decay(self):
        ret = {"temp"}
        ret.update({"vision_encoder." + k for k in self.vision_encoder.no_weight_decay()})
        return ret

    @property
    def dtype(self):
        return self.vision_encoder.patch_embed.proj.weight.dtype

    @torch.no_grad()
    def clip_contrastive_temperature(self, min_val=0.001, max_val=0.5):
        """:291-294"""
        self.temp.clamp_(min_val, max_val)

    # ---- encoders ---------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_teacher(self, image):
        """:172-235: image (B,C,T,H,W) -> (mask (B,1+N) bool | None, visible CLIP targets | None, final CLIP target | None)"""
        from . import masking
        B, C, T, H, W = image.shape
        mask_type = self.image_mask_type if T == 1 else self.video_mask_type
        window = self.image_window_size if T == 1 else self.video_window_size
        ratio = self.image_mask_ratio if T == 1 else self.video_mask_ratio
        only_mask = (self.uta_image_only and T != 1) or _cfg(self.config, "model.vision_encoder.only_mask", False)
        if only_mask or self.clip_teacher is None or self.loss_weight.uta == 0:
            if not only_mask or mask_type == "none":
                return None, None, None
            if mask_type == "tube":
                m = masking.tube_masks(window, ratio, B, image.device)
            elif mask_type == "random":
                m = masking.random_masks(window, ratio, B, image.device)
            else:
                raise NotImplementedError(mask_type)
            return masking.with_cls_column(m), None, None
        if H != self.clip_img_size:
            image = torch.nn.functional.interpolate(image.reshape(B, C * T, H, W), size=(self.clip_img_size, self.clip_img_size),
                                                    mode="bicubic", align_corners=False).view(B, C, T, self.clip_img_size, self.clip_img_size)
        norm_clip_middle, norm_clip_final, attn = self.clip_teacher(image)
        if mask_type == "tube":
            mask = masking.with_cls_column(masking.tube_masks(window, ratio, B, image.device))
        elif mask_type == "random":
            mask = masking.with_cls_column(masking.random_masks(window, ratio, B, image.device))
        elif mask_type == "attention":
            mask = masking.attention_guided_mask(attn, B, ratio)
        else:
            raise NotImplementedError(mask_type)
        return mask, masking.gather_visible(norm_clip_middle, mask=mask), norm_clip_final

    def encode_vision(self, image, test: bool = False):
        """:237-269: image (B,T,C,H,W)"""
        T = image.shape[1]
        use_image = T == 1
        image = image.permute(0, 2, 1, 3, 4)
        if test:
            vision_embeds, pooled, _, _ = self.vision_encoder(image, None, use_image)
            return vision_embeds, pooled
        mask, tg_middle, tg_final = self.encode_teacher(image)
        vision_embeds, pooled, student_output, student_output_final = self.vision_encoder(image, mask, use_image)
        return vision_embeds, pooled, student_output, student_output_final, tg_middle, tg_final

    def encode_text(self, text):
        """:271-289 -> (text_embeds (B,L,C), pooled_text_embeds (B,C) = the [CLS] state)"""
        out = self.get_text_encoder()(text.input_ids, attention_mask=text.attention_mask, return_dict=True, mode="text")
        text_embeds = out.last_hidden_state
        return text_embeds, text_embeds[:, 0]

    # ---- training forward ---------------------------------------------------------------------------------------------------------
    # `batch_text_passes` (attribute, default False = the reference's call structure): when VTM and MLM are both on, run the text tower ONCE
    # on [ids | masked ids] (2B rows of tokens) and the fusion layers ONCE on the 3B VTM pairs + the B MLM rows, instead of two passes each.
    # Same weights, same rows, same arithmetic per row -- every loss is bit-identical to the unbatched forward given the same random draws
    # (tests/test_bert_gpu.py); only the order in which the MLM and VTM draws consume the RNG differs.  At the stage-2 batch (64 texts of 32
    # tokens) a text pass is ~700 kernels on 2048 rows: batching halves the launches and fills the GEMM waves.
    batch_text_passes = False

    def _forward_batched_text(self, image, text, idx, mlm_draws=None, neg_indices=None):
        from . import functional as Fn
        from .xbert import LinearCrossEntropyFn, right_padded_lengths
        T = image.shape[1]
        use_image = T == 1
        lw = self.loss_weight
        vision_embeds, pooled_vision_embeds, student_output, student_output_final, tg_middle, tg_final = self.encode_vision(image)
        B = text.input_ids.shape[0]
        # MLM token masking first (criterions.py:243-252), then one text-mode pass over both id sets
        ids_m, labels = self.criterion_mlm.mask(text.input_ids.clone(), self.text_encoder.config.vocab_size, text.input_ids.device,
                                                targets=text.input_ids.clone(),
                                                probability_matrix=torch.full(text.input_ids.shape, self.criterion_mlm.masking_prob,
                                                                              device=text.input_ids.device), draws=mlm_draws)
        att = text.attention_mask
        n = right_padded_lengths(att, "attention_mask")
        att2 = torch.cat([att, att], dim=0)
        att2._ivh_kv_len = None if n is None else torch.cat([n, n]).contiguous()
        bert = self.get_text_encoder()
        both = bert(torch.cat([text.input_ids, ids_m], dim=0), attention_mask=att2, return_dict=True, mode="text").last_hidden_state
        text_embeds, text_embeds_m = both[:B], both[B:]
        vision_proj = Fn.LinearFn.apply(pooled_vision_embeds, self.vision_proj.weight, self.vision_proj.bias)
        text_proj = Fn.LinearFn.apply(text_embeds[:, 0], self.text_proj.weight, self.text_proj.bias)
        zero = torch.zeros((), device=image.device)
        loss_uta = loss_vtc = zero
        if lw.uta != 0 and not (self.uta_image_only and not use_image) and tg_middle is not None:
            loss_uta = self.criterion_uta.uta_loss(student_output, student_output_final, tg_middle, tg_final)
        if lw.vtc != 0:
            loss_vtc = self.criterion_vtc_vtm.vtc_loss(vision_proj, text_proj, idx, self.temp, all_gather=True)
        # hard negatives (criterions.py:133-154), then one fusion pass: [pos | (neg video, text) | (video, neg text) | (video, masked text)]
        crit = self.criterion_vtc_vtm
        w_v2t, w_t2v, same = crit.vtm_negative_weights(vision_proj.detach(), text_proj.detach(), self.temp, idx)
        if neg_indices is not None:
            v_neg, t_neg = neg_indices
        elif crit.vtm_hard_neg:
            v_neg, t_neg = torch.multinomial(w_t2v, 1).squeeze(1), torch.multinomial(w_v2t, 1).squeeze(1)
        else:
            v_neg, t_neg = crit.get_rand_indices(same, 1).squeeze(1), crit.get_rand_indices(same, 1).squeeze(1)
        v_all = torch.cat([vision_embeds, vision_embeds[v_neg], vision_embeds, vision_embeds], dim=0)
        t_all = torch.cat([text_embeds, text_embeds, text_embeds[t_neg], text_embeds_m], dim=0)
        a_all = torch.cat([att, att, att[t_neg], att], dim=0)
        a_all._ivh_kv_len = None if n is None else torch.cat([n, n, n[t_neg], n]).contiguous()
        fused = bert(encoder_embeds=t_all, attention_mask=a_all, encoder_hidden_states=v_all, encoder_attention_mask=None, return_dict=True,
                     mode="fusion").last_hidden_state
        vtm_labels = torch.ones(3 * B, dtype=torch.int32, device=fused.device)
        vtm_labels[B:] = 0
        loss_vtm = LinearCrossEntropyFn.apply(fused[:3 * B, 0], self.itm_head.weight, self.itm_head.bias, vtm_labels, -100)
        pred = self.text_encoder.cls.predictions
        loss_mlm = LinearCrossEntropyFn.apply(pred.transform(fused[3 * B:]), pred.decoder.weight, pred.bias, labels.reshape(-1), -100)
        return dict(loss_uta=loss_uta * lw.uta, loss_vtc=loss_vtc * lw.vtc, loss_vtm=loss_vtm * lw.vtm, loss_mlm=loss_mlm * lw.mlm)

    def forward(self, image, text, idx, media_type="image"):
        """:80-170"""
        from . import functional as Fn
        self.clip_contrastive_temperature()
        if self.batch_text_passes and self.is_pretrain and self.loss_weight.vtm != 0 and self.loss_weight.mlm != 0 and hasattr(self.text_encoder, "cls"):
            return self._forward_batched_text(image, text, idx)
        T = image.shape[1]
        use_image = T == 1
        vision_embeds, pooled_vision_embeds, student_output, student_output_final, tg_middle, tg_final = self.encode_vision(image)
        text_embeds, pooled_text_embeds = self.encode_text(text)
        vision_proj = Fn.LinearFn.apply(pooled_vision_embeds, self.vision_proj.weight, self.vision_proj.bias)
        text_proj = Fn.LinearFn.apply(pooled_text_embeds, self.text_proj.weight, self.text_proj.bias)
        zero = torch.zeros((), device=image.device)
        lw = self.loss_weight
        loss_uta = loss_vtc = loss_vtm = loss_mlm = zero
        if lw.uta != 0 and not (self.uta_image_only and not use_image) and tg_middle is not None:
            loss_uta = self.criterion_uta.uta_loss(student_output, student_output_final, tg_middle, tg_final)
        if lw.vtc != 0:
            loss_vtc = self.criterion_vtc_vtm.vtc_loss(vision_proj, text_proj, idx, self.temp, all_gather=True)
        if lw.vtm != 0:
            loss_vtm = self.criterion_vtc_vtm.vtm_loss(self.get_text_encoder(), self.itm_head, self.temp, vision_embeds, text_embeds,
                                                       vision_proj, text_proj, text.attention_mask, idx)
        if self.is_pretrain and lw.mlm != 0:
            loss_mlm = self.criterion_mlm.mlm_loss(self.text_encoder, text, vision_embeds, None)
        return dict(loss_uta=loss_uta * lw.uta, loss_vtc=loss_vtc * lw.vtc, loss_vtm=loss_vtm * lw.vtm, loss_mlm=loss_mlm * lw.mlm)
