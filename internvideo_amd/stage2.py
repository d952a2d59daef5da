"""Stage-2 video<->text contrastive seam.

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
        tval = float(temp)                                   # criterions.py passes the clamped nn.Parameter; one scalar read
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


def get_sim(vision_proj: torch.Tensor, text_proj: torch.Tensor, temp=1.0, agg_method="mean"):
    """criterions.py:15-55 for the 2-D case used by stage 2 (vision_proj [B,C], text_proj [B,C]) -> (sim_v2t, sim_t2v)."""
    if vision_proj.ndim != 2 or text_proj.ndim != 2:
        raise NotImplementedError("the MI355X path implements the pooled (2-D) features of InternVideo2 stage 2")
    _, sim, _, _, _ = ops.vtc_loss_fwd_bwd(vision_proj.float(), text_proj.float(), None, float(temp), want_grad=False)
    return sim, sim.T


class VTC_VTM_Loss(nn.Module):
    """criterions.py:58-103 (the VTC half; VTM/MLM need the BERT fusion tower: SURVEY.md 8(f) row 2)."""

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
        if vision_proj.ndim != 2 or text_proj.ndim != 2:
            raise NotImplementedError("the MI355X path implements the pooled (2-D) features of InternVideo2 stage 2")
        args = self.get_gather_args()
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
