"""The stage-1 distillation step as the reference drives it (InternVideo2/single_modality/engines/engine_for_pretraining.py:63-148,
"E:"): frozen teachers -> attention-guided mask -> visible teacher targets -> masked student step.

    videos (B, 3, 16, H, W)
      |- mae_teacher(videos)                                   (K', B, 8*h*w, Cm)          E:69-78,103   (tubelet 2: all 16 frames)
      |- videos[:, :, ::td_ratio] -> clip_teacher(...)         (K, B, 1+8*h*w, Cc), (B, Cf), attn (B*8, h*w)     E:81-101
      |- attn -> torch.multinomial -> mask (B, 1+8*h*w)                                    E:105-116
      |- targets = teacher features at the visible tokens                                   E:118-125
      `- student(videos[:, :, ::td_ratio], mask) -> three cosine losses -> backward -> AdamW   E:127-148 (+ utils.py:821-871)

Everything between the frame tensor and the loss stays on the device: the multinomial draw, the mask -> index compaction and the
target gathers are kernels / device ops, and the student step never synchronises with the host (the reference syncs at every
boolean-mask index).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import masking
from .internvideo2_pretrain import build_gather_indices
from .lib import InternVideoHipError


class Stage1Distiller:
    def __init__(self, engine, clip_teacher, mae_teacher=None, mask_type: str = "attention", mask_ratio: float = 0.8,
                 td_ratio: int = 2, generator: Optional[torch.Generator] = None):
        """engine: internvideo_amd.engine.IVTrainEngine around the student (or any object with `.model` and `train_step`);
        clip_teacher: internvl_clip_vision.InternVL_CLIP or internvideo2_teacher.InternVideo2 (return_attn=True for mask_type 'attention': a
        per-frame (B*T, H*W) or a per-clip (B, T*H*W) map -- engine_for_pretraining.py:105-116 / engine_for_distill.py:89-98); mae_teacher:
        videomae_teacher.VisionTransformer or None (distillation models without the MAE branch, engine_for_distill.py);
        td_ratio = mae_tubelet_size // tubelet_size (run_pretraining.py: the CLIP teacher and the student see every td_ratio-th frame)."""
        if mask_type not in ("attention", "tube", "random"):
            raise ValueError("mask_type must be 'attention', 'tube' or 'random'")
        self.engine, self.clip_teacher, self.mae_teacher = engine, clip_teacher, mae_teacher
        self.mask_type, self.mask_ratio, self.td_ratio, self.generator = mask_type, mask_ratio, int(td_ratio), generator

    @torch.no_grad()
    def teacher_targets(self, videos: torch.Tensor, bool_masked_pos: Optional[torch.Tensor] = None):
        """E:69-125 -> (student clip (B,3,T,H,W), mask (B,1+N) bool on the device, targets tuple, (vis_idx, inv_idx))"""
        if not videos.is_cuda:
            raise InternVideoHipError("Stage1Distiller needs HBM-resident clips: there is no CPU path")
        B = videos.shape[0]
        norm_mae = self.mae_teacher(videos) if self.mae_teacher is not None else None          # E:103 (all frames, tubelet 2)
        clip_videos = videos[:, :, ::self.td_ratio].contiguous() if self.td_ratio > 1 else videos  # E:81-82
        out = self.clip_teacher(clip_videos)                                                     # E:98-101
        if self.mask_type == "attention":
            if len(out) != 3:
                raise InternVideoHipError("mask_type 'attention' needs a clip teacher built with return_attn=True")
            norm_clip_middle, norm_clip_final, attn = out
            mask = masking.attention_guided_mask(attn, B, self.mask_ratio, generator=self.generator)    # E:105-116
        else:
            norm_clip_middle, norm_clip_final = out[0], out[1]
            if bool_masked_pos is None:
                raise ValueError("tube / random masks come with the batch (DataLoader side, datasets/masking_generator.py)")
            mask = masking.with_cls_column(bool_masked_pos.to(videos.device))                    # E:63-66
        # kept tokens per clip: from the map's LAYOUT (per frame: InternVL teacher; per clip: internvideo2_teacher.InternVideo2), no host sync
        L = int((~mask[0]).sum().item()) if self.mask_type != "attention" else masking.visible_tokens(attn.shape, B, self.mask_ratio)
        vis_idx, inv_idx = build_gather_indices(mask, videos.device, L=L, check=False)
        tg_clip = masking.gather_visible(norm_clip_middle, vis_idx=vis_idx)                      # E:118-121
        targets = [tg_clip, norm_clip_final]
        if norm_mae is not None:
            targets.append(masking.gather_visible(norm_mae, vis_idx=vis_idx, drop_cls=True))     # E:123-125
        return clip_videos, mask, tuple(targets), (vis_idx, inv_idx)

    def step(self, videos: torch.Tensor, bool_masked_pos: Optional[torch.Tensor] = None, lr: Optional[float] = None):
        """one optimizer step of E:63-199 on a batch of clips.  -> (loss, parts) device scalars"""
        clip_videos, mask, targets, vis_inv = self.teacher_targets(videos, bool_masked_pos)
        return self.engine.train_step(clip_videos, mask, targets, vis_inv=vis_inv, lr=lr)
