"""MI355X-native mirror of InternVideo2/single_modality/models/internvideo2_distill.py ("D:"): the S/B/L-14 distillation students.

`DistInternVideo2` (D:417-697) is the masked student of internvideo2_pretrain.py without the VideoMAE branch: same trunk (tubelet
patch embed of the visible tokens, fused-residual blocks, attention-pool projector), `clip_return_layer` decoders chosen by name
(`clip_student_decoder` in {'Linear_Decoder', 'MLP_Decoder'}, D:17-21,412-414), optional explicit `clip_student_return_index`, and
a 2-tuple forward `(x_clip_align (K,B,L,Cc), x_align (B,Cf))` (D:612-697).  Same state_dict keys / shapes as the reference; all
arithmetic runs in the gfx950 kernels shared with the 1B/6B pre-training student.
"""
from __future__ import annotations

from functools import partial

import torch
from torch import nn

from .internvideo2_pretrain import (AttentionPoolingBlock, Block, Linear_Decoder, MLP_Decoder, PatchEmbed, PretrainInternVideo2,
                                    RMSNorm, _trunc_normal_, register_model)
from .pos_embed import get_3d_sincos_pos_embed

DECODER_REGISTRY = {'Linear_Decoder': Linear_Decoder, 'MLP_Decoder': MLP_Decoder}       # D:17-21,412-414


class DistInternVideo2(PretrainInternVideo2):
    """D:417-697.  Inherits the forward machinery of the pre-training student; only construction and the output tuple differ."""

    def __init__(
            self, in_chans: int = 3, patch_size: int = 14, img_size: int = 224, qkv_bias: bool = False,
            drop_path_rate: float = 0.05, embed_dim: int = 384, num_heads: int = 6, mlp_ratio: float = 4,
            init_values: float = 1e-5, qk_normalization: bool = True, depth: int = 12,
            use_flash_attn: bool = True, use_fused_rmsnorm: bool = True, use_fused_mlp: bool = True,
            fused_mlp_heuristic: int = 1, attn_pool_num_heads: int = 16, clip_embed_dim: int = 768,
            layerscale_no_force_fp32: bool = False, num_frames: int = 8, tubelet_size: int = 1,
            sep_pos_embed: bool = False, use_checkpoint: bool = False, checkpoint_num: int = 0,
            clip_teacher_embed_dim: int = 3200, clip_teacher_final_dim: int = 768, clip_norm_type: str = 'l2',
            clip_return_layer: int = 1, clip_student_return_interval: int = 1, clip_student_return_index: list = None,
            clip_student_decoder: str = 'Linear_Decoder', fused_mlp_act: str = "erf",
    ):
        nn.Module.__init__(self)
        assert use_flash_attn == use_fused_rmsnorm == use_fused_mlp, \
            'use_flash_attn, use_fused_rmsnorm and use_fused_mlp should be consistent'
        if clip_student_decoder not in DECODER_REGISTRY:
            raise KeyError(f"clip_student_decoder must be one of {sorted(DECODER_REGISTRY)} (D:17-21)")
        self.use_flash_attn = use_flash_attn
        self.embed_dim, self.depth, self.num_heads = embed_dim, depth, num_heads
        self.fused_mlp_act = {"erf": "gelu_erf", "tanh": "gelu_tanh"}[fused_mlp_act]
        self.clip_norm_type = clip_norm_type
        if clip_student_return_index:                                                            # D:462-466
            self.clip_return_index = list(clip_student_return_index)
        else:
            self.clip_return_index = [depth - int(i * clip_student_return_interval) - 1 for i in range(clip_return_layer)]
        self.mae_return_index = []
        self.norm_layer_for_blocks = partial(RMSNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, num_frames=num_frames, tubelet_size=tubelet_size)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.sep_pos_embed = bool(sep_pos_embed)
        if self.sep_pos_embed:                                 # D:481-494: spatial + temporal (+ cls) tables, joined on the fly (D:622-637, 677-692)
            grid = self.patch_embed.grid_size
            self.grid_size = grid
            for pre in ("", "clip_"):
                setattr(self, pre + "pos_embed_spatial", nn.Parameter(torch.zeros(1, grid[1] * grid[2], embed_dim)))
                setattr(self, pre + "pos_embed_temporal", nn.Parameter(torch.zeros(1, grid[0], embed_dim)))
                setattr(self, pre + "pos_embed_cls", nn.Parameter(torch.zeros(1, 1, embed_dim)))
        else:
            self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
            self.clip_pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
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
        dec = DECODER_REGISTRY[clip_student_decoder]
        self.clip_decoder = nn.ModuleList([
            dec(in_channels=embed_dim, out_channels=clip_teacher_embed_dim, norm_layer=partial(nn.LayerNorm, eps=1e-5),
                norm_type=clip_norm_type) for _ in range(clip_return_layer)])
        self.final_clip_decoder = nn.Identity()
        if clip_teacher_final_dim > 0:
            self.final_clip_decoder = dec(in_channels=clip_embed_dim, out_channels=clip_teacher_final_dim,
                                          norm_layer=partial(nn.LayerNorm, eps=1e-5), norm_type=clip_norm_type)
        self.init_pos_embed()
        _trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)
        self.fix_init_weight()
        self.grad_ready_hook = None

    def init_pos_embed(self):                                                                    # D:549-573
        if self.sep_pos_embed:                                 # D:551-563 (the cls tables stay zero)
            from .pos_embed import get_1d_sincos_pos_embed, get_2d_sincos_pos_embed
            D = self.pos_embed_spatial.shape[-1]
            sp = torch.from_numpy(get_2d_sincos_pos_embed(D, self.patch_embed.grid_size[1])).float().unsqueeze(0)
            tm = torch.from_numpy(get_1d_sincos_pos_embed(D, self.patch_embed.grid_size[0])).float().unsqueeze(0)
            for pre in ("", "clip_"):
                getattr(self, pre + "pos_embed_spatial").data.copy_(sp)
                getattr(self, pre + "pos_embed_temporal").data.copy_(tm)
            return
        pe = get_3d_sincos_pos_embed(self.pos_embed.shape[-1], self.patch_embed.grid_size[1], self.patch_embed.grid_size[0], cls_token=True)
        self.pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        self.clip_pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))

    @torch.jit.ignore
    def no_weight_decay(self):                                                                   # D:598-610
        return {'pos_embed', 'pos_embed_spatial', 'pos_embed_temporal', 'pos_embed_cls', 'cls_token',
                'clip_pos_embed', 'clip_pos_embed_spatial', 'clip_pos_embed_temporal', 'clip_pos_embed_cls'}

    def forward(self, x, mask=None):
        """D:612-697 -> (x_clip_align (K,B,L,Cc), x_align (B,Cf))"""
        if mask is None:
            raise ValueError("DistInternVideo2.forward needs the (B, 1+N) mask (D:641 `x[~mask]`)")
        taps, vis_idx, inv_idx, B, L = self.forward_features(x, mask, bf16_taps=True)
        pooled = self.clip_projector(taps[self.depth - 1], B, L)                                 # D:665
        return self._clip_branch(taps, vis_idx, inv_idx), self._final_branch(pooled)

    def forward_loss(self, x, mask, targets, clip_loss_ratio=(1.0, 1.0), mae_loss_ratio=0.0, vis_inv=None):
        """Student forward + the loss of engines/engine_for_distill.py:107-121 with the decoder tails fused.
        targets = (clip_middle (K,B,L,Cc), clip_final (B,Cf) | None).  -> (loss, (loss_clip_middle, loss_clip_final))"""
        tg_clip, tg_final = targets[0], (targets[1] if len(targets) > 1 else None)
        taps, vis_idx, inv_idx, B, L = self.forward_features(x, mask, vis_inv, bf16_taps=True)
        pooled = self.clip_projector(taps[self.depth - 1], B, L)
        l_clip = self._clip_branch(taps, vis_idx, inv_idx, targets=tg_clip) / float(tg_clip.shape[0] * B * L)
        if tg_final is not None and clip_loss_ratio[1] > 0 and not isinstance(self.final_clip_decoder, nn.Identity):
            l_final = self._final_branch(pooled, tg_final) / float(B)
        else:                                                  # engine_for_distill.py:111-112: zeros when the final feature is not distilled
            l_final = torch.zeros(1, dtype=torch.float32, device=l_clip.device)
        loss = l_clip * clip_loss_ratio[0] + l_final * clip_loss_ratio[1]
        return loss.reshape(()), (l_clip.reshape(()), l_final.reshape(()))


@register_model
def distill_internvideo2_small_patch14_224(pretrained=False, **kwargs):
    """D:700-708"""
    return DistInternVideo2(img_size=224, patch_size=14, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4,
                            attn_pool_num_heads=16, clip_embed_dim=768, **kwargs)


@register_model
def distill_internvideo2_base_patch14_224(pretrained=False, **kwargs):
    """D:711-719"""
    return DistInternVideo2(img_size=224, patch_size=14, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4,
                            attn_pool_num_heads=16, clip_embed_dim=768, **kwargs)


@register_model
def distill_internvideo2_large_patch14_224(pretrained=False, **kwargs):
    """D:722-730"""
    return DistInternVideo2(img_size=224, patch_size=14, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4,
                            attn_pool_num_heads=16, clip_embed_dim=768, **kwargs)
