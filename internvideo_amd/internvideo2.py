"""MI355X-native mirror of InternVideo2/single_modality/models/internvideo2.py ("F:"): the fine-tuning / inference classifier
 -- the student trunk at FULL sequence length (no mask: L = 1 + T*H*W = 2049 tokens for 8 x 224^2, the
regime where the attention kernels dominate), the attention-pool projector, `fc_norm` and the classification `head`.

    InternVideo2.forward(x (B, C, T, H, W)) -> logits (B, num_classes)                           (F:500-543)

Same class / registry names (`internvideo2_{small,base,large,1B,6B}_patch14_224`), constructor kwargs and state_dict keys / shapes
as the reference, so fine-tuning checkpoints load with strict=True and `run_finetuning.py` + `optim_factory.py` (layer-wise lr decay
reads `get_num_layers()` and the parameter names) run unchanged.  Forward and backward run in the same gfx950 kernels as pre-training.
"""
from __future__ import annotations

from functools import partial

import torch
import torch.nn.functional as F
from torch import nn

from . import functional as Fn
from .internvideo2_pretrain import AttentionPoolingBlock, Block, PatchEmbed, PretrainInternVideo2, RMSNorm, _trunc_normal_, register_model


class InternVideo2(PretrainInternVideo2):
    """F:337-543.  Inherits the token / block machinery of the pre-training student (forward_features with mask=None)."""

    def __init__(
            self, in_chans: int = 3, patch_size: int = 14, img_size: int = 224, qkv_bias: bool = False,
            drop_path_rate: float = 0.25, embed_dim: int = 1408, head_drop_path_rate: float = 0., num_heads: int = 16,
            mlp_ratio: float = 4.3637, init_values: float = 1e-5, qk_normalization: bool = True, depth: int = 40,
            use_flash_attn: bool = True, use_fused_rmsnorm: bool = True, use_fused_mlp: bool = True,
            fused_mlp_heuristic: int = 1, attn_pool_num_heads: int = 16, clip_embed_dim: int = 768,
            layerscale_no_force_fp32: bool = False, num_frames: int = 8, tubelet_size: int = 1,
            sep_pos_embed: bool = False, use_checkpoint: bool = False, checkpoint_num: int = 0,
            fc_drop_rate: float = 0., num_classes: int = 1000, init_scale: float = 0.001, fused_mlp_act: str = "erf",
    ):
        nn.Module.__init__(self)
        assert use_flash_attn == use_fused_rmsnorm == use_fused_mlp, \
            'use_flash_attn, use_fused_rmsnorm and use_fused_mlp should be consistent'
        self.use_flash_attn = use_flash_attn
        self.embed_dim, self.depth, self.num_heads = embed_dim, depth, num_heads
        self.fused_mlp_act = {"erf": "gelu_erf", "tanh": "gelu_tanh"}[fused_mlp_act]
        self.clip_return_index, self.mae_return_index = [], []
        self.norm_layer_for_blocks = partial(RMSNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, num_frames=num_frames, tubelet_size=tubelet_size)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.sep_pos_embed = bool(sep_pos_embed)
        if self.sep_pos_embed:                                 # F:390-397: spatial + temporal + cls tables, joined on the fly in forward (F:510-525)
            grid = self.patch_embed.grid_size
            self.grid_size = grid
            self.pos_embed_spatial = nn.Parameter(torch.zeros(1, grid[1] * grid[2], embed_dim))
            self.pos_embed_temporal = nn.Parameter(torch.zeros(1, grid[0], embed_dim))
            self.pos_embed_cls = nn.Parameter(torch.zeros(1, 1, embed_dim))
        else:
            self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches + 1, embed_dim))
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
        self.fc_norm = nn.LayerNorm(clip_embed_dim)
        self.fc_dropout = nn.Dropout(p=fc_drop_rate) if fc_drop_rate > 0 else nn.Identity()
        self.head = nn.Linear(clip_embed_dim, num_classes)
        self.init_pos_embed()
        _trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)
        self.fix_init_weight()
        self.head.weight.data.mul_(init_scale)                                                   # F:449-450
        self.head.bias.data.mul_(init_scale)
        self.grad_ready_hook = None

    def init_pos_embed(self):                                                                    # F:452-475
        from .pos_embed import get_1d_sincos_pos_embed, get_2d_sincos_pos_embed, get_3d_sincos_pos_embed
        if self.sep_pos_embed:                                 # F:454-465 (the cls table stays zero)
            D = self.pos_embed_spatial.shape[-1]
            self.pos_embed_spatial.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(D, self.patch_embed.grid_size[1])).float().unsqueeze(0))
            self.pos_embed_temporal.data.copy_(torch.from_numpy(get_1d_sincos_pos_embed(D, self.patch_embed.grid_size[0])).float().unsqueeze(0))
            return
        pe = get_3d_sincos_pos_embed(self.pos_embed.shape[-1], self.patch_embed.grid_size[1], self.patch_embed.grid_size[0], cls_token=True)
        self.pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))

    @torch.jit.ignore
    def no_weight_decay(self):                                                                   # F:490-498
        return {'pos_embed', 'pos_embed_spatial', 'pos_embed_temporal', 'pos_embed_cls', 'cls_token'}

    def forward(self, x):
        """F:500-543 -> logits (B, num_classes) bf16"""
        taps, _, _, B, L = self.forward_features(x, None)
        pooled = self.clip_projector(taps[self.depth - 1], B, L)                                 # F:538
        h = Fn.LayerNormFn.apply(pooled, self.fc_norm.weight, self.fc_norm.bias, self.fc_norm.eps)   # F:539
        h = self.fc_dropout(h)                                                                   # F:540 (a (B, 768) tensor)
        w, b = self.head.weight, self.head.bias
        n = w.shape[0]
        if n % 8:                                          # class counts like 174 / 339: the GEMM moves 8-column chunks -> zero-pad
            pad = 8 - n % 8
            w, b = F.pad(w, (0, 0, 0, pad)), F.pad(b, (0, pad))
        return Fn.LinearFn.apply(h, w, b)[:, :n]

    def forward_loss(self, *a, **k):
        raise NotImplementedError("fine-tuning losses (cross-entropy / mixup) are the caller's (engines/engine_for_finetuning.py)")


def _ft(embed_dim, depth, num_heads, mlp_ratio, **kwargs):
    return InternVideo2(img_size=224, patch_size=14, embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=mlp_ratio,
                        attn_pool_num_heads=16, clip_embed_dim=768, **kwargs)


@register_model
def internvideo2_small_patch14_224(pretrained=False, **kwargs):
    """F:546-554"""
    return _ft(384, 12, 6, 4, **kwargs)


@register_model
def internvideo2_base_patch14_224(pretrained=False, **kwargs):
    """F:557-565"""
    return _ft(768, 12, 12, 4, **kwargs)


@register_model
def internvideo2_large_patch14_224(pretrained=False, **kwargs):
    """F:568-576"""
    return _ft(1024, 24, 16, 4, **kwargs)


@register_model
def internvideo2_1B_patch14_224(pretrained=False, **kwargs):
    """F:579-587"""
    return _ft(1408, 40, 16, 48 / 11, **kwargs)


@register_model
def internvideo2_6B_patch14_224(pretrained=False, **kwargs):
    """F:590-598"""
    return _ft(3200, 48, 25, 4, **kwargs)
