"""MI355X-native mirror of InternVideo2/multi_modality/models/backbones/internvideo2/internvideo2.py ("V:"): the stage-2 vision
encoder that `InternVideo2_Stage2_visual.build_vision_encoder` (multi_modality/models/internvideo2_stage2_visual.py:301-308) selects
by name.

`PretrainInternVideo2` (V:381-685) is the masked student trunk of single_modality without the VideoMAE branch plus
  * `mask=None` (every token kept, V:611-614) -- evaluation / retrieval forward at full sequence length,
  * image mode `use_image=True` (T = 1): the positional tables are `img_pos_embed` / `clip_img_pos_embed` when
    `sep_image_video_pos_embed`, else the video tables averaged over the frames (V:592-607, 652-667),
  * `x_vis_return_idx`: stop after block depth + idx (V:633-635),
  * a 4-tuple forward `(x_vis (B,L,D), x_pool_vis (B,768), x_clip_align (K,B,L,Cc), x_align (B,Cf))`, or `x_vis` alone (V:641-647).
State-dict keys and shapes are the reference's.  `pretrain_internvideo2_{1b,6b}_patch14_224(config)` read `config.vision_encoder.*`
exactly like V:688-760 (attribute or key access; `.get` for the three fused-op flags).
"""
from __future__ import annotations

from functools import partial

import torch
from torch import nn

from . import functional as Fn
from .internvideo2_pretrain import (AttentionPoolingBlock, Block, PatchEmbed as _PatchEmbed, PretrainInternVideo2 as _SMStudent,
                                    RMSNorm, _trunc_normal_, Linear_Decoder as _SMLinearDecoder)
from .pos_embed import get_3d_sincos_pos_embed


class PatchEmbed(_PatchEmbed):
    """V:312-345 (adds num_img_patches)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.num_img_patches = self.grid_size[1] * self.grid_size[2]


class Linear_Decoder(_SMLinearDecoder):
    """V:348-378: the multi_modality copy spells the kwarg `clip_norm_type`."""

    def __init__(self, in_channels=1408, out_channels=3200, norm_layer=nn.LayerNorm, clip_norm_type='l2'):
        super().__init__(in_channels=in_channels, out_channels=out_channels, norm_layer=norm_layer, norm_type=clip_norm_type)
        self.clip_norm_type = clip_norm_type


class PretrainInternVideo2(_SMStudent):
    """V:381-685."""

    def __init__(
            self, in_chans: int = 3, patch_size: int = 14, img_size: int = 224, qkv_bias: bool = False,
            drop_path_rate: float = 0.25, embed_dim: int = 1408, num_heads: int = 16, mlp_ratio: float = 4.3637,
            init_values: float = 1e-5, qk_normalization: bool = True, depth: int = 40,
            use_flash_attn: bool = True, use_fused_rmsnorm: bool = True, use_fused_mlp: bool = True,
            fused_mlp_heuristic: int = 1, attn_pool_num_heads: int = 16, clip_embed_dim: int = 768,
            layerscale_no_force_fp32: bool = False, num_frames: int = 8, tubelet_size: int = 1,
            sep_pos_embed: bool = False, sep_image_video_pos_embed: bool = False,
            use_checkpoint: bool = False, checkpoint_num: int = 0,
            clip_teacher_embed_dim: int = 3200, clip_teacher_final_dim: int = 768, clip_norm_type: str = 'l2',
            clip_return_layer: int = 1, clip_student_return_interval: int = 1, fused_mlp_act: str = "erf",
    ):
        nn.Module.__init__(self)
        self.num_frames, self.tubelet_size = num_frames, tubelet_size
        assert use_flash_attn == use_fused_rmsnorm == use_fused_mlp, \
            'use_flash_attn, use_fused_rmsnorm and use_fused_mlp should be consistent'
        if sep_pos_embed:
            raise NotImplementedError                                                            # V:446-447: the reference raises too
        self.use_flash_attn = use_flash_attn
        self.embed_dim, self.depth, self.num_heads = embed_dim, depth, num_heads
        self.fused_mlp_act = {"erf": "gelu_erf", "tanh": "gelu_tanh"}[fused_mlp_act]
        self.clip_norm_type = clip_norm_type
        self.return_index = [depth - int(i * clip_student_return_interval) - 1 for i in range(clip_return_layer)]
        self.clip_return_index, self.mae_return_index = self.return_index, []
        self.norm_layer_for_blocks = partial(RMSNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, num_frames=num_frames, tubelet_size=tubelet_size)
        num_patches, num_img_patches = self.patch_embed.num_patches, self.patch_embed.num_img_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.sep_pos_embed, self.sep_image_video_pos_embed = False, sep_image_video_pos_embed
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        if sep_image_video_pos_embed:                                                            # V:449-455
            self.img_pos_embed = nn.Parameter(torch.zeros(1, num_img_patches + 1, embed_dim))
        self.clip_pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        if sep_image_video_pos_embed:
            self.clip_img_pos_embed = nn.Parameter(torch.zeros(1, num_img_patches + 1, embed_dim))
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
            Linear_Decoder(in_channels=embed_dim, out_channels=clip_teacher_embed_dim, norm_layer=partial(nn.LayerNorm, eps=1e-5),
                           clip_norm_type=clip_norm_type) for _ in range(clip_return_layer)])
        self.final_clip_decoder = nn.Identity()
        if clip_teacher_final_dim > 0:
            self.final_clip_decoder = Linear_Decoder(in_channels=clip_embed_dim, out_channels=clip_teacher_final_dim,
                                                     norm_layer=partial(nn.LayerNorm, eps=1e-5), clip_norm_type=clip_norm_type)
        self.init_pos_embed()
        _trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)
        self.fix_init_weight()
        self.grad_ready_hook = None

    def init_pos_embed(self):                                                                    # V:509-533
        D = self.pos_embed.shape[-1]
        pe = get_3d_sincos_pos_embed(D, self.patch_embed.grid_size[1], self.patch_embed.grid_size[0], cls_token=True)
        self.pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        self.clip_pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        if self.sep_image_video_pos_embed:
            ipe = get_3d_sincos_pos_embed(D, self.patch_embed.grid_size[1], 1, cls_token=True)
            self.img_pos_embed.data.copy_(torch.from_numpy(ipe).float().unsqueeze(0))
            self.clip_img_pos_embed.data.copy_(torch.from_numpy(ipe).float().unsqueeze(0))

    @torch.jit.ignore
    def no_weight_decay(self):                                                                   # V:561-574
        return {'pos_embed', 'pos_embed_spatial', 'pos_embed_temporal', 'pos_embed_cls', 'img_pos_embed', 'cls_token',
                'clip_pos_embed', 'clip_pos_embed_spatial', 'clip_pos_embed_temporal', 'clip_pos_embed_cls', 'clip_img_pos_embed'}

    def _image_table(self, video_table, img_table):
        """V:592-607 / V:652-667: the (1, 1 + H*W, D) table of image mode.  A parameter-sized (<= 257 x D) reduction: plain torch,
        differentiable, so the gradient reaches the video table through autograd."""
        if self.sep_image_video_pos_embed:
            return img_table
        grid = self.patch_embed.grid_size
        n_img = grid[1] * grid[2]
        img = video_table[:, 1:, :].view(1, grid[0], n_img, self.embed_dim).mean(dim=1)
        return torch.cat([video_table[:, 0:1, :], img], dim=1)

    def forward(self, x, mask=None, use_image=False, x_vis_return_idx=-1, x_vis_only=False):
        """V:578-685.  x (B,C,T,H,W); mask (B,1+N) bool or None -> (x_vis, x_pool_vis, x_clip_align, x_align) | x_vis"""
        pos = clip_pos = None
        if use_image:
            pos = self._image_table(self.pos_embed, getattr(self, "img_pos_embed", None))
            clip_pos = self._image_table(self.clip_pos_embed, getattr(self, "clip_img_pos_embed", None))
        n_run = self.depth + int(x_vis_return_idx) + 1                                           # V:633-635 `break`
        taps, vis_idx, inv_idx, B, L = self.forward_features(x, mask, pos_embed=pos, n_blocks=n_run)
        x_last = taps[n_run - 1]
        x_vis = Fn.StreamToBf16Fn.apply(x_last, B, L)                                            # V:637-644
        if x_vis_only:
            return x_vis
        x_pool_vis = self.clip_projector(x_last, B, L)                                           # V:646
        x_align = self._final_branch(x_pool_vis)                                                 # V:647
        x_clip_align = self._clip_branch(taps, vis_idx, inv_idx, clip_pos_embed=clip_pos)        # V:650-682
        return x_vis, x_pool_vis, x_clip_align, x_align

    def forward_loss(self, *a, **k):
        raise NotImplementedError("the stage-2 encoder's losses (UTA / VTC) are assembled by the caller "
                                  "(multi_modality/models/internvideo2_stage2_visual.py:117-160); see internvideo_amd.stage2")


def _cfg_get(cfg, name, default=None):
    if hasattr(cfg, "get"):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


def _from_config(config, **dims):
    ve = config["vision_encoder"] if isinstance(config, dict) else config.vision_encoder
    g = lambda n: ve[n] if isinstance(ve, dict) else getattr(ve, n)                              # noqa: E731
    model = PretrainInternVideo2(
        in_chans=3, img_size=224, patch_size=14, clip_embed_dim=g("clip_embed_dim"), attn_pool_num_heads=16, qkv_bias=False,
        init_values=0.00001, qk_normalization=True,
        use_flash_attn=_cfg_get(ve, 'use_flash_attn', True), use_fused_rmsnorm=_cfg_get(ve, 'use_fused_rmsnorm', True),
        use_fused_mlp=_cfg_get(ve, 'use_fused_mlp', True), fused_mlp_heuristic=1, layerscale_no_force_fp32=False,
        num_frames=g("num_frames"), tubelet_size=g("tubelet_size"), sep_pos_embed=False,
        sep_image_video_pos_embed=g("sep_image_video_pos_embed"), use_checkpoint=g("use_checkpoint"), checkpoint_num=g("checkpoint_num"),
        clip_teacher_embed_dim=g("clip_teacher_embed_dim"), clip_teacher_final_dim=g("clip_teacher_final_dim"),
        clip_norm_type=g("clip_norm_type"), clip_return_layer=g("clip_return_layer"),
        clip_student_return_interval=g("clip_student_return_interval"), **dims)
    pretrained = _cfg_get(ve, "pretrained", None)
    if pretrained is not None:                                                                   # V:715-720
        state_dict = torch.load(pretrained, map_location='cpu')
        from .pos_embed import interpolate_pos_embed_internvideo2
        interpolate_pos_embed_internvideo2(state_dict, model, orig_t_size=8)
        model.load_state_dict(state_dict, strict=False)
    return model


def pretrain_internvideo2_1b_patch14_224(config):
    """V:688-723"""
    return _from_config(config, embed_dim=1408, depth=40, num_heads=16, mlp_ratio=48 / 11, drop_path_rate=0.25)


def pretrain_internvideo2_6b_patch14_224(config):
    """V:726-760"""
    return _from_config(config, embed_dim=3200, depth=48, num_heads=25, mlp_ratio=4, drop_path_rate=0.3)
