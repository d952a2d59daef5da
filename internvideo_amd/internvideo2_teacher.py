"""MI355X-native mirror of InternVideo2/single_modality/models/internvideo2_teacher.py ("T2:"): the UNMASKED InternVideo2 encoder used as the
frozen CLIP-side teacher of the distillation recipes (scripts/distillation/B14_dist_1B_stage2.sh:26 `--clip_teacher
teacher_internvideo2_stage2_1B`; run_distill.py:27,267).

`InternVideo2` (T2:350-616) runs the whole clip as ONE sequence of 1 + T*H*W tokens through the stage-1 / stage-2 vision tower and returns
    z    (K, B, 1 + T*H*W, C)  the residual stream after the `return_index` blocks, l2-normalised (T2:573-601),
    x    (B, clip_embed_dim)   the attention-pooled clip token, l2-normalised (T2:597-603),
    attn (B, T*H*W)            the pooling query's head-averaged attention over the patch tokens (T2:52-89, 606): a CLIP-LEVEL map -- one
                               multinomial draw per clip in engines/engine_for_distill.py:89-98 (the InternVL teacher's is per frame).
Same constructor keywords, state_dict keys / shapes and factory names as the reference; the arithmetic is the frozen-teacher path of
internvl_clip_vision.py on the same gfx950 kernels (no autograd, bf16 weight copies cast once)."""
from __future__ import annotations

from collections import OrderedDict
from functools import partial

import torch
from torch import nn

from . import functional as Fn
from . import ops
from .internvideo2_pretrain import AttentionPoolingBlock, Block, PatchEmbed, RMSNorm, _trunc_normal_, register_model
from .lib import InternVideoHipError
from .pos_embed import get_1d_sincos_pos_embed, get_2d_sincos_pos_embed, get_3d_sincos_pos_embed, interpolate_pos_embed


class InternVideo2(nn.Module):
    """T2:350-616."""

    def __init__(
            self, in_chans: int = 3, patch_size: int = 14, img_size: int = 224, qkv_bias: bool = False,
            drop_path_rate: float = 0.25, embed_dim: int = 1408, head_drop_path_rate: float = 0., num_heads: int = 16,
            mlp_ratio: float = 4.3637, init_values: float = 1e-5, qk_normalization: bool = True, depth: int = 40,
            use_flash_attn: bool = True, use_fused_rmsnorm: bool = True, use_fused_mlp: bool = True,
            fused_mlp_heuristic: int = 1, attn_pool_num_heads: int = 16, clip_embed_dim: int = 768,
            layerscale_no_force_fp32: bool = False, num_frames: int = 8, tubelet_size: int = 1,
            sep_pos_embed: bool = False, use_checkpoint: bool = False, checkpoint_num: int = 0,
            clip_norm_type: str = 'l2', return_attn: bool = True, clip_return_layer: int = 1, clip_return_interval: int = 1,
            clip_return_index: list = None, fused_mlp_act: str = "erf",
    ):
        super().__init__()
        assert use_flash_attn == use_fused_rmsnorm == use_fused_mlp, \
            'use_flash_attn, use_fused_rmsnorm and use_fused_mlp should be consistent'
        if clip_norm_type not in ('l2', 'none'):
            raise NotImplementedError                                                            # T2:601-602
        self.use_flash_attn = use_flash_attn
        self.embed_dim, self.depth, self.num_heads = embed_dim, depth, num_heads
        self.T = num_frames // tubelet_size
        self.fused_mlp_act = {"erf": "gelu_erf", "tanh": "gelu_tanh"}[fused_mlp_act]
        self.clip_norm_type, self.return_attn = clip_norm_type, return_attn
        if clip_return_index:                                                                    # T2:408-413
            self.return_index = list(clip_return_index)
        else:
            self.return_index = [depth - int(i * clip_return_interval) - 1 for i in range(clip_return_layer)]
        self.norm_layer_for_blocks = partial(RMSNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim, num_frames=num_frames, tubelet_size=tubelet_size)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.sep_pos_embed = bool(sep_pos_embed)
        if self.sep_pos_embed:                                                                   # T2:430-437
            grid = self.patch_embed.grid_size
            self.grid_size = grid
            self.pos_embed_spatial = nn.Parameter(torch.zeros(1, grid[1] * grid[2], embed_dim))
            self.pos_embed_temporal = nn.Parameter(torch.zeros(1, grid[0], embed_dim))
            self.pos_embed_cls = nn.Parameter(torch.zeros(1, 1, embed_dim))
        else:
            self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth, device="cpu")]
        with_cp_list = [use_checkpoint and idx < checkpoint_num for idx in range(depth)]
        self.blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias=qkv_bias, drop_path=dpr[i], init_values=init_values, attn_drop=0.,
                  use_flash_attn=use_flash_attn, use_fused_mlp=use_fused_mlp, fused_mlp_heuristic=fused_mlp_heuristic,
                  with_cp=with_cp_list[i], qk_normalization=qk_normalization,
                  layerscale_no_force_fp32=layerscale_no_force_fp32, use_fused_rmsnorm=use_fused_rmsnorm)
            for i in range(depth)])
        self.clip_projector = AttentionPoolingBlock(dim=embed_dim, num_heads=attn_pool_num_heads, qkv_bias=True,
                                                    norm_layer=partial(nn.LayerNorm, eps=1e-5), out_dim=clip_embed_dim)
        self.init_pos_embed()
        _trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)
        self.fix_init_weight()

    def init_pos_embed(self):                                                                    # T2:473-498
        g = self.patch_embed.grid_size
        if self.sep_pos_embed:
            D = self.pos_embed_spatial.shape[-1]
            self.pos_embed_spatial.data.copy_(torch.from_numpy(get_2d_sincos_pos_embed(D, g[1])).float().unsqueeze(0))
            self.pos_embed_temporal.data.copy_(torch.from_numpy(get_1d_sincos_pos_embed(D, g[0])).float().unsqueeze(0))
            return
        pe = get_3d_sincos_pos_embed(self.pos_embed.shape[-1], g[1], g[0], cls_token=True)
        self.pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))

    def _init_weights(self, m):                                                                  # T2:500-507
        if isinstance(m, nn.Linear):
            _trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def fix_init_weight(self):                                                                   # T2:509-515
        for layer_id, layer in enumerate(self.blocks):
            layer.attn.proj.weight.data.div_((2.0 * (layer_id + 1)) ** 0.5)
            layer.mlp.fc2.weight.data.div_((2.0 * (layer_id + 1)) ** 0.5)

    @property
    def dtype(self):
        return self.patch_embed.proj.weight.dtype

    def get_num_layers(self):
        return len(self.blocks)

    @torch.jit.ignore
    def no_weight_decay(self):                                                                   # T2:524-532
        return {'pos_embed', 'pos_embed_spatial', 'pos_embed_temporal', 'pos_embed_cls', 'cls_token'}

    def _pos_table(self):
        if not self.sep_pos_embed:                                                               # T2:543-560
            return self.pos_embed
        g = self.grid_size
        pos = self.pos_embed_spatial.repeat(1, g[0], 1) + torch.repeat_interleave(self.pos_embed_temporal, g[1] * g[2], dim=1)
        return torch.cat([self.pos_embed_cls.expand(pos.shape[0], -1, -1), pos], 1)

    def _bf16_weights(self):
        """frozen teacher: bf16 copies of the matrices, cast once (refreshed when a parameter's storage or version changes)"""
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if getattr(self, "_w_key", None) != key:
            for p in self.parameters():
                if p.dim() >= 2 and p.dtype != torch.bfloat16:
                    p._ivh_bf16 = p.detach().to(torch.bfloat16).reshape(p.shape[0], -1) if p.dim() > 2 else p.detach().to(torch.bfloat16)
            self._w_key = key

    def _clips_per_pass(self, L: int) -> int:
        """32-bit buffer descriptors: the widest activation of a pass (fc1 output / packed qkv, bf16) stays below 2 GiB"""
        widest = max(3 * self.embed_dim, self.blocks[0].mlp.fc1.weight.shape[0])
        rows = ((1 << 31) - (1 << 25)) // (2 * widest)
        return max(1, rows // L)

    @torch.no_grad()
    def forward(self, x):
        """x (B, C, T, H, W) -> (z, x, attn) | (z, x)   (T2:534-608)"""
        if not x.is_cuda:
            raise InternVideoHipError("InternVideo2 teacher: HBM-resident inputs only (there is no CPU path)")
        self._bf16_weights()
        pe = self.patch_embed
        L = 1 + (x.shape[2] // pe.tubelet_size) * (x.shape[3] // pe.patch_size[0]) * (x.shape[4] // pe.patch_size[1])
        cpp = self._clips_per_pass(L)
        if x.shape[0] > cpp:                                   # clips are independent sequences: groups run back to back
            outs = [self._forward_pass(x[b0:b0 + cpp]) for b0 in range(0, x.shape[0], cpp)]
            cat_dim = (1, 0, 0)
            return tuple(torch.cat([o[i] for o in outs], dim=cat_dim[i]) for i in range(len(outs[0])))
        return self._forward_pass(x)

    def _forward_pass(self, video):
        B = video.shape[0]
        pe = self.patch_embed
        x0, S, L = Fn.embed_all_tokens(video, pe.proj.weight, pe.proj.bias, self.cls_token, self._pos_table(), pe.tubelet_size, pe.patch_size[0],
                                       per_frame=False)
        taps = Fn.block_stack_infer(x0, [blk.flat_params() for blk in self.blocks], S, L, self.num_heads, 1e-6, self.fused_mlp_act,
                                    self.return_index)
        del x0
        cp, ca = self.clip_projector, self.clip_projector.cross_attn
        pooled, attn = Fn.attn_pool_infer(taps[self.depth - 1], S, L, cp.num_heads, cp.norm1_q.eps,
                                          cp.norm1_q.weight, cp.norm1_q.bias, cp.norm1_k.weight, cp.norm1_k.bias,
                                          cp.norm1_v.weight, cp.norm1_v.bias, ca.q.weight, ca.q_bias, ca.k.weight, ca.k_bias,
                                          ca.v.weight, ca.v_bias, ca.proj.weight, ca.proj.bias, want_attn=self.return_attn)
        order = [i for i in range(self.depth) if i in self.return_index]                      # the loop appends in block order (T2:566-581)
        if self.clip_norm_type == 'l2':                                                          # T2:593-598
            z = torch.stack([ops.frames_merge_l2(taps[i], B, 1, L, l2=True) for i in order])
            x = ops.frames_merge_l2(pooled, B, 1, 1, l2=True).view(B, -1)
        else:
            z = torch.stack([ops.rows_to_bf16(taps[i], S, L, 0).view(S, L, -1) for i in order])
            x = pooled
        if self.return_attn:
            return z, x, attn                                                                    # attn (B, T*H*W) fp32: T2:606 `attn[:, 0, 1:]`
        return z, x


def _teacher(embed_dim, depth, num_heads, mlp_ratio, **kw):
    return InternVideo2(img_size=224, patch_size=14, embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=mlp_ratio,
                        attn_pool_num_heads=16, clip_embed_dim=768, **kw)


def stage2_vision_state_dict(ckpt: dict, model) -> OrderedDict:
    """T2:639-656: the vision tower of a stage-2 checkpoint as this model's state_dict -- `pos_embed` resized from the stage-2 model's 4 frames
    (interpolate_pos_embed(orig_t_size=4)), `vision_encoder.` stripped, the student-side heads and positional tables dropped."""
    ckpt = dict(ckpt)
    interpolate_pos_embed(ckpt, model, orig_t_size=4)
    out = OrderedDict()
    for k, v in ckpt.items():
        if not k.startswith('vision_encoder.'):
            continue
        if 'clip_decoder' in k or 'final_clip_decoder' in k or 'clip_pos_embed' in k or 'clip_img_pos_embed' in k or 'img_pos_embed' in k:
            continue
        out[k.replace('vision_encoder.', '')] = v
    return out


@register_model
def teacher_internvideo2_1B(clip_norm_type='l2', return_attn=True, clip_return_layer=1, clip_return_interval=1, clip_return_index=None,
                            checkpoint=None, **kw):
    """T2:611-631.  `checkpoint`: path of the stage-1 1B weights (the reference's _MODELS["stage1_1B_pt"]); None = random init."""
    model = _teacher(1408, 40, 16, 48 / 11, clip_norm_type=clip_norm_type, return_attn=return_attn, clip_return_layer=clip_return_layer,
                     clip_return_interval=clip_return_interval, clip_return_index=clip_return_index, **kw)
    if checkpoint is not None:
        print(model.load_state_dict(torch.load(checkpoint, map_location='cpu')['module'], strict=False))
    return model


@register_model
def teacher_internvideo2_stage2_1B(clip_norm_type='l2', return_attn=True, clip_return_layer=1, clip_return_interval=1, clip_return_index=None,
                                   checkpoint=None, **kw):
    """T2:634-659 (scripts/distillation/B14_dist_1B_stage2.sh:26).  `checkpoint`: path of the stage-2 1B weights (_MODELS["stage2_1B_pt"])."""
    model = _teacher(1408, 40, 16, 48 / 11, clip_norm_type=clip_norm_type, return_attn=return_attn, clip_return_layer=clip_return_layer,
                     clip_return_interval=clip_return_interval, clip_return_index=clip_return_index, **kw)
    if checkpoint is not None:
        print(model.load_state_dict(stage2_vision_state_dict(torch.load(checkpoint, map_location='cpu')['module'], model), strict=False))
    return model


@register_model
def teacher_internvideo2_6B(clip_norm_type='l2', return_attn=True, clip_return_layer=1, clip_return_interval=1, clip_return_index=None,
                            checkpoint=None, **kw):
    """T2:662-682"""
    model = _teacher(3200, 48, 25, 4, clip_norm_type=clip_norm_type, return_attn=return_attn, clip_return_layer=clip_return_layer,
                     clip_return_interval=clip_return_interval, clip_return_index=clip_return_index, **kw)
    if checkpoint is not None:
        print(model.load_state_dict(torch.load(checkpoint, map_location='cpu')['module'], strict=False))
    return model
