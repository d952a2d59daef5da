"""MI355X-native mirror of InternVideo2/single_modality/models/internvl_clip_vision.py ("T:"): the frozen InternVL-6B CLIP teacher
of stage-1 pre-training, forward only.

The teacher is the student's block (RMSNorm pre-norm, q/k RMSNorm over the full width, LayerScale, erf-GELU MLP; T:157-300) run on
every frame as its own sequence of 1 + H*W tokens (T:415-421), so it reuses the student's gfx950 kernels unchanged (hd = 128 for the
6B model).  What it adds: the frame merge + l2 normalisation of the tapped features (T:445-453), the frame-averaged pooled feature
(T:455-456) and the head-averaged attention map of the pooling query over the patches (T:82-83,463) that drives attention-guided
masking.  Same constructor kwargs, parameter names / shapes and return tuple as the reference:

    z (K, B, 1 + T*H*W, C) l2-normalised, x (B, clip_embed_dim) l2-normalised[, attn (B*T, H*W) fp32]

Nothing is saved for backward (the teacher runs under no_grad, engines/engine_for_pretraining.py:69-103): every activation is
released as soon as it is consumed, so a whole batch of frames goes through in one pass.
"""
from __future__ import annotations

from functools import partial

import torch
from torch import nn

from . import functional as Fn
from . import ops
from .internvideo2_pretrain import AttentionPoolingBlock, Block, RMSNorm, _trunc_normal_
from .lib import InternVideoHipError


class PatchEmbed(nn.Module):
    """T:303-333: Conv3d k = s = (1, p, p) parameters; arithmetic in functional.embed_all_tokens."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True):
        super().__init__()
        img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.img_size, self.patch_size = img_size, patch_size
        self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=(1, patch_size[0], patch_size[1]), stride=(1, patch_size[0], patch_size[1]))
        self.norm = nn.Identity()


class InternVL_CLIP(nn.Module):
    """T:336-465."""

    def __init__(
            self, in_chans: int = 3, patch_size: int = 14, img_size: int = 224, qkv_bias: bool = False,
            drop_path_rate: float = 0.2, embed_dim: int = 3200, num_heads: int = 25, mlp_ratio: int = 4,
            init_values: float = 0.1, qk_normalization: bool = True, depth: int = 48,
            use_flash_attn: bool = True, use_fused_rmsnorm: bool = True, use_fused_mlp: bool = True,
            fused_mlp_heuristic: int = 1, with_cp: bool = False, attn_pool_num_heads: int = 16, clip_embed_dim: int = 768,
            layerscale_no_force_fp32: bool = True, clip_norm_type: str = 'l2', return_attn: bool = True,
            clip_return_layer: int = 1, clip_return_interval: int = 1, fused_mlp_act: str = "erf",
    ):
        super().__init__()
        assert use_flash_attn == use_fused_rmsnorm == use_fused_mlp, \
            'use_flash_attn, use_fused_rmsnorm and use_fused_mlp should be consistent'
        if clip_norm_type not in ('l2', 'none'):
            raise NotImplementedError                                                            # T:457-460
        self.use_flash_attn = use_flash_attn
        self.embed_dim, self.depth, self.num_heads = embed_dim, depth, num_heads
        self.fused_mlp_act = {"erf": "gelu_erf", "tanh": "gelu_tanh"}[fused_mlp_act]
        self.clip_norm_type, self.return_attn = clip_norm_type, return_attn
        self.return_index = [depth - int(i * clip_return_interval) - 1 for i in range(clip_return_layer)]
        self.norm_layer_for_blocks = partial(RMSNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.num_patches = self.patch_embed.num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + 1, embed_dim), requires_grad=False)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth, device="cpu")]
        self.blocks = nn.ModuleList([
            Block(embed_dim, num_heads, mlp_ratio, qkv_bias=qkv_bias, drop_path=dpr[i], init_values=init_values, attn_drop=0.,
                  use_flash_attn=use_flash_attn, use_fused_mlp=use_fused_mlp, fused_mlp_heuristic=fused_mlp_heuristic,
                  with_cp=with_cp, qk_normalization=qk_normalization, layerscale_no_force_fp32=layerscale_no_force_fp32,
                  use_fused_rmsnorm=use_fused_rmsnorm)
            for i in range(depth)])
        self.clip_projector = AttentionPoolingBlock(dim=embed_dim, num_heads=attn_pool_num_heads, qkv_bias=True,
                                                    norm_layer=partial(nn.LayerNorm, eps=1e-5), out_dim=clip_embed_dim)

    @property
    def dtype(self):
        return self.patch_embed.proj.weight.dtype

    def _bf16_weights(self):
        """the teacher is frozen: cast its matrices to bf16 once and let the GEMMs read the copies (refreshed when a parameter's
        storage or version counter changes: .to() moves storage, load_state_dict / in-place writes bump _version)"""
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())   # _version: load_state_dict copies IN PLACE
        if getattr(self, "_w_key", None) != key:
            for p in self.parameters():
                if p.dim() >= 2 and p.dtype != torch.bfloat16:
                    p._ivh_bf16 = p.detach().to(torch.bfloat16).reshape(p.shape[0], -1) if p.dim() > 2 else p.detach().to(torch.bfloat16)
            self._w_key = key

    def _clips_per_pass(self, T: int) -> int:
        """the GEMM kernels address operands through 32-bit buffer descriptors (< 2 GiB per operand): the widest activation of a pass
        (fc1 output / packed qkv, bf16) bounds the frame sequences one pass may carry.  6B: 12800 columns -> 83k rows -> 40 clips of 8 frames"""
        widest = max(3 * self.embed_dim, self.blocks[0].mlp.fc1.weight.shape[0])
        rows = ((1 << 31) - (1 << 25)) // (2 * widest)
        return max(1, rows // (T * (self.num_patches + 1)))

    @torch.no_grad()
    def forward(self, image):
        """image (B, C, T, H, W) -> (z, x, attn) | (z, x)   (T:411-465)"""
        if not image.is_cuda:
            raise InternVideoHipError("InternVL_CLIP.forward needs HBM-resident inputs: there is no CPU path")
        self._bf16_weights()
        cpp = self._clips_per_pass(image.shape[2])
        if image.shape[0] > cpp:                               # frames are independent sequences: run clip groups back to back
            outs = [self._forward_pass(image[b0:b0 + cpp]) for b0 in range(0, image.shape[0], cpp)]
            cat_dim = (1, 0, 0)                                # z (K, B, ., C) | x (B, C) | attn (B*T, HW)
            return tuple(torch.cat([o[i] for o in outs], dim=cat_dim[i]) for i in range(len(outs[0])))
        return self._forward_pass(image)

    def _forward_pass(self, image):
        B, T = image.shape[0], image.shape[2]
        pe = self.patch_embed
        x0, S, L = Fn.embed_all_tokens(image, pe.proj.weight, pe.proj.bias, self.cls_token, self.pos_embed, 1, pe.patch_size[0],
                                       per_frame=True)
        # `fp8_gemm` (attribute, default False; the reference runs its teachers in bf16): the frozen blocks' GEMMs on the e4m3 MFMA path
        taps = Fn.block_stack_infer(x0, [blk.flat_params() for blk in self.blocks], S, L, self.num_heads, 1e-6,
                                    self.fused_mlp_act, self.return_index, fp8=bool(getattr(self, "fp8_gemm", False)))
        del x0
        cp, ca = self.clip_projector, self.clip_projector.cross_attn
        pooled, attn = Fn.attn_pool_infer(taps[self.depth - 1], S, L, cp.num_heads, cp.norm1_q.eps,
                                          cp.norm1_q.weight, cp.norm1_q.bias, cp.norm1_k.weight, cp.norm1_k.bias,
                                          cp.norm1_v.weight, cp.norm1_v.bias, ca.q.weight, ca.q_bias, ca.k.weight, ca.k_bias,
                                          ca.v.weight, ca.v_bias, ca.proj.weight, ca.proj.bias, want_attn=self.return_attn)
        l2 = self.clip_norm_type == 'l2'
        if l2:       # T:445-456: merge the frames (cls rows averaged), l2-normalise; pooled feature: mean over frames, l2
            z = torch.stack([ops.frames_merge_l2(taps[i], B, T, L, l2=True) for i in sorted(self.return_index)])
            x = ops.frames_merge_l2(pooled, B, T, 1, l2=True).view(B, -1)
        else:        # T:457-458 'none': the per-frame features as they are (list -> stacked (K, B*T, L, C)), pooled (B*T, C)
            z = torch.stack([ops.rows_to_bf16(taps[i], S, L, 0).view(S, L, -1) for i in sorted(self.return_index)])
            x = pooled
        if self.return_attn:
            return z, x, attn                                                                    # attn (B*T, H*W) fp32
        return z, x


def internvl_clip_6b(img_size, clip_norm_type='l2', return_attn=True, clip_return_layer=1, clip_return_interval=1, checkpoint=None):
    """T:506-526.  `checkpoint` (path to the InternVL-C weights, the reference's _MODELS["internvl_c_13b_224px"]): loaded through
    `process_checkpoint` (patch-embed inflation + bicubic positional interpolation, T:469-503); None = random init."""
    model = InternVL_CLIP(img_size=img_size, layerscale_no_force_fp32=False, clip_norm_type=clip_norm_type, return_attn=return_attn,
                          clip_return_layer=clip_return_layer, clip_return_interval=clip_return_interval)
    if checkpoint is not None:
        ckpt = torch.load(checkpoint, map_location='cpu')
        model.load_state_dict(process_checkpoint(ckpt, model), strict=False)
    return model.eval()


def inflate_weight(weight_2d, time_dim, center=True):
    """T:469-479: a (D, C, p, p) image kernel as a (D, C, t, p, p) tubelet kernel."""
    if center:
        w3 = torch.zeros(*weight_2d.shape).unsqueeze(2).repeat(1, 1, time_dim, 1, 1)
        w3[:, :, time_dim // 2, :, :] = weight_2d
        return w3
    return weight_2d.unsqueeze(2).repeat(1, 1, time_dim, 1, 1) / time_dim


def process_checkpoint(ckpt, model):
    """T:482-503: take ckpt['module'], inflate the patch-embed kernel when its shape differs, bicubically resize pos_embed."""
    target = model.state_dict()
    out = {}
    for k, v in ckpt['module'].items():
        if 'patch_embed' in k and k in target and v.shape != target[k].shape:
            v = inflate_weight(v, target[k].shape[2])
        out[k] = v
    pe = out['pos_embed']
    D = pe.shape[-1]
    s_old, s_new = int((pe.shape[-2] - 1) ** 0.5), int(model.num_patches ** 0.5)
    if s_old != s_new:
        grid = pe[:, 1:].reshape(-1, s_old, s_old, D).permute(0, 3, 1, 2)
        grid = torch.nn.functional.interpolate(grid, size=(s_new, s_new), mode='bicubic', align_corners=False)
        out['pos_embed'] = torch.cat((pe[:, :1], grid.permute(0, 2, 3, 1).flatten(0, 2).unsqueeze(0)), dim=1)
    return out
