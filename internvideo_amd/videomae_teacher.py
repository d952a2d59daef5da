"""MI355X-native mirror of InternVideo2/single_modality/models/videomae.py ("VT:"): the frozen VideoMAEv2-g teacher of stage-1
pre-training, forward only, on the LayerNorm-block kernels of the VideoMAE pixel path.

    VisionTransformer.forward(x (B,3,T,H,W), mask (B,N) bool | None) -> (K, B, N | N_vis, C) l2-normalised tapped features

ATTENTION SEMANTICS.  VT:91-96 builds q, k, v as (B, H, N, hd) (`permute(2, 0, 3, 1, 4)`, inherited from the matmul formulation of
InternVideo1/Pretrain/VideoMAE/modeling_finetune.py:104-129) and passes them to `flash_attn_func`, whose contract is
(batch, seqlen, nheads, headdim).  As coded, therefore, every token attends over its own H head slots and the result is re-read
as (B, N, H*hd).  `attn_semantics="reference"` (default) reproduces exactly that -- parity with the reference as written is the
gate, and the strided attention kernel runs the layout in place; `attn_semantics="standard"` computes token-to-token attention
(what the VideoMAEv2 checkpoint was trained with).  flash_attn is not installed in the authoring container, so the fixtures for
this file come from the reference module run with a `flash_attn_func` stand-in that follows flash_attn's documented contract
(tests/golden/ref_loader.py): "parity unpinned" for the fused call itself, pinned for everything around it.
"""
from __future__ import annotations

import os
from functools import partial

import numpy as np
import torch
from torch import nn

from . import functional as Fn
from . import ops
from .lib import InternVideoHipError
from .videomae_pretrain import Block, PatchEmbed, mae_gather_indices

MODEL_PATH = os.environ.get('INTERNVIDEO2_MODEL_PATH', 'your_model_path') + '/videomae'
_MODELS = {"vit_g14_hybrid": os.path.join(MODEL_PATH, "vit_g_hybrid_1200e_pre.pth")}


def get_sinusoid_encoding_table(n_position, d_hid, cur_frame=-1, pre_n_position=1568):
    """VT:159-205: the checkpoint's sinusoid table (pre_n_position rows = 8 frames x 14x14 or 16x16), bicubically resized in space
    and linearly in time when the model's grid differs; a plain tensor when nothing changed, else a learnable Parameter."""
    j = np.arange(d_hid)
    tab = np.arange(pre_n_position, dtype=np.float64)[:, None] / np.power(10000, 2 * (j // 2) / d_hid)[None, :]
    tab[:, 0::2] = np.sin(tab[:, 0::2])
    tab[:, 1::2] = np.cos(tab[:, 1::2])
    tab = torch.tensor(tab, dtype=torch.float, requires_grad=False).unsqueeze(0)
    F = torch.nn.functional
    if cur_frame != -1 and n_position // cur_frame * 8 != pre_n_position:                        # spatial resize (VT:173-186)
        T, C = 8, d_hid
        P = int((pre_n_position // T) ** 0.5)          # 14 for the 1568-row table (hard-coded 14 in VT:174; 16 for the 2048-row one)
        new_P = int((n_position // cur_frame) ** 0.5)
        t = tab.reshape(-1, T, P, P, C).reshape(-1, P, P, C).permute(0, 3, 1, 2)
        t = F.interpolate(t, size=(new_P, new_P), mode='bicubic', align_corners=False)
        tab = t.permute(0, 2, 3, 1).reshape(-1, T, new_P, new_P, C).flatten(1, 3)
    if cur_frame != -1 and cur_frame != 8:                                                       # temporal resize (VT:187-199)
        T, new_T, C = 8, cur_frame, d_hid
        P = int((n_position // cur_frame) ** 0.5)
        t = tab.reshape(-1, T, P, P, C).permute(0, 2, 3, 4, 1).reshape(-1, C, T)
        t = F.interpolate(t, size=new_T, mode='linear')
        tab = t.reshape(1, P, P, C, new_T).permute(0, 4, 1, 2, 3).flatten(1, 3)
    if n_position == pre_n_position:
        return tab
    return nn.Parameter(tab, requires_grad=True)


class VisionTransformer(nn.Module):
    """VT:207-312."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4., qkv_bias=False,
                 qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0., norm_layer=nn.LayerNorm, init_values=0.,
                 all_frames=16, tubelet_size=2, mae_norm_type='l2', mae_return_layer=1, mae_return_interval=1,
                 attn_semantics: str = "reference"):
        super().__init__()
        if mae_norm_type not in ('l2', 'none'):
            raise NotImplementedError                                                            # VT:309-310
        if attn_semantics not in ("reference", "standard"):
            raise ValueError("attn_semantics must be 'reference' (videomae.py:91-96 as coded) or 'standard'")
        if drop_rate:
            raise InternVideoHipError("VideoMAE teacher (MI355X): dropout is not implemented (the teacher runs with 0)")
        self.mae_norm_type, self.attn_semantics = mae_norm_type, attn_semantics
        self.return_index = [depth - int(i * mae_return_interval) - 1 for i in range(mae_return_layer)]
        self.tubelet_size, self.depth, self.embed_dim, self.num_heads = tubelet_size, depth, embed_dim, num_heads
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      num_frames=all_frames, tubelet_size=tubelet_size)
        num_patches = self.patch_embed.num_patches
        self.pos_embed = get_sinusoid_encoding_table(num_patches, embed_dim, all_frames // tubelet_size,
                                                     pre_n_position=2048 if patch_size == 14 else 1568)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth, device="cpu")]
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                  attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer, init_values=init_values) for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.apply(self._init_weights)

    def _init_weights(self, m):                                                                  # VT:270-277
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02, a=-2., b=2.)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def get_num_layers(self):
        return len(self.blocks)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def _bf16_weights(self):
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())   # _version: load_state_dict copies IN PLACE
        if getattr(self, "_w_key", None) != key:
            for p in self.parameters():
                if p.dim() >= 2 and p.dtype != torch.bfloat16:
                    p._ivh_bf16 = p.detach().to(torch.bfloat16).reshape(p.shape[0], -1) if p.dim() > 2 else p.detach().to(torch.bfloat16)
            self._w_key = key

    @torch.no_grad()
    def forward(self, x, mask=None):
        """VT:285-312"""
        if not x.is_cuda:
            raise InternVideoHipError("VideoMAE teacher forward needs HBM-resident inputs: there is no CPU path")
        self._bf16_weights()
        # < 2 GiB per GEMM operand (32-bit buffer descriptors): bound the clips per pass by the widest activation (fc1 output / qkv)
        widest = max(3 * self.embed_dim, self.blocks[0].mlp.fc1.weight.shape[0])
        cpp = max(1, (((1 << 31) - (1 << 25)) // (2 * widest)) // self.patch_embed.num_patches)
        if x.shape[0] > cpp:
            return torch.cat([self._forward_pass(x[b0:b0 + cpp], None if mask is None else mask[b0:b0 + cpp])
                              for b0 in range(0, x.shape[0], cpp)], dim=1)
        return self._forward_pass(x, mask)

    def _forward_pass(self, x, mask=None):
        pe = self.patch_embed
        B = x.shape[0]
        N = (x.shape[2] // pe.tubelet_size) * (x.shape[3] // pe.patch_size[0]) * (x.shape[4] // pe.patch_size[1])
        if mask is not None:                                                                     # VT:293-294 `x[~mask]`
            vis_idx, _ = mae_gather_indices(mask, x.device)
        else:
            vis_idx = torch.arange(N + 1, dtype=torch.int32, device=x.device).unsqueeze(0).expand(B, N + 1).contiguous()
        L = vis_idx.shape[1] - 1
        pos = self.pos_embed.detach()
        if pos.device != x.device:
            pos = pos.to(x.device)
            if not isinstance(self.pos_embed, nn.Parameter):
                self.pos_embed = pos
        if pos.shape[-2] != N:
            raise ValueError(f"the clip has {N} tokens but the positional table {pos.shape[-2]}")
        D = self.embed_dim
        kreal = pe.proj.weight[0].numel()
        kp = (kreal + 63) // 64 * 64
        wp = torch.zeros((D, kp), dtype=torch.bfloat16, device=x.device)
        wp[:, :kreal] = Fn.mat(pe.proj.weight).reshape(D, kreal)
        tok = ops.gemm(ops.patch_im2col(x, vis_idx, pe.tubelet_size, pe.patch_size[0], kp), wp, bias=Fn.vec(pe.proj.bias))
        x0 = ops.assemble_tokens_nocls(tok, pos.reshape(-1, D).float().contiguous(), vis_idx)     # VT:289-294
        del tok
        outs = Fn.ln_block_stack_infer(x0, [blk.flat_params() for blk in self.blocks], B, L, self.num_heads, self.blocks[0].norm1.eps,
                                       taps=self.return_index, final_norm=(self.norm.weight, self.norm.bias, self.norm.eps),
                                       heads_as_sequence=(self.attn_semantics == "reference"))
        l2 = self.mae_norm_type == 'l2'
        z = [ops.frames_merge_l2(outs[i].contiguous(), B, 1, L, l2=l2) for i in sorted(self.return_index)]   # VT:304-307 (T = 1: identity merge)
        return torch.stack(z)


def load_state_dict(model, state_dict):
    """VT:315-326: keep the `encoder.` sub-tree of a VideoMAE pre-training checkpoint; average the tubelet kernel over time when the
    teacher runs with tubelet_size 1."""
    from collections import OrderedDict
    new_state_dict = OrderedDict()
    for k, v in state_dict.items():
        if k.startswith('encoder.'):
            new_k = k[8:]
            if new_k == "patch_embed.proj.weight" and model.tubelet_size == 1:
                v = v.mean(dim=2, keepdim=True)
            new_state_dict[new_k] = v
    return model.load_state_dict(new_state_dict)


def mae_g14_hybrid(pretrained=True, **kwargs):
    """VT:329-338"""
    model = VisionTransformer(patch_size=14, embed_dim=1408, depth=40, num_heads=16, mlp_ratio=48 / 11, qkv_bias=True,
                              norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
    if pretrained:
        state_dict = torch.load(_MODELS["vit_g14_hybrid"], map_location='cpu')
        load_state_dict(model, state_dict['model'])
    return model
