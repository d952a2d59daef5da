"""MI355X-native mirror of the VideoMAE pixel-reconstruction model, InternVideo1/Pretrain/VideoMAE/modeling_pretrain.py ("MP:")
with the blocks of modeling_finetune.py ("MF:") and the target / loss of engine_for_pretraining.py ("ME:") -- SURVEY.md 8(a) row a23.

    PretrainVisionTransformer.forward(x (B,3,T,H,W), mask (B,N) bool, True = masked) -> (B, N_mask, 3 * tubelet * p^2) predictions

  encoder  : tubelet patch embed of the VISIBLE cubes only + fixed sinusoid table -> LayerNorm pre-norm blocks (q/v bias, optional
             gamma_1/2, DropPath) -> LayerNorm                                                            (MP:34-170)
  bridge   : encoder_to_decoder Linear (no bias); decoder input = [x_vis + pos[~mask] ; mask_token + pos[mask]]  (MP:375-389)
  decoder  : the same blocks on all N tokens, LayerNorm + Linear head on the last N_mask rows              (MP:173-268)
  target   : per-cube normalised pixels of the masked tokens, MSE                                           (ME:53-106)

Same class names, constructor kwargs, state_dict keys / shapes and registry names as the reference; all arithmetic in the gfx950
kernels (LN-block autograd seam functional.LNBlockStackFn); no CPU path.  `use_learnable_pos_emb=True` builds a table with a cls
row that the reference's own forward cannot add to the cls-free tokens (MP:77-79 vs MP:127-129), so it is rejected.
"""
from __future__ import annotations

from functools import partial
from typing import Optional

import numpy as np
import torch
from torch import nn

from . import functional as Fn
from . import ops
from .internvideo2_pretrain import build_gather_indices, register_model
from .lib import InternVideoHipError

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)


def trunc_normal_(tensor, mean=0., std=1.):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=-std, b=std)                      # MP:21-22


def get_sinusoid_encoding_table(n_position: int, d_hid: int) -> torch.Tensor:
    """MF:224-241: table[pos, j] = pos / 10000^(2 (j // 2) / d); sin on even j, cos on odd j (float64 numpy -> fp32), (1, N, d)."""
    j = np.arange(d_hid)
    ang = np.arange(n_position, dtype=np.float64)[:, None] / np.power(10000, 2 * (j // 2) / d_hid)[None, :]
    ang[:, 0::2] = np.sin(ang[:, 0::2])
    ang[:, 1::2] = np.cos(ang[:, 1::2])
    return torch.tensor(ang, dtype=torch.float, requires_grad=False).unsqueeze(0)


def mae_gather_indices(mask: torch.Tensor, device):
    """mask (B, N) bool, True = masked -> (vis_idx int32 [B, 1+Nvis] with a leading pseudo-cls 0, msk_idx int32 [B, Nmask]), token
    ids + 1, ascending: `x[~mask]` / `x[mask]` order (MP:133, MP:384-385), bit-exact.  Raises on ragged rows like the reshape."""
    m = mask.reshape(mask.shape[0], -1).to(torch.bool)
    head = torch.zeros((m.shape[0], 1), dtype=torch.bool, device=m.device)
    vis, _ = build_gather_indices(torch.cat([head, m], 1), device)
    msk, _ = build_gather_indices(torch.cat([~head, ~m], 1), device)
    return vis, msk


class Mlp(nn.Module):
    """MF:49-72 parameter container."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)


class Attention(nn.Module):
    """MF:75-129 parameter container: qkv without bias + separate q_bias / v_bias (k has none), proj."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., attn_head_dim=None):
        super().__init__()
        if attn_drop or proj_drop or qk_scale is not None or attn_head_dim is not None:
            raise InternVideoHipError("VideoMAE Attention (MI355X): dropout / qk_scale / attn_head_dim overrides are not implemented")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        if qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(dim))
            self.v_bias = nn.Parameter(torch.zeros(dim))
        else:
            self.q_bias = None
            self.v_bias = None
        self.proj = nn.Linear(dim, dim)


class Block(nn.Module):
    """MF:132-181 parameter container; arithmetic in functional.LNBlockStackFn."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0., drop_path=0.,
                 init_values=None, act_layer=nn.GELU, norm_layer=nn.LayerNorm, attn_head_dim=None):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop,
                              attn_head_dim=attn_head_dim)
        self.drop_path = float(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), drop=drop)
        if init_values is not None and init_values > 0:
            self.gamma_1 = nn.Parameter(init_values * torch.ones((dim)), requires_grad=True)
            self.gamma_2 = nn.Parameter(init_values * torch.ones((dim)), requires_grad=True)
        else:
            self.gamma_1, self.gamma_2 = None, None

    def flat_params(self):
        a = self.attn
        return [self.norm1.weight, self.norm1.bias, a.qkv.weight, a.q_bias, a.v_bias, a.proj.weight, a.proj.bias, self.gamma_1,
                self.norm2.weight, self.norm2.bias, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias,
                self.gamma_2]


class PatchEmbed(nn.Module):
    """MF:184-219: Conv3d k = s = (tubelet, p, p) parameters."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, num_frames=16, tubelet_size=2):
        super().__init__()
        img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.tubelet_size = int(tubelet_size)
        self.img_size, self.patch_size = img_size, patch_size
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0]) * (num_frames // self.tubelet_size)
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=(self.tubelet_size, patch_size[0], patch_size[1]),
                              stride=(self.tubelet_size, patch_size[0], patch_size[1]))


def _xavier_init(m):
    """MP:100-107"""
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)


def _drop_path_scales(rates, training, B, device):
    """timm DropPath per (block, branch, sample): floor(keep + U) / keep (MF:31-46); None when inactive"""
    if not training or max(rates, default=0.0) == 0.0:
        return None
    keep = 1.0 - torch.tensor(rates, dtype=torch.float32, device=device).view(-1, 1, 1)
    u = torch.rand((len(rates), 2, B), dtype=torch.float32, device=device)
    return (torch.floor(keep + u) / keep).contiguous()


def _run_blocks(blocks, x0, B, L, num_heads, eps, training):
    rates = [blk.drop_path for blk in blocks]
    meta = dict(B=B, L=L, H=num_heads, eps=eps)
    params = [p for blk in blocks for p in blk.flat_params()]
    return Fn.LNBlockStackFn.apply(x0, _drop_path_scales(rates, training, B, x0.device), meta, *params)


class PretrainVisionTransformerEncoder(nn.Module):
    """MP:34-170."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=0, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.,
                 qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0., norm_layer=nn.LayerNorm,
                 init_values=None, tubelet_size=2, use_learnable_pos_emb=False, with_cp=False, num_frames=16):
        super().__init__()
        if use_learnable_pos_emb:
            raise NotImplementedError("use_learnable_pos_emb: the reference's (1, N+1, D) table cannot be added to its cls-free tokens")
        if num_classes or drop_rate:
            raise InternVideoHipError("VideoMAE encoder (MI355X): num_classes > 0 / dropout are not part of the pre-training recipe")
        self.num_classes, self.num_features, self.embed_dim, self.num_heads = num_classes, embed_dim, embed_dim, num_heads
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim,
                                      tubelet_size=tubelet_size, num_frames=num_frames)
        self.with_cp = with_cp
        self.pos_embed = get_sinusoid_encoding_table(self.patch_embed.num_patches, embed_dim)      # plain tensor, as MP:80 (not a buffer)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth, device="cpu")]
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                  attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer, init_values=init_values) for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Identity()
        self.apply(_xavier_init)

    def get_num_layers(self):
        return len(self.blocks)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def _pos(self, device):
        if self.pos_embed.device != torch.device(device):
            self.pos_embed = self.pos_embed.to(device)                                            # MP:129 `.to(x.device)`, cached
        return self.pos_embed.reshape(-1, self.embed_dim)

    def forward_features(self, x, mask, vis_idx=None):
        """-> bf16 (B, N_vis, C): encoder.norm(blocks(patch_embed(x) + pos)[~mask])    (MP:125-139)"""
        if not x.is_cuda:
            raise InternVideoHipError("VideoMAE forward needs HBM-resident inputs: there is no CPU path")
        if vis_idx is None:
            vis_idx, _ = mae_gather_indices(mask, x.device)
        B, Nvis = vis_idx.shape[0], vis_idx.shape[1] - 1
        pe = self.patch_embed
        x0 = Fn.PatchEmbedVisibleFn.apply(x, vis_idx, pe.proj.weight, pe.proj.bias, self._pos(x.device), pe.tubelet_size, pe.patch_size[0])
        h = _run_blocks(self.blocks, x0, B, Nvis, self.num_heads, self.blocks[0].norm1.eps, self.training)
        return Fn.LayerNormFn.apply(h, self.norm.weight, self.norm.bias, self.norm.eps).view(B, Nvis, -1)

    def forward(self, x, mask):
        return self.head(self.forward_features(x, mask))


class PretrainVisionTransformerDecoder(nn.Module):
    """MP:173-268."""

    def __init__(self, patch_size=16, num_classes=768, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4., qkv_bias=False,
                 qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0., norm_layer=nn.LayerNorm, init_values=None,
                 num_patches=196, tubelet_size=2, with_cp=False, with_fp16=True):
        super().__init__()
        self.num_classes = num_classes
        assert num_classes == 3 * tubelet_size * patch_size ** 2
        self.num_features, self.embed_dim, self.num_heads = embed_dim, embed_dim, num_heads
        self.patch_size, self.with_cp, self.with_fp16 = patch_size, with_cp, with_fp16
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth, device="cpu")]
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop_rate,
                  attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer, init_values=init_values) for i in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        self.apply(_xavier_init)

    def get_num_layers(self):
        return len(self.blocks)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token'}

    def forward_stream(self, x_rows, B, N, return_token_num):
        """x_rows: fp32 stream rows [B*N, Cd] -> bf16 (B, return_token_num | N, num_classes)    (MP:254-268)"""
        h = _run_blocks(self.blocks, x_rows, B, N, self.num_heads, self.blocks[0].norm1.eps, self.training)
        cnt = return_token_num if return_token_num > 0 else N
        tail = Fn.RowsWindowFn.apply(h, B, N, N - cnt, cnt)
        y = Fn.LayerNormFn.apply(tail, self.norm.weight, self.norm.bias, self.norm.eps)
        if isinstance(self.head, nn.Linear):
            y = Fn.LinearFn.apply(y, self.head.weight, self.head.bias)
        return y.view(B, cnt, -1)

    def forward(self, x, return_token_num):
        """x (B, N, Cd) any float dtype, as the reference's signature"""
        B, N, Cd = x.shape
        return self.forward_stream(x.reshape(B * N, Cd).float().contiguous(), B, N, return_token_num)


class PretrainVisionTransformer(nn.Module):
    """MP:271-392."""

    def __init__(self, img_size=224, patch_size=16, encoder_in_chans=3, encoder_num_classes=0, encoder_embed_dim=768,
                 encoder_depth=12, encoder_num_heads=12, decoder_num_classes=1536, decoder_embed_dim=512, decoder_depth=8,
                 decoder_num_heads=8, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer=nn.LayerNorm, init_values=0., use_learnable_pos_emb=False, tubelet_size=2,
                 num_classes=0, in_chans=0, with_cp=False, num_frames=16):
        super().__init__()
        self.encoder = PretrainVisionTransformerEncoder(
            img_size=img_size, patch_size=patch_size, in_chans=encoder_in_chans, num_classes=encoder_num_classes,
            embed_dim=encoder_embed_dim, depth=encoder_depth, num_heads=encoder_num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
            qk_scale=qk_scale, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=drop_path_rate, norm_layer=norm_layer,
            init_values=init_values, tubelet_size=tubelet_size, use_learnable_pos_emb=use_learnable_pos_emb, with_cp=with_cp,
            num_frames=num_frames)
        self.decoder = PretrainVisionTransformerDecoder(
            patch_size=patch_size, num_patches=self.encoder.patch_embed.num_patches, num_classes=decoder_num_classes,
            embed_dim=decoder_embed_dim, depth=decoder_depth, num_heads=decoder_num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
            qk_scale=qk_scale, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate, drop_path_rate=drop_path_rate, norm_layer=norm_layer,
            init_values=init_values, tubelet_size=tubelet_size, with_cp=with_cp, with_fp16=True)
        self.encoder_to_decoder = nn.Linear(encoder_embed_dim, decoder_embed_dim, bias=False)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.pos_embed = get_sinusoid_encoding_table(self.encoder.patch_embed.num_patches, decoder_embed_dim)
        trunc_normal_(self.mask_token, std=.02)
        self.patch_size, self.tubelet_size = patch_size, tubelet_size

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'mask_token'}

    def _dec_pos(self, device):
        if self.pos_embed.device != torch.device(device):
            self.pos_embed = self.pos_embed.to(device)
        return self.pos_embed.reshape(-1, self.pos_embed.shape[-1])

    def forward(self, x, mask, indices=None):
        """MP:375-392.  x (B,3,T,H,W); mask (B,N) bool (True = masked, equal count per clip) -> bf16 (B, N_mask, 3*tubelet*p^2)"""
        vis_idx, msk_idx = indices if indices is not None else mae_gather_indices(mask, x.device)
        B, Nvis, Nmask = vis_idx.shape[0], vis_idx.shape[1] - 1, msk_idx.shape[1]
        x_vis = self.encoder.forward_features(x, mask, vis_idx=vis_idx)                          # (B, Nvis, Ce) bf16
        x_vis = Fn.LinearFn.apply(x_vis, self.encoder_to_decoder.weight, None)                    # MP:377
        x_full = Fn.MaeDecoderInputFn.apply(x_vis.reshape(B * Nvis, -1), self.mask_token, self._dec_pos(x.device), vis_idx, msk_idx)
        return self.decoder.forward_stream(x_full, B, Nvis + Nmask, Nmask)                        # MP:390

    def pixel_target(self, images, mask=None, msk_idx=None, normlize_target: bool = True):
        """the regression labels of engine_for_pretraining.py:66-98 for ImageNet-normalised `images` (fp32 (B, N_mask, 3*tubelet*p^2))"""
        if msk_idx is None:
            _, msk_idx = mae_gather_indices(mask, images.device)
        return ops.pixel_target(images, msk_idx, self.tubelet_size, self.patch_size, normalize=normlize_target,
                                mean=IMAGENET_DEFAULT_MEAN, std=IMAGENET_DEFAULT_STD)

    def forward_loss(self, images, mask, normlize_target: bool = True):
        """one pre-training step's forward: labels (ME:66-98), predictions (MP:375-392), nn.MSELoss (ME:101-106) -> fp32 scalar"""
        idx = mae_gather_indices(mask, images.device)
        with torch.no_grad():
            labels = self.pixel_target(images, msk_idx=idx[1], normlize_target=normlize_target)
        return Fn.MseLossFn.apply(self.forward(images, mask, indices=idx), labels)


def _mae(pretrained=False, **kw):
    init_ckpt = kw.pop("init_ckpt", None)
    model = PretrainVisionTransformer(**kw)
    if pretrained:
        model.load_state_dict(torch.load(init_ckpt, map_location="cpu")["model"])
    return model


_LN6 = partial(nn.LayerNorm, eps=1e-6)


@register_model
def pretrain_mae_small_patch16_224(pretrained=False, **kwargs):
    """MP:395-413"""
    return _mae(pretrained, img_size=224, patch_size=16, encoder_embed_dim=384, encoder_depth=12, encoder_num_heads=6,
                encoder_num_classes=0, decoder_num_classes=1536, decoder_embed_dim=192, decoder_num_heads=3, mlp_ratio=4,
                qkv_bias=True, norm_layer=_LN6, **kwargs)


@register_model
def pretrain_mae_base_patch16_224(pretrained=False, **kwargs):
    """MP:416-434"""
    return _mae(pretrained, img_size=224, patch_size=16, encoder_embed_dim=768, encoder_depth=12, encoder_num_heads=12,
                encoder_num_classes=0, decoder_num_classes=1536, decoder_embed_dim=384, decoder_num_heads=6, mlp_ratio=4,
                qkv_bias=True, norm_layer=_LN6, **kwargs)


@register_model
def pretrain_mae_large_patch16_224(pretrained=False, **kwargs):
    """MP:437-455"""
    return _mae(pretrained, img_size=224, patch_size=16, encoder_embed_dim=1024, encoder_depth=24, encoder_num_heads=16,
                encoder_num_classes=0, decoder_num_classes=1536, decoder_embed_dim=512, decoder_num_heads=8, mlp_ratio=4,
                qkv_bias=True, norm_layer=_LN6, **kwargs)


@register_model
def pretrain_mae_huge_patch16_224(pretrained=False, **kwargs):
    """MP:458-476"""
    return _mae(pretrained, img_size=224, patch_size=16, encoder_embed_dim=1280, encoder_depth=32, encoder_num_heads=16,
                encoder_num_classes=0, decoder_num_classes=1536, decoder_embed_dim=512, decoder_num_heads=8, mlp_ratio=4,
                qkv_bias=True, norm_layer=_LN6, **kwargs)


@register_model
def pretrain_mae_giant_patch16_224(pretrained=False, **kwargs):
    """MP:479-497"""
    return _mae(pretrained, img_size=224, patch_size=16, encoder_embed_dim=1408, encoder_depth=40, encoder_num_heads=16,
                encoder_num_classes=0, decoder_num_classes=1536, decoder_embed_dim=512, decoder_num_heads=8, mlp_ratio=48 / 11,
                qkv_bias=True, norm_layer=_LN6, **kwargs)


@register_model
def pretrain_mae_giant_patch14_224(pretrained=False, **kwargs):
    """MP:500-518"""
    return _mae(pretrained, img_size=224, patch_size=14, encoder_embed_dim=1408, encoder_depth=40, encoder_num_heads=16,
                encoder_num_classes=0, decoder_num_classes=1176, decoder_embed_dim=512, decoder_num_heads=8, mlp_ratio=48 / 11,
                qkv_bias=True, norm_layer=_LN6, **kwargs)


@register_model
def pretrain_mae_gigantic_patch14_224(pretrained=False, **kwargs):
    """MP:521-539"""
    return _mae(pretrained, img_size=224, patch_size=14, encoder_embed_dim=1664, encoder_depth=48, encoder_num_heads=16,
                encoder_num_classes=0, decoder_num_classes=1176, decoder_embed_dim=512, decoder_num_heads=8, mlp_ratio=64 / 13,
                qkv_bias=True, norm_layer=_LN6, **kwargs)
