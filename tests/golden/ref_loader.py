"""Import the REFERENCE's own InternVideo2 modules on CPU (authoring container only).

/root/reference does not exist on the GPU box, so nothing under `-m gpu`, smoke() or bench.py uses
this file; it is used by tests/golden/make_golden.py (fixture generation) and by
bench.py's cpu_baseline leg when the tree is present (kind "reference").

The reference hard-imports `timm` and `flash_attn` at module top
(InternVideo2/single_modality/models/internvideo2_pretrain.py:4-5,13-15); neither is installed, so
we pre-seed sys.modules with the few names it needs (timm 0.5.4 semantics) and load
the three model files by path as a synthetic package.  No reference source is copied.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import torch
from torch import nn

REF_ROOT = os.environ.get("IV_REFERENCE_ROOT", "/root/reference")
SM_MODELS = os.path.join(REF_ROOT, "InternVideo2", "single_modality", "models")
MM_MODELS = os.path.join(REF_ROOT, "InternVideo2", "multi_modality", "models")


def available() -> bool:
    return os.path.isfile(os.path.join(SM_MODELS, "internvideo2_pretrain.py"))


class _DropPath(nn.Module):
    """timm 0.5.4 DropPath: per-sample Bernoulli keep, scaled by 1/keep (identity in eval)."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if not self.drop_prob or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        rnd = keep + torch.rand(shape, dtype=x.dtype, device=x.device)
        return x.div(keep) * rnd.floor_()


def _install_stubs():
    if "timm.models.layers" in sys.modules and getattr(sys.modules["timm"], "_iv_stub", False):
        return

    def to_2tuple(x):
        return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

    def trunc_normal_(t, mean=0., std=1., a=-2., b=2.):
        return nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    def register_model(fn):
        return fn

    timm = types.ModuleType("timm"); timm._iv_stub = True
    tm = types.ModuleType("timm.models")
    tl = types.ModuleType("timm.models.layers")
    tr = types.ModuleType("timm.models.registry")
    tl.DropPath, tl.to_2tuple, tl.trunc_normal_ = _DropPath, to_2tuple, trunc_normal_
    tr.register_model = register_model
    timm.models = tm; tm.layers = tl; tm.registry = tr
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl, "timm.models.registry": tr})

    def _missing(*a, **k):
        raise RuntimeError("flash_attn is not available: use the unfused reference path")

    fa = types.ModuleType("flash_attn")
    fa_mod = types.ModuleType("flash_attn.modules"); fa_mlp = types.ModuleType("flash_attn.modules.mlp")
    fa_ops = types.ModuleType("flash_attn.ops"); fa_rms = types.ModuleType("flash_attn.ops.rms_norm")
    fa_if = types.ModuleType("flash_attn.flash_attn_interface"); fa_bp = types.ModuleType("flash_attn.bert_padding")
    fa_mlp.FusedMLP = _missing; fa_rms.DropoutAddRMSNorm = _missing
    fa_if.flash_attn_varlen_qkvpacked_func = _missing; fa_if.flash_attn_func = _missing
    fa_bp.unpad_input = _missing; fa_bp.pad_input = _missing
    sys.modules.update({"flash_attn": fa, "flash_attn.modules": fa_mod, "flash_attn.modules.mlp": fa_mlp,
                        "flash_attn.ops": fa_ops, "flash_attn.ops.rms_norm": fa_rms,
                        "flash_attn.flash_attn_interface": fa_if, "flash_attn.bert_padding": fa_bp})


def _load(pkg: str, name: str, path: str):
    full = f"{pkg}.{name}"
    if full in sys.modules:
        return sys.modules[full]
    spec = importlib.util.spec_from_file_location(full, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[full] = mod
    spec.loader.exec_module(mod)
    return mod


def load_sm_pretrain():
    """Returns the reference module InternVideo2/single_modality/models/internvideo2_pretrain.py."""
    _install_stubs()
    pkg = "_iv_ref_sm_models"
    if pkg not in sys.modules:
        p = types.ModuleType(pkg); p.__path__ = [SM_MODELS]
        sys.modules[pkg] = p
    _load(pkg, "pos_embed", os.path.join(SM_MODELS, "pos_embed.py"))
    _load(pkg, "flash_attention_class", os.path.join(SM_MODELS, "flash_attention_class.py"))
    return _load(pkg, "internvideo2_pretrain", os.path.join(SM_MODELS, "internvideo2_pretrain.py"))


def load_sm_distill():
    """Returns the reference module InternVideo2/single_modality/models/internvideo2_distill.py."""
    load_sm_pretrain()
    return _load("_iv_ref_sm_models", "internvideo2_distill", os.path.join(SM_MODELS, "internvideo2_distill.py"))


def load_sm_clip_teacher():
    """Returns the reference module InternVideo2/single_modality/models/internvl_clip_vision.py."""
    load_sm_pretrain()
    return _load("_iv_ref_sm_models", "internvl_clip_vision", os.path.join(SM_MODELS, "internvl_clip_vision.py"))


def build_reference_clip_teacher(cfg, **extra):
    ref = load_sm_clip_teacher()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref.InternVL_CLIP(
            in_chans=cfg.in_chans, patch_size=cfg.patch_size, img_size=cfg.img_size, qkv_bias=False, drop_path_rate=0.0,
            embed_dim=cfg.embed_dim, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, init_values=0.1, qk_normalization=True,
            depth=cfg.depth, use_flash_attn=False, use_fused_rmsnorm=False, use_fused_mlp=False,
            attn_pool_num_heads=cfg.attn_pool_num_heads, clip_embed_dim=cfg.clip_embed_dim, layerscale_no_force_fp32=False,
            clip_norm_type="l2", return_attn=True, clip_return_layer=cfg.clip_return_layer,
            clip_return_interval=cfg.clip_student_return_interval, **extra)
    return m


def load_iv1_videomae():
    """InternVideo1/Pretrain/VideoMAE/modeling_pretrain.py (imports `modeling_finetune` by bare name; timm names stubbed)."""
    _install_stubs()
    tl = sys.modules["timm.models.layers"]
    if not hasattr(tl, "drop_path"):
        def drop_path(x, drop_prob: float = 0., training: bool = False):
            if drop_prob == 0. or not training:
                return x
            keep = 1 - drop_prob
            shape = (x.shape[0],) + (1,) * (x.ndim - 1)
            rnd = keep + torch.rand(shape, dtype=x.dtype, device=x.device)
            return x.div(keep) * rnd.floor_()
        tl.drop_path = drop_path
    base = os.path.join(REF_ROOT, "InternVideo1", "Pretrain", "VideoMAE")
    if "modeling_finetune" not in sys.modules:
        spec = importlib.util.spec_from_file_location("modeling_finetune", os.path.join(base, "modeling_finetune.py"))
        mod = importlib.util.module_from_spec(spec); sys.modules["modeling_finetune"] = mod
        spec.loader.exec_module(mod)
    if "_iv_ref_mae_pretrain" not in sys.modules:
        spec = importlib.util.spec_from_file_location("_iv_ref_mae_pretrain", os.path.join(base, "modeling_pretrain.py"))
        mod = importlib.util.module_from_spec(spec); sys.modules["_iv_ref_mae_pretrain"] = mod
        spec.loader.exec_module(mod)
    return sys.modules["_iv_ref_mae_pretrain"]


def build_reference_videomae(cfg):
    """reference PretrainVisionTransformer for an oracle MaeConfig (the reference's PatchEmbed assumes 16 frames: num_patches is
    patched onto the instance for other clip lengths -- only the two sinusoid tables depend on it)."""
    from functools import partial
    ref = load_iv1_videomae()
    import modeling_finetune as mf
    orig = mf.PatchEmbed.__init__

    def patched(self, *a, **k):
        k["num_frames"] = cfg.num_frames
        orig(self, *a, **k)
    mf.PatchEmbed.__init__ = patched
    try:
        m = ref.PretrainVisionTransformer(
            img_size=cfg.img_size, patch_size=cfg.patch_size, encoder_embed_dim=cfg.enc_dim, encoder_depth=cfg.enc_depth,
            encoder_num_heads=cfg.enc_heads, encoder_num_classes=0, decoder_num_classes=cfg.num_classes, decoder_embed_dim=cfg.dec_dim,
            decoder_depth=cfg.dec_depth, decoder_num_heads=cfg.dec_heads, mlp_ratio=cfg.mlp_ratio, qkv_bias=cfg.qkv_bias,
            norm_layer=partial(nn.LayerNorm, eps=cfg.ln_eps), init_values=cfg.init_values, tubelet_size=cfg.tubelet_size)
    finally:
        mf.PatchEmbed.__init__ = orig
    return m


def load_sm_videomae_teacher():
    """single_modality/models/videomae.py.  It imports `flash_attn.flash_attn_func` (not installed): the stand-in follows flash_attn's
    documented contract -- q, k, v (batch, seqlen, nheads, headdim) -> (batch, seqlen, nheads, headdim) -- restated in
    oracle.internvideo2_oracle.flash_attn_func_contract."""
    load_iv1_videomae()                                  # installs timm.models.layers.drop_path
    from oracle import internvideo2_oracle as O

    def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, **kw):
        assert dropout_p == 0.0 and not causal
        return O.flash_attn_func_contract(q, k, v, softmax_scale if softmax_scale is not None else q.shape[-1] ** -0.5)
    sys.modules["flash_attn"].flash_attn_func = flash_attn_func
    return _load("_iv_ref_sm_models", "videomae", os.path.join(SM_MODELS, "videomae.py"))


def build_reference_finetune(cfg, num_classes, **extra):
    """reference fine-tuning classifier (single_modality/models/internvideo2.py InternVideo2, unfused path)"""
    load_sm_pretrain()
    ref = _load("_iv_ref_sm_models", "internvideo2", os.path.join(SM_MODELS, "internvideo2.py"))
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref.InternVideo2(
            in_chans=cfg.in_chans, patch_size=cfg.patch_size, img_size=cfg.img_size, qkv_bias=False, drop_path_rate=0.0,
            embed_dim=cfg.embed_dim, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio, init_values=1e-5, qk_normalization=True,
            depth=cfg.depth, use_flash_attn=False, use_fused_rmsnorm=False, use_fused_mlp=False,
            attn_pool_num_heads=cfg.attn_pool_num_heads, clip_embed_dim=cfg.clip_embed_dim, num_frames=cfg.num_frames,
            tubelet_size=cfg.tubelet_size, sep_pos_embed=extra.pop("sep_pos_embed", False), num_classes=num_classes, **extra)
    return m


def load_mm_vision():
    """Returns the reference module InternVideo2/multi_modality/models/backbones/internvideo2/internvideo2.py."""
    _install_stubs()
    pkg = "_iv_ref_mm_iv2"
    base = os.path.join(MM_MODELS, "backbones", "internvideo2")
    if pkg not in sys.modules:
        p = types.ModuleType(pkg); p.__path__ = [base]
        sys.modules[pkg] = p
    _load(pkg, "pos_embed", os.path.join(base, "pos_embed.py"))
    _load(pkg, "flash_attention_class", os.path.join(base, "flash_attention_class.py"))
    return _load(pkg, "internvideo2", os.path.join(base, "internvideo2.py"))


def load_mm_mask():
    """multi_modality/models/mask.py (batched generators)."""
    return _load_pkgless("mm_mask", os.path.join(MM_MODELS, "mask.py"))


def _load_pkgless(name, path):
    spec = importlib.util.spec_from_file_location("_iv_ref_" + name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build_reference_distill(cfg, **extra):
    """Reference DistInternVideo2 (unfused path) for an oracle StudentConfig with has_mae=False."""
    ref = load_sm_distill()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref.DistInternVideo2(
            in_chans=cfg.in_chans, patch_size=cfg.patch_size, img_size=cfg.img_size, qkv_bias=False,
            drop_path_rate=0.0, embed_dim=cfg.embed_dim, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
            init_values=1e-5, qk_normalization=True, depth=cfg.depth,
            use_flash_attn=False, use_fused_rmsnorm=False, use_fused_mlp=False,
            attn_pool_num_heads=cfg.attn_pool_num_heads, clip_embed_dim=cfg.clip_embed_dim,
            num_frames=cfg.num_frames, tubelet_size=cfg.tubelet_size, sep_pos_embed=extra.pop("sep_pos_embed", False),
            clip_teacher_embed_dim=cfg.clip_teacher_embed_dim, clip_teacher_final_dim=cfg.clip_teacher_final_dim,
            clip_norm_type="l2", clip_return_layer=cfg.clip_return_layer,
            clip_student_return_interval=cfg.clip_student_return_interval,
            clip_student_return_index=list(cfg.clip_return_index_override) if cfg.clip_return_index_override else None,
            clip_student_decoder={"linear": "Linear_Decoder", "mlp": "MLP_Decoder"}[cfg.clip_decoder_kind],
            **extra)
    return m


def build_reference_mm_vision(cfg, **extra):
    """Reference stage-2 vision encoder (multi_modality PretrainInternVideo2, unfused path)."""
    ref = load_mm_vision()
    m = ref.PretrainInternVideo2(
        in_chans=cfg.in_chans, patch_size=cfg.patch_size, img_size=cfg.img_size, qkv_bias=False,
        drop_path_rate=0.0, embed_dim=cfg.embed_dim, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
        init_values=1e-5, qk_normalization=True, depth=cfg.depth,
        use_flash_attn=False, use_fused_rmsnorm=False, use_fused_mlp=False,
        attn_pool_num_heads=cfg.attn_pool_num_heads, clip_embed_dim=cfg.clip_embed_dim,
        num_frames=cfg.num_frames, tubelet_size=cfg.tubelet_size, sep_pos_embed=False,
        sep_image_video_pos_embed=cfg.sep_image_video_pos_embed,
        clip_teacher_embed_dim=cfg.clip_teacher_embed_dim, clip_teacher_final_dim=cfg.clip_teacher_final_dim,
        clip_norm_type="l2", clip_return_layer=cfg.clip_return_layer,
        clip_student_return_interval=cfg.clip_student_return_interval, **extra)
    return m


def build_reference_student(cfg, **extra):
    """Instantiate the reference PretrainInternVideo2 (unfused path) for an oracle StudentConfig."""
    ref = load_sm_pretrain()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref.PretrainInternVideo2(
            in_chans=cfg.in_chans, patch_size=cfg.patch_size, img_size=cfg.img_size, qkv_bias=False,
            drop_path_rate=0.0, embed_dim=cfg.embed_dim, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
            init_values=1e-5, qk_normalization=True, depth=cfg.depth,
            use_flash_attn=False, use_fused_rmsnorm=False, use_fused_mlp=False,
            attn_pool_num_heads=cfg.attn_pool_num_heads, clip_embed_dim=cfg.clip_embed_dim,
            num_frames=cfg.num_frames, tubelet_size=cfg.tubelet_size, sep_pos_embed=False,
            clip_teacher_embed_dim=cfg.clip_teacher_embed_dim, clip_teacher_final_dim=cfg.clip_teacher_final_dim,
            clip_norm_type="l2", clip_return_layer=cfg.clip_return_layer,
            clip_student_return_interval=cfg.clip_student_return_interval,
            mae_teacher_embed_dim=cfg.mae_teacher_embed_dim, mae_norm_type="l2",
            mae_return_layer=cfg.mae_return_layer, mae_student_return_interval=cfg.mae_student_return_interval,
            **extra)
    return m


def load_mm_criterions():
    """multi_modality/models/criterions.py as a module.  It imports three package-relative helpers (`.utils.allgather_wgrad`,
    `..utils.distributed.get_rank / get_world_size`, `..utils.easydict.EasyDict`); a synthetic parent package supplies single-process
    stand-ins for them, the file itself is imported unmodified."""
    pkg = "_iv_ref_mm"
    if pkg + ".models.criterions" in sys.modules:
        return sys.modules[pkg + ".models.criterions"]
    root = types.ModuleType(pkg); root.__path__ = []
    models = types.ModuleType(pkg + ".models"); models.__path__ = []
    mutils = types.ModuleType(pkg + ".models.utils")
    utils = types.ModuleType(pkg + ".utils"); utils.__path__ = []
    udist = types.ModuleType(pkg + ".utils.distributed")
    ueasy = types.ModuleType(pkg + ".utils.easydict")
    mutils.allgather_wgrad = lambda t, args: t                      # world size 1
    udist.get_rank, udist.get_world_size = (lambda: 0), (lambda: 1)

    class EasyDict(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    ueasy.EasyDict = EasyDict
    sys.modules.update({pkg: root, pkg + ".models": models, pkg + ".models.utils": mutils, pkg + ".utils": utils,
                        pkg + ".utils.distributed": udist, pkg + ".utils.easydict": ueasy})
    return _load(pkg + ".models", "criterions", os.path.join(MM_MODELS, "criterions.py"))


def load_mm_criterions_functions():
    """(get_sim, VTC_VTM_Loss) of the reference's criterions.py"""
    m = load_mm_criterions()
    return m.get_sim, m.VTC_VTM_Loss


def load_mm_xbert():
    """multi_modality/models/backbones/bert/xbert.py under the installed transformers (5.x).  The file was written against
    transformers 4.2x: three helpers it imports from `transformers.modeling_utils` now live in `transformers.pytorch_utils`, the
    docstring decorators left `transformers.file_utils`, and `PreTrainedModel.init_weights / get_head_mask` changed; the shim restores
    those names (module attributes only -- the reference file is imported unmodified) and re-ties the MLM decoder to the word
    embeddings as transformers 4 did for `tie_word_embeddings` (xbert.py:1599-1614)."""
    name = "_iv_ref_xbert"
    if name in sys.modules:
        return sys.modules[name]
    import transformers.file_utils as fu
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    for n in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, n):
            setattr(mu, n, getattr(pu, n))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = getattr(pu, "find_pruneable_heads_and_indices", lambda *a, **k: (set(), None))
    for n in ("add_start_docstrings", "add_start_docstrings_to_model_forward", "replace_return_docstrings", "add_code_sample_docstrings"):
        if not hasattr(fu, n):
            setattr(fu, n, lambda *a, **k: (lambda f: f))
    spec = importlib.util.spec_from_file_location(name, os.path.join(MM_MODELS, "backbones", "bert", "xbert.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    m.BertPreTrainedModel.init_weights = lambda self: self.apply(self._init_weights)
    m.BertPreTrainedModel.get_head_mask = lambda self, head_mask, n, *a, **k: [None] * n
    return m


def build_reference_bert(cfg):
    """reference BertForMaskedLM (what build_bert(pretrain=True) instantiates, builder.py:31-45) for an oracle BertTowerConfig,
    dropout 0, decoder tied to the word embeddings."""
    m = load_mm_xbert()
    c = m.BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                     num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size, hidden_act="gelu",
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, max_position_embeddings=cfg.max_position_embeddings,
                     type_vocab_size=cfg.type_vocab_size, layer_norm_eps=cfg.layer_norm_eps, pad_token_id=cfg.pad_token_id)
    c.fusion_layer, c.encoder_width, c.cross_module = cfg.fusion_layer, cfg.encoder_width, "ca"      # builder.py:18-24, config_bert_large.json
    mdl = m.BertForMaskedLM(c)
    mdl.cls.predictions.decoder.weight = mdl.bert.embeddings.word_embeddings.weight
    return mdl
