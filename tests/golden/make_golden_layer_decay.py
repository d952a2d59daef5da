"""Golden fixture for the layer-wise lr decay.  Authoring container only.

Runs the REFERENCE's own `single_modality/optim_factory.py` (`LayerDecayValueAssigner`, `get_parameter_groups`; timm's optimizer classes, which
the file imports at its top and this recipe never touches, are stubbed) on the parameter names of the fine-tuning classifier at the tiny88
geometry, builds `torch.optim.AdamW` over the groups it returns exactly as run_finetuning.py:548-579 + engines/engine_for_finetuning.py:56
do (`group["lr"] = lr * group["lr_scale"]`, betas (0.9, 0.999), eps 1e-8, weight decay 0.05 -- the fine-tuning script's values), and
applies three steps of seeded gradients (matrix gradients rounded to bf16 first: the engine's matrix-gradient buffer is bf16).

    python tests/golden/make_golden_layer_decay.py      ->  tests/golden/layer_decay.npz
       name:<i>, layer:<i>, scale:<i>, wd:<i>           per parameter, in named_parameters() order
       final:<name>                                     parameter values after the three steps: whole tensor up to 2048 elements, else its
                                                        first 512 + last 512 elements (segment boundaries lie between parameters) and
       norm:<name>                                      its l2 norm (float64)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("IV_REFERENCE_ROOT", "/root/reference")
LAYER_DECAY, LR, WD, STEPS, SEED = 0.75, 1e-2, 0.05, 3, 21


def load_reference_optim_factory():
    for name in ("timm", "timm.optim", "timm.optim.adafactor", "timm.optim.adahessian", "timm.optim.adamp", "timm.optim.lookahead", "timm.optim.nadam",
                 "timm.optim.nvnovograd", "timm.optim.radam", "timm.optim.rmsprop_tf", "timm.optim.sgdp"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    for mod, cls in (("adafactor", "Adafactor"), ("adahessian", "Adahessian"), ("adamp", "AdamP"), ("lookahead", "Lookahead"), ("nadam", "Nadam"),
                     ("nvnovograd", "NvNovoGrad"), ("radam", "RAdam"), ("rmsprop_tf", "RMSpropTF"), ("sgdp", "SGDP")):
        setattr(sys.modules["timm.optim." + mod], cls, object)
    spec = importlib.util.spec_from_file_location("_iv_ref_optim_factory", os.path.join(REF, "InternVideo2", "single_modality", "optim_factory.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def synthetic_grad(name, p, step):
    g = torch.Generator().manual_seed(SEED * 1000 + step * 131 + (sum(map(ord, name)) % 997))
    t = torch.randn(p.shape, generator=g) * 0.05
    decay = not (p.dim() == 1 or name.endswith(".bias"))
    return t.to(torch.bfloat16).float() if decay else t            # engine layout: matrices carry bf16 gradients (no_weight_decay names: see below)


def build_model():
    from internvideo_amd import internvideo2 as FT
    from oracle import internvideo2_oracle as O
    cfg = O.named_config("tiny88")
    params = O.synthetic_finetune_params(cfg, 10, seed=12)
    m = FT.InternVideo2(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
                        num_frames=cfg.num_frames, drop_path_rate=0.0, attn_pool_num_heads=cfg.attn_pool_num_heads,
                        clip_embed_dim=cfg.clip_embed_dim, num_classes=10)
    m.load_state_dict(params, strict=True)
    return m


def main():
    import contextlib, io
    ref = load_reference_optim_factory()
    m = build_model()
    depth = m.get_num_layers()
    assigner = ref.LayerDecayValueAssigner(list(LAYER_DECAY ** (depth + 1 - i) for i in range(depth + 2)))       # run_finetuning.py:549
    skip = m.no_weight_decay()
    with contextlib.redirect_stdout(io.StringIO()):
        groups = ref.get_parameter_groups(m, WD, skip, assigner.get_layer_id, assigner.get_scale)
    by_id = {id(p): n for n, p in m.named_parameters()}
    out = {}
    info = {}
    for g in groups:
        for p in g["params"]:
            info[by_id[id(p)]] = (assigner.get_layer_id(by_id[id(p)]), g["lr_scale"], g["weight_decay"])
    for i, (n, p) in enumerate(m.named_parameters()):
        out[f"name:{i}"] = np.array(n)
        out[f"layer:{i}"] = np.array(info[n][0]); out[f"scale:{i}"] = np.array(info[n][1], dtype=np.float64); out[f"wd:{i}"] = np.array(info[n][2], dtype=np.float64)
    opt = torch.optim.AdamW(groups, lr=LR, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0)          # create_optimizer zeroes the global wd (:112)
    for step in range(1, STEPS + 1):
        for g in opt.param_groups:
            g["lr"] = LR * g["lr_scale"]                                                              # engine_for_finetuning.py:56
        for n, p in m.named_parameters():
            # names in no_weight_decay() are 'no_decay' in the reference but >= 2-D here: the engine keeps them in its fp32 vector region
            gr = synthetic_grad(n, p, step)
            if n in skip:
                gr = torch.randn(p.shape, generator=torch.Generator().manual_seed(SEED * 1000 + step * 131 + (sum(map(ord, n)) % 997))) * 0.05
            p.grad = gr
        opt.step()
    for n, p in m.named_parameters():
        flat = p.detach().reshape(-1)
        out["final:" + n] = (flat if flat.numel() <= 2048 else torch.cat([flat[:512], flat[-512:]])).numpy().copy()
        out["norm:" + n] = np.array(float(flat.double().norm()))
    out["meta"] = np.array([LAYER_DECAY, LR, WD, STEPS, SEED], dtype=np.float64)
    path = os.path.join(HERE, "layer_decay.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;", len(groups), "groups, depth", depth)


if __name__ == "__main__":
    main()
