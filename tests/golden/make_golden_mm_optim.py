"""Golden fixture for the stage-2 optimizer groups and lr schedule (stage-2 flavour).  Authoring container only.

Loads the REFERENCE's multi_modality/utils/optimizer.py and utils/scheduler.py by file path (`utils.distributed.is_main_process`, which the
former imports for its logging, is stubbed) and records
  * the LambdaLR multiplier of `get_cosine_schedule_with_warmup` at every step of three (warm-up, total, min_lr_multi, cycles) settings;
  * `add_weight_decay` -> `add_different_lr` -> `create_optimizer_params_group` on a small module with every kind of parameter name the
    rules look at (1-D, `.bias`, a no-decay list entry, a frozen weight, names matching / not matching the different-lr patterns).

    python tests/golden/make_golden_mm_optim.py      ->  tests/golden/mm_optim.json
"""
import importlib.util
import json
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("IV_REFERENCE_ROOT", "/root/reference")

SCHEDULES = [dict(num_warmup_steps=10, num_training_steps=100, min_lr_multi=0.01, num_cycles=0.5),
             dict(num_warmup_steps=0, num_training_steps=37, min_lr_multi=0.0, num_cycles=0.5),
             dict(num_warmup_steps=5, num_training_steps=5, min_lr_multi=0.2, num_cycles=1.0)]
DIFF = dict(names=[r"text_encoder\.", r"proj$", "temp"], lr=1e-3, default=5e-5)
NO_DECAY = ("vision_encoder.pos_embed", "text_encoder.special.weight")


class Toy(torch.nn.Module):
    """parameter names shaped like the stage-2 model's"""

    def __init__(self):
        super().__init__()
        self.vision_encoder = torch.nn.Module()
        self.vision_encoder.pos_embed = torch.nn.Parameter(torch.zeros(1, 5, 8))
        self.vision_encoder.fc = torch.nn.Linear(8, 8)
        self.vision_encoder.norm = torch.nn.LayerNorm(8)
        self.vision_encoder.frozen = torch.nn.Linear(8, 8)
        self.vision_encoder.frozen.weight.requires_grad_(False)
        self.text_encoder = torch.nn.Module()
        self.text_encoder.emb = torch.nn.Embedding(11, 8)
        self.text_encoder.special = torch.nn.Linear(8, 4, bias=False)
        self.vision_proj = torch.nn.Linear(8, 4)
        self.text_proj = torch.nn.Linear(8, 4)
        self.temp = torch.nn.Parameter(torch.ones([]) * 0.07)
        self.itm_head = torch.nn.Linear(8, 2)


def load_reference():
    for name in ("utils", "utils.distributed"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["utils.distributed"].is_main_process = lambda: False
    mods = []
    for fn in ("optimizer.py", "scheduler.py"):
        spec = importlib.util.spec_from_file_location("_iv_ref_mm_" + fn[:-3], os.path.join(REF, "InternVideo2", "multi_modality", "utils", fn))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods.append(m)
    return mods


def main():
    opt, sch = load_reference()
    out = {"schedules": [], "groups": None}
    for kw in SCHEDULES:
        o = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
        lam = sch.get_cosine_schedule_with_warmup(o, **kw).lr_lambdas[0]
        out["schedules"].append(dict(kw=kw, factors=[lam(i) for i in range(kw["num_training_steps"] + 3)]))
    model = Toy()
    names = {id(p): n for n, p in model.named_parameters()}
    t = opt.add_weight_decay(model, 0.05, NO_DECAY, True)
    t = opt.add_different_lr(t, DIFF["names"], DIFF["lr"], DIFF["default"])
    out["tuples"] = [[n, wd, lr] for n, _, wd, lr in t]
    out["groups"] = [dict(weight_decay=g["weight_decay"], lr=g["lr"], params=[names[id(p)] for p in g["params"]])
                     for g in opt.create_optimizer_params_group(t)]
    t2 = opt.add_weight_decay(model, 0.1, (), False)                  # filter off: only the explicit list (empty) exempts
    out["tuples_nofilter"] = [[n, wd] for n, _, wd in t2]
    path = os.path.join(HERE, "mm_optim.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, len(out["tuples"]), "parameters,", len(out["groups"]), "groups")


if __name__ == "__main__":
    main()
