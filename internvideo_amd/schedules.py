"""Per-step learning-rate / weight-decay schedules of the InternVideo2 recipes (host side, numpy).

  * `cosine_scheduler` -- single_modality/utils.py:468-485: linear warm-up from `start_warmup_value` to `base_value` over
    `warmup_epochs * niter_per_ep` (or `warmup_steps`) iterations, then a half cosine down to `final_value`; one value per iteration.
  * `scale_lr` -- single_modality/run_pretraining.py:349-353: lr, min_lr and warmup_lr are quoted per 256 clips and scaled by the
    global batch (`batch_size * world_size * num_sample / 256`).
`IVTrainEngine.train_step(..., lr=lr_schedule[it], weight_decay=wd_schedule[it])` consumes them the way
engines/engine_for_pretraining.py:56-61 assigns `param_group["lr"] = lr_schedule_values[it] * lr_scale` (lr_scale = 1 in pre-training).
  * `layer_id_for_vit` / `LayerDecayValueAssigner` -- single_modality/optim_factory.py:24-53 and run_finetuning.py:548-549: the layer-wise
    lr decay of the fine-tuning recipe.  `IVTrainEngine(model, layer_decay=0.75)` (or `lr_scales=name -> scale`) turns it into a per-segment
    table beside the flat buffers, applied inside the fused AdamW kernel (`ivh_adamw_step_scaled`).
  * stage 2 (multi_modality/utils/scheduler.py:26-60, utils/optimizer.py:17-84): `cosine_warmup_factor` is the LambdaLR multiplier of the
    recipe's schedule (scripts/pretraining/stage2/1B/config.py:107: cosine, min_lr_multi 0.01, one warm-up epoch); `add_weight_decay` /
    `add_different_lr` / `create_optimizer_params_group` describe its parameter groups (no decay for 1-D tensors, `.bias` and the model's
    no-decay list; a different lr for parameters whose name matches one of `different_lr.module_names` as a regular expression);
    `different_lr_scales` hands the latter to `IVTrainEngine(lr_scales=...)`.
"""
from __future__ import annotations

import math

import numpy as np


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1):
    total = int(epochs * niter_per_ep)
    warmup_iters = int(warmup_steps) if warmup_steps > 0 else int(warmup_epochs * niter_per_ep)
    warm = np.linspace(start_warmup_value, base_value, warmup_iters) if warmup_epochs > 0 else np.array([])
    n = total - warmup_iters
    i = np.arange(n)
    decay = np.array([final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * k / n)) for k in i])
    schedule = np.concatenate((warm, decay))
    assert len(schedule) == total
    return schedule


def scale_lr(lr: float, batch_size: int, world_size: int, num_sample: int = 1) -> float:
    """run_pretraining.py:349-353: lr * (batch_size * world_size) * num_sample / 256"""
    return lr * (batch_size * world_size) * num_sample / 256


_EMBED_NAMES = {"cls_token", "mask_token", "pos_embed", "class_embedding", "positional_embedding", "temporal_positional_embedding"}


def layer_id_for_vit(var_name: str, num_max_layer: int) -> int:
    """optim_factory.get_num_layer_for_vit (:24-42): embeddings are layer 0, `blocks.i.*` / `transformer.resblocks.i.*` layer i + 1, the relative
    position bias and everything after the stack (norms, heads, projectors) the last layer."""
    if var_name in _EMBED_NAMES or var_name.startswith(("patch_embed", "conv1")):
        return 0
    parts = var_name.split(".")
    if parts[0] == "blocks":
        return int(parts[1]) + 1
    if var_name.startswith("transformer.resblocks"):
        return int(parts[2]) + 1
    return num_max_layer - 1


class LayerDecayValueAssigner:
    """optim_factory.LayerDecayValueAssigner (:45-53): values[layer id] is that layer's lr_scale"""

    def __init__(self, values):
        self.values = list(values)

    @classmethod
    def for_depth(cls, num_layers: int, layer_decay: float):
        """run_finetuning.py:548-549: layer_decay ** (num_layers + 1 - i) for i in 0 .. num_layers + 1"""
        return cls(layer_decay ** (num_layers + 1 - i) for i in range(num_layers + 2))

    def get_scale(self, layer_id: int) -> float:
        return self.values[layer_id]

    def get_layer_id(self, var_name: str) -> int:
        return layer_id_for_vit(var_name, len(self.values))


def parameter_groups(named_parameters, weight_decay=1e-5, skip_list=(), get_num_layer=None, get_layer_scale=None):
    """optim_factory.get_parameter_groups (:56-98) as a host-side description: {group name: {"weight_decay", "lr_scale", "params": [names]}}.
    Used by the tests to build the torch.optim.AdamW the engine is compared with; the engine itself needs only name -> lr_scale."""
    groups = {}
    for name, p in named_parameters:
        if not p.requires_grad:
            continue
        no_decay = p.dim() == 1 or name.endswith(".bias") or name in skip_list
        gname = "no_decay" if no_decay else "decay"
        layer_id = None
        if get_num_layer is not None:
            layer_id = get_num_layer(name)
            gname = f"layer_{layer_id}_{gname}"
        if gname not in groups:
            groups[gname] = {"weight_decay": 0.0 if no_decay else weight_decay, "params": [],
                             "lr_scale": get_layer_scale(layer_id) if get_layer_scale is not None else 1.0}
        groups[gname]["params"].append(name)
    return groups


# ----------------------------------------------------------------------------------------------------------------------------------
# stage 2 (multi_modality)
# ----------------------------------------------------------------------------------------------------------------------------------
def cosine_warmup_factor(current_step: int, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5,
                         min_lr_multi: float = 0.0) -> float:
    """utils/scheduler.py:53-58: lr multiplier at `current_step` -- linear 0 -> 1 over the warm-up, then cosine to 0 over the rest, never below
    min_lr_multi (the floor applies inside the warm-up too).  `IVTrainEngine.train_step(..., lr=base_lr * cosine_warmup_factor(it, ...))`."""
    if current_step < num_warmup_steps:
        return max(min_lr_multi, float(current_step) / float(max(1, num_warmup_steps)))
    progress = float(current_step - num_warmup_steps) / float(max(1, num_training_steps - num_warmup_steps))
    return max(min_lr_multi, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


def add_weight_decay(model, weight_decay, no_decay_list=(), filter_bias_and_bn=True):
    """utils/optimizer.py:17-28 -> [[name, param, weight_decay]] for the trainable parameters, in named_parameters() order."""
    out = []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if filter_bias_and_bn and (len(param.shape) == 1 or name.endswith(".bias")):
            out.append([name, param, 0])
        elif name in no_decay_list:
            out.append([name, param, 0])
        else:
            out.append([name, param, weight_decay])
    return out


def add_different_lr(named_param_tuples, diff_lr_names, diff_lr, default_lr):
    """utils/optimizer.py:31-62 -> [[name, param, weight_decay, lr]]: `diff_lr` where re.search(pattern, name) hits for any pattern."""
    import re
    out = []
    for name, p, wd in named_param_tuples:
        hit = any(re.search(pat, name) is not None for pat in diff_lr_names)
        out.append([name, p, wd, diff_lr if hit else default_lr])
    return out


def create_optimizer_params_group(named_param_tuples_with_lr):
    """utils/optimizer.py:65-84: one group per (weight_decay, lr) pair, weight-decay values in first-seen order, lrs in first-seen order
    inside each -> [dict(params=[...], weight_decay=wd, lr=lr)] ready for torch.optim.AdamW."""
    group = {}
    for _, p, wd, lr in named_param_tuples_with_lr:
        group.setdefault(wd, {}).setdefault(lr, []).append(p)
    return [dict(params=ps, weight_decay=wd, lr=lr) for wd, by_lr in group.items() for lr, ps in by_lr.items()]


def different_lr_scales(diff_lr_names, diff_lr: float, default_lr: float):
    """name -> lr_scale for `IVTrainEngine(lr=default_lr, lr_scales=...)`: the `different_lr` block of the stage-2 config
    (scripts/pretraining/stage2/1B/config.py:104) as the per-segment factor of the fused AdamW."""
    import re
    pats = [re.compile(p) for p in diff_lr_names]
    ratio = float(diff_lr) / float(default_lr)
    return lambda name: ratio if any(p.search(name) is not None for p in pats) else 1.0
