"""Per-step learning-rate / weight-decay schedules of the InternVideo2 recipes (host side, numpy).

  * `cosine_scheduler` -- single_modality/utils.py:468-485: linear warm-up from `start_warmup_value` to `base_value` over
    `warmup_epochs * niter_per_ep` (or `warmup_steps`) iterations, then a half cosine down to `final_value`; one value per iteration.
  * `scale_lr` -- single_modality/run_pretraining.py:349-353: lr, min_lr and warmup_lr are quoted per 256 clips and scaled by the
    global batch (`batch_size * world_size * num_sample / 256`).
`IVTrainEngine.train_step(..., lr=lr_schedule[it], weight_decay=wd_schedule[it])` consumes them the way
engines/engine_for_pretraining.py:56-61 assigns `param_group["lr"] = lr_schedule_values[it] * lr_scale` (lr_scale = 1 in pre-training:
layer-wise decay is a fine-tuning feature, optim_factory.py:24-53).
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
