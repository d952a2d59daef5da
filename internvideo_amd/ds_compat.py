"""The DeepSpeed-engine surface the reference's pre-training scripts drive, on IVTrainEngine.

The reference hands its student to `deepspeed.initialize` (InternVideo2/single_modality/run_pretraining.py:363-375, `ds_init`) and from then on
talks to what comes back -- inside `train_one_epoch` (engines/engine_for_pretraining.py:17-199, "E:") and the checkpoint helpers
(utils.py:500-519, 688-700) -- through a small protocol:

    model.train() / model.zero_grad() / model.micro_steps = 0                      E:34, 44-45
    optimizer.param_groups[i]["lr" | "lr_scale" | "weight_decay"]                  E:56-61   (per-step schedule write-back), E:172-184 (logging)
    outputs = model(videos.bfloat16(), bool_masked_pos)                            E:127-128 (three l2-normalised outputs; the loop builds the loss)
    model.backward(loss) ; model.step()                                            E:164-165
    model.optimizer.loss_scale | cur_scale, model.optimizer._global_grad_norm      E:10-17   (get_loss_scale_for_deepspeed)
    model.gradient_accumulation_steps()                                            run_pretraining.py:373-375
    model.save_checkpoint(save_dir, tag, client_state) / model.load_checkpoint(dir, tag) -> (path, client_state)     utils.py:519, 695

`initialize(args=, model=, model_parameters=)` returns exactly that tuple shape `(engine, optimizer, None, None)`; a maintainer swaps
`ds_init = deepspeed.initialize` for `ds_init = internvideo_amd.ds_compat.initialize` and the loop runs unchanged:
`backward` is IVTrainEngine.backward (hand-derived kernels, weight gradients written into the flat buffers, buckets reduced over RCCL as
backward reaches them), `step` finishes the reduction and launches clip + fused AdamW with the lr / weight decay the loop just wrote into
the parameter groups.  The recorded call trace of the reference loop that this class is held to: tests/golden/step_protocol.json
(tests/golden/make_golden_step_protocol.py runs the reference's own train_one_epoch against recording fakes)."""
from __future__ import annotations

import os
from typing import Any, Dict, List, Optional

import torch

MODEL_STATES = "mp_rank_00_model_states.pt"                 # DeepSpeed's file name for the (unsharded) module weights


class _OptimizerView:
    """what the loop and the logging code read / write on `optimizer` (a DeepSpeed engine's `.optimizer` and the `optimizer` that
    ds_init returns are the same object for these purposes): torch-style param_groups with the reference's extra `lr_scale` key."""

    def __init__(self, param_groups: List[Dict[str, Any]], lr: float):
        self.param_groups = []
        for g in param_groups:
            g = dict(g)
            g.setdefault("lr_scale", 1.0)
            g.setdefault("weight_decay", 0.0)
            g.setdefault("lr", lr * g["lr_scale"])
            self.param_groups.append(g)
        self._global_grad_norm = None                       # DeepSpeed sets it in step(); E:17 reads it
        self.loss_scale = 1.0                               # bf16: no loss scaling (E:12-15 reads `loss_scale` or `cur_scale`)
        self.cur_scale = 1.0

    def zero_grad(self, set_to_none: bool = True):          # noqa: ARG002  (the engine zeroes its flat buffers itself)
        return None

    def base_lr(self) -> Optional[float]:
        """the schedule value the loop wrote this step: E:58 stores lr_schedule_values[it] * lr_scale per group, so lr / lr_scale is ONE number
        over all groups (the per-group factor is applied inside the optimizer kernel from the engine's lr_scales table); groups that disagree
        were not written by that loop and would silently train at the first group's rate -- refused"""
        base = None
        for g in self.param_groups:
            sc = float(g.get("lr_scale", 1.0))
            if not sc:
                continue
            b = float(g["lr"]) / sc
            if base is None:
                base = b
            elif abs(b - base) > 1e-6 * max(abs(base), 1e-30):
                raise RuntimeError(f"IVDeepSpeedEngine: parameter groups disagree on lr / lr_scale ({base} vs {b}); the reference loop writes "
                                   "lr_schedule_values[it] * lr_scale into every group (engine_for_pretraining.py:56-58)")
        return base

    def weight_decay(self) -> Optional[float]:
        """E:59-60 writes the scheduled value into every group that decays at all"""
        for g in self.param_groups:
            if g["weight_decay"] > 0:
                return float(g["weight_decay"])
        return None


class IVDeepSpeedEngine:
    """`model` as the reference's loop sees it after ds_init.  Wraps the student (`module`) and an IVTrainEngine around it."""

    def __init__(self, module, model_parameters=None, lr: float = 1.5e-4, betas=(0.9, 0.98), eps: float = 1e-6, weight_decay: float = 0.05,
                 max_grad_norm: float = 3.0, gradient_accumulation_steps: int = 1, engine_cls=None, **engine_kw):
        """gradient_accumulation_steps > 1 (`--update_freq`, run_pretraining.py:42,375; deepspeed==0.10.1 runtime/engine.py: backward scales the
        loss by 1 / gas, step() only acts when micro_steps is a multiple of gas and zeroes the gradients there): every micro-step's gradients are
        added to fp32 accumulators (IVTrainEngine.accumulate), the ranks' sums meet at the boundary, then clip + AdamW with the schedule values
        the loop wrote for THAT iteration."""
        if int(gradient_accumulation_steps) < 1:
            raise ValueError(f"gradient_accumulation_steps = {gradient_accumulation_steps}")
        if engine_cls is None:
            from .engine import IVTrainEngine as engine_cls        # noqa: N813
        self.module = module
        if int(gradient_accumulation_steps) > 1:
            engine_kw.setdefault("overlap", False)               # nothing goes to the wire before the boundary
        if model_parameters is None:                           # optim_factory.get_parameter_groups (:56-98) without layer decay
            skip = set(module.no_weight_decay()) if hasattr(module, "no_weight_decay") else set()
            decay, no_decay = [], []
            for n, p in module.named_parameters():
                if p.requires_grad:
                    (no_decay if (p.dim() == 1 or n.endswith(".bias") or n in skip) else decay).append(p)
            model_parameters = [dict(params=decay, weight_decay=weight_decay, lr_scale=1.0), dict(params=no_decay, weight_decay=0.0, lr_scale=1.0)]
        self.optimizer = _OptimizerView(list(model_parameters), lr)
        # layer-wise lr decay (optim_factory.get_parameter_groups `lr_scale`, run_finetuning.py:548-549): the groups carry one factor per
        # parameter; the engine applies it inside the optimizer kernel from a name -> scale table.  Without this every layer would silently
        # train at scale 1.
        if "lr_scales" not in engine_kw and "layer_decay" not in engine_kw:
            by_id = {id(p): float(g.get("lr_scale", 1.0)) for g in self.optimizer.param_groups for p in g.get("params", [])}
            table = {n: by_id[id(p)] for n, p in module.named_parameters() if id(p) in by_id}
            if any(v != 1.0 for v in table.values()):
                engine_kw["lr_scales"] = lambda name, _t=table: _t.get(name, 1.0)
        self.engine = engine_cls(module, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm, **engine_kw)
        self.micro_steps = 0
        self.global_steps = 0
        self._gas = int(gradient_accumulation_steps)
        self._stepped = True                                   # the flat gradient buffers are clean

    # ---- nn.Module-like surface ------------------------------------------------------------------------------------------------
    def train(self, mode: bool = True):
        self.module.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def parameters(self):
        return self.module.parameters()

    def named_parameters(self):
        return self.module.named_parameters()

    def state_dict(self, *a, **k):
        return self.module.state_dict(*a, **k)

    def no_weight_decay(self):
        return self.module.no_weight_decay()

    def __getattr__(self, name):                               # anything else the scripts read off the model (get_num_layers, ...)
        if name in ("module", "engine", "optimizer"):
            raise AttributeError(name)
        return getattr(self.module, name)

    def gradient_accumulation_steps(self) -> int:
        return self._gas

    def zero_grad(self):
        self.engine.zero_grad()
        self._stepped = True

    # ---- the step, as E:127-165 drives it --------------------------------------------------------------------------------------
    def __call__(self, *args, **kwargs):
        if self._stepped:                                      # first forward after a step: fresh accumulators, next dropout epoch
            self.engine.zero_grad()
            self.engine._begin_step_on_device()
            self._stepped = False
        return self.module(*args, **kwargs)

    forward = __call__

    def backward(self, loss):
        if self._gas > 1:                                      # deepspeed runtime/engine.py _scale_loss_by_gas
            self.engine.backward(loss / float(self._gas))
            self.engine.accumulate()
        else:
            self.engine.backward(loss)
        self.micro_steps += 1
        return loss

    def is_gradient_accumulation_boundary(self) -> bool:
        return self.micro_steps % self._gas == 0

    def step(self):
        eng = self.engine
        self._stepped = True                                   # the next forward starts from clean per-micro-step buffers either way
        if self._gas > 1:
            if not self.is_gradient_accumulation_boundary():   # DeepSpeed's step() between boundaries: nothing is applied
                return
            eng.optimizer_step(self.optimizer.base_lr(), self.optimizer.weight_decay(), accumulated=True)
        else:
            eng._finish_reduce()
            if getattr(eng, "_defer_reduce", False):
                eng.reduce_all_now()
            eng.optimizer_step(self.optimizer.base_lr(), self.optimizer.weight_decay())
        self.optimizer._global_grad_norm = eng.grad_norm       # a device scalar: logging it is the caller's sync, not the step's
        self.global_steps += 1

    # ---- checkpoints (utils.py:500-519 save_model, :688-700 load_specific_model) -------------------------------------------------
    def save_checkpoint(self, save_dir, tag=None, client_state=None, save_latest=True):
        tag = tag or f"global_step{self.global_steps}"
        path = os.path.join(save_dir, str(tag))
        if hasattr(self.engine, "consolidate"):
            self.engine.consolidate()                          # COLLECTIVE for zero1 engines: DeepSpeed's save_checkpoint is called on every rank too
        rank = getattr(self.engine, "rank", 0)
        if rank == 0:
            os.makedirs(path, exist_ok=True)
            # (1) what the NEXT stage of the reference reads: DeepSpeed's model-states file with the NAMED weights under 'module'
            #     (run_finetuning.py:385-388 / utils.py:568-647 load `--finetune` through model_key 'model|module'); host tensors, fp32 as held
            named = {k: v.detach().to("cpu") for k, v in self.module.state_dict().items()}
            torch.save({"module": named, "global_steps": self.global_steps, "client_state": dict(client_state or {}), **dict(client_state or {})},
                       os.path.join(path, MODEL_STATES))
            # (2) this engine's resume state: flat fp32 master / moments + a fingerprint of the layout they were written under
            torch.save({"engine": self.engine.state_dict(), "client_state": dict(client_state or {}), "global_steps": self.global_steps,
                        "micro_steps": self.micro_steps,
                        "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.optimizer.param_groups]},
                       os.path.join(path, "ivh_engine_states.pt"))
            if save_latest:
                with open(os.path.join(save_dir, "latest"), "w") as f:
                    f.write(str(tag))
        return True

    def load_checkpoint(self, load_dir, tag=None, load_module_only: bool = False, **_unused):
        """-> (path, client_state).  Resumes from this engine's own state file when it is there and its layout fingerprint matches; a
        directory that only holds DeepSpeed's `mp_rank_00_model_states.pt` (a checkpoint written by the reference itself, or
        load_module_only=True) loads the named weights and starts the optimizer state afresh."""
        if tag is None:
            latest = os.path.join(load_dir, "latest")
            if not os.path.isfile(latest):
                return None, None
            with open(latest) as f:
                tag = f.read().strip()
        path = os.path.join(load_dir, str(tag), "ivh_engine_states.pt")
        named = os.path.join(load_dir, str(tag), MODEL_STATES)
        if os.path.isfile(path) and not load_module_only:
            sd = torch.load(path, map_location=self.engine.device, weights_only=False)
            self.engine.load_state_dict(sd["engine"])         # raises when the flat layout is not this engine's
            self.global_steps = int(sd.get("global_steps", 0))
            self.micro_steps = int(sd.get("micro_steps", self.global_steps * self._gas))
            for g, saved in zip(self.optimizer.param_groups, sd.get("param_groups", [])):
                g.update(saved)
            return path, sd.get("client_state", {})
        if os.path.isfile(named):
            sd = torch.load(named, map_location="cpu", weights_only=False)
            self.module.load_state_dict(sd["module"], strict=True)     # (the engine's load_state_dict post-hook refreshes its bf16 copy)
            if hasattr(self.engine, "sync_shadow"):
                self.engine.sync_shadow()
            self.global_steps = int(sd.get("global_steps", 0))
            return named, sd.get("client_state", {})
        return None, None


def _init_distributed():
    """`dist_init_required=True` (run_pretraining.py:368 passes `not args.distributed`): DeepSpeed then creates the process group itself from
    the launcher's environment; a single process gets a 1-rank group, so that the loop's dist.get_world_size() / all_gather of the loss
    (engine_for_pretraining.py:151-158) work."""
    import torch.distributed as dist
    if not dist.is_available() or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo", rank=rank, world_size=world)


def initialize(args=None, model=None, model_parameters=None, dist_init_required=None, engine_cls=None, **engine_kw):
    """drop-in for `deepspeed.initialize` as run_pretraining.py:366-369 calls it.  Optimizer hyper-parameters come from `args` the way
    utils.create_internvideo2_ds_config (utils.py:803-871) writes them into the DeepSpeed JSON: lr, weight_decay, opt_betas, opt_eps, clip_grad,
    update_freq, zero_stage (1 -> the engine's ZeRO-1 mode: sharded fp32 state, utils.py:863-903).  -> (engine, optimizer, None, None)"""
    g = (lambda k, d: getattr(args, k, d) if args is not None else d)
    if dist_init_required:
        _init_distributed()
    if int(g("zero_stage", 0) or 0) == 1 and "reduce_mode" not in engine_kw and engine_cls is None:
        engine_kw["reduce_mode"] = "zero1"
    betas = g("opt_betas", None) or (0.9, 0.98)
    eng = IVDeepSpeedEngine(model, model_parameters=model_parameters, lr=float(g("lr", 1.5e-4)), betas=(float(betas[0]), float(betas[1])),
                            eps=float(g("opt_eps", 1e-6)), weight_decay=float(g("weight_decay", 0.05)),
                            max_grad_norm=float(g("clip_grad", 3.0) or 0.0), gradient_accumulation_steps=int(g("update_freq", 1)),
                            engine_cls=engine_cls, **engine_kw)
    return eng, eng.optimizer, None, None
