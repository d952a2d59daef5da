"""The reference's step loop as a PROTOCOL (CPU, no GPU).

tests/golden/step_protocol.json is the call trace of the reference's own `train_one_epoch` (InternVideo2/single_modality/engines/
engine_for_pretraining.py:17-199, "E:") run by tests/golden/make_golden_step_protocol.py with recording stand-ins for what the loop is handed
(DeepSpeed-shaped model wrapper, teachers, optimizer).  Here the facts a drop-in must honour are read off that trace, the host-side mask logic
is held to the masks the loop produced, and the recorded call sequence is replayed against internvideo_amd.ds_compat.IVDeepSpeedEngine (what
`ds_init` returns on this side) with a recording engine core: every call the loop makes exists, and is translated into the native engine's
step in the right order with the schedule values of that step.  The HIP-side replay (losses against the reference trajectory) is
tests/test_model_gpu.py::test_reference_step_loop_protocol_on_the_hip_path."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from internvideo_amd import ds_compat, masking  # noqa: E402
from oracle import internvideo2_oracle as O  # noqa: E402

FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "step_protocol.json")))
TRACE = FIX["trace"]


def teacher_features(step, cfg, B):
    """the seeded teacher outputs of the fixture generator (make_golden_step_protocol.fake_features): clip taps, clip final, attention, mae taps"""
    g = torch.Generator().manual_seed(FIX["teacher_seed_base"] + step)
    T, h, w = cfg.grid
    N = h * w
    unit = lambda *s: torch.nn.functional.normalize(torch.randn(*s, generator=g), dim=-1)      # noqa: E731
    clip_mid = unit(cfg.clip_return_layer, B, 1 + T * N, cfg.clip_teacher_embed_dim)
    clip_fin = unit(B, cfg.clip_teacher_final_dim)
    attn = torch.rand(B * T, N, generator=g) + 0.05
    mae = unit(cfg.mae_return_layer, B, T * N, cfg.mae_teacher_embed_dim)
    return clip_mid, clip_fin, attn / attn.sum(-1, keepdim=True), mae


def per_step():
    steps, cur = [], None
    for e in TRACE:
        if e["call"] == "clip_teacher":
            cur = {}
            steps.append(cur)
        if cur is not None:
            cur[e["call"]] = e
    return steps


def test_what_the_reference_loop_hands_to_and_expects_from_the_model():
    cfg = O.named_config(FIX["config"])
    B, T, (Tt, h, w) = FIX["batch"], cfg.num_frames, cfg.grid
    N = h * w
    calls = [e["call"] for e in TRACE]
    # E:34,44-45: train(), ONE zero_grad() before the loop (the engine zeroes its own gradients in step()); then per batch, in this order
    assert calls[:2] == ["model.train", "model.zero_grad"] and calls.count("model.zero_grad") == 1 and FIX["micro_steps_after"] == 0
    assert calls[2:] == ["clip_teacher", "mae_teacher", "model.__call__", "model.backward", "model.step"] * FIX["steps"]
    n_vis = N - int(N * FIX["mask_ratio"])
    for i, st in enumerate(per_step()):
        # E:81-103: the CLIP teacher sees every td_ratio-th frame, the MAE teacher all of them, both in the loader's dtype (autocast inside)
        assert st["clip_teacher"]["videos"] == dict(dtype="float32", shape=[B, 3, T, cfg.img_size, cfg.img_size])
        assert st["mae_teacher"]["videos"] == dict(dtype="float32", shape=[B, 3, T * FIX["td_ratio"], cfg.img_size, cfg.img_size])
        c = st["model.__call__"]
        # E:127-128: the student gets the CLIP teacher's frames as bfloat16 and a BOOL mask (B, 1 + T N) whose cls column is never masked
        assert c["videos"] == dict(dtype="bfloat16", shape=[B, 3, T, cfg.img_size, cfg.img_size])
        assert c["mask"]["dtype"] == "bool" and c["mask"]["shape"] == [B, 1 + T * N] and not c["mask"]["cls_column_masked"]
        assert c["mask"]["visible_per_sample"] == [1 + T * n_vis] * B                     # E:105-116: the same count on every frame of every clip
        # E:56-61: the step's schedule values are in the parameter groups BEFORE the forward; weight decay only where it is > 0
        assert c["lr_in_groups"] == [FIX["lr_schedule"][i]] * 2 and c["wd_in_groups"] == [FIX["wd_schedule"][i], 0.0]
        # three outputs, in this order, l2-normalised student features at the visible tokens (E:128; shapes of the targets E:118-125)
        assert [o["shape"] for o in c["outputs"]] == [[cfg.clip_return_layer, B, 1 + T * n_vis, cfg.clip_teacher_embed_dim], [B, cfg.clip_teacher_final_dim],
                                                     [cfg.mae_return_layer, B, T * n_vis, cfg.mae_teacher_embed_dim]]
        # E:150,164-165: ONE scalar loss goes to model.backward, then model.step; the loss is fp32 (bf16 outputs x fp32 targets)
        assert st["model.backward"]["loss_shape"] == [] and st["model.backward"]["loss_dtype"] == "float32"
        assert st["model.step"]["lr_applied"] == FIX["lr_schedule"][i] and st["model.step"]["wd_applied"] == FIX["wd_schedule"][i]


def test_attention_guided_masks_equal_the_ones_the_reference_loop_drew():
    """E:105-116 on the host: `torch.multinomial(attn, N)` from torch's global RNG, first N_vis kept, cls column prepended.  masking.
    attention_guided_mask consumes the RNG exactly as the loop does: under the fixture's seed it reproduces the loop's masks bit for bit."""
    if torch.__version__ != FIX["torch_version"]:
        pytest.skip("the multinomial draw is pinned to the torch build that generated the fixture")
    cfg = O.named_config(FIX["config"])
    B = FIX["batch"]
    torch.manual_seed(FIX["mask_rng_seed"])
    for i, st in enumerate(per_step()):
        attn = teacher_features(i, cfg, B)[2]
        mask = masking.attention_guided_mask(attn, B, FIX["mask_ratio"])
        assert mask.dtype == torch.bool and list(mask.shape) == st["model.__call__"]["mask"]["shape"]
        assert np.packbits(mask.numpy(), axis=1).tolist() == st["model.__call__"]["mask"]["packed"], f"step {i}"


class _Core:
    """recording stand-in for IVTrainEngine (the compute needs a GPU): the calls ds_compat makes on it, in order"""

    def __init__(self, module, **kw):
        self.module, self.kw, self.log, self.device, self.rank = module, kw, [], torch.device("cpu"), 0
        self.grad_norm = torch.tensor(0.0)

    def zero_grad(self):
        self.log.append("zero_grad")

    def _begin_step_on_device(self):
        self.log.append("begin_step")

    def backward(self, loss):
        self.log.append(("backward", float(loss.detach())))
        loss.backward()

    def _finish_reduce(self):
        self.log.append("finish_reduce")

    def optimizer_step(self, lr=None, weight_decay=None):
        self.log.append(("optimizer_step", lr, weight_decay))
        self.grad_norm = torch.tensor(1.25)


class _Student(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Linear(4, 4)
        self.seen = []

    def no_weight_decay(self):
        return {"w.bias"}

    def forward(self, videos, mask):
        self.seen.append((videos.dtype, mask.dtype))
        s = self.w(torch.ones(1, 4)).sum()
        return s * torch.ones(2, 1, 3, 5), s * torch.ones(1, 5), s * torch.ones(2, 1, 2, 5)


def test_the_recorded_call_sequence_drives_the_deepspeed_shaped_adapter():
    """every call of the trace, made on ds_compat.initialize(...)'s engine exactly as the loop makes them (the four schedule lines E:56-61 are
    restated here), lands on the native engine as: fresh accumulators + dropout epoch at the first forward of a step, backward, then
    _finish_reduce -> optimizer_step(lr, weight_decay) with THAT step's schedule values; and what E:10-17 reads afterwards is there"""
    from types import SimpleNamespace
    args = SimpleNamespace(lr=FIX["lr"], weight_decay=FIX["weight_decay"], opt_betas=FIX["betas"], opt_eps=FIX["eps"], clip_grad=FIX["clip"], update_freq=1)
    student = _Student()
    model, optimizer, _, _ = ds_compat.initialize(args=args, model=student, model_parameters=None, dist_init_required=False, engine_cls=_Core)
    core = model.engine
    assert core.kw == dict(lr=FIX["lr"], betas=tuple(FIX["betas"]), eps=FIX["eps"], weight_decay=FIX["weight_decay"], max_grad_norm=FIX["clip"])
    assert model.gradient_accumulation_steps() == 1                                   # run_pretraining.py:373-375
    assert [sorted(k for k in g if k != "params") for g in optimizer.param_groups] == [["lr", "lr_scale", "weight_decay"]] * 2
    assert [g["weight_decay"] for g in optimizer.param_groups] == [FIX["weight_decay"], 0.0] and [len(g["params"]) for g in optimizer.param_groups] == [1, 1]
    it = -1
    for e in TRACE:
        c = e["call"]
        if c == "model.train":
            model.train()
            assert student.training
        elif c == "model.zero_grad":
            model.zero_grad()
            model.micro_steps = 0                                                       # E:45
        elif c == "clip_teacher":
            it += 1
            for group in optimizer.param_groups:                                        # E:56-61
                group["lr"] = FIX["lr_schedule"][it] * group["lr_scale"]
                if group["weight_decay"] > 0:
                    group["weight_decay"] = FIX["wd_schedule"][it]
        elif c == "model.__call__":
            vid = torch.zeros(e["videos"]["shape"]).bfloat16()                           # E:127
            msk = torch.from_numpy(np.unpackbits(np.array(e["mask"]["packed"], dtype=np.uint8), axis=1)[:, :e["mask"]["shape"][1]].astype(bool))
            out = model(vid, msk)
            assert len(out) == 3
            loss = sum(o.float().mean() for o in out)
        elif c == "model.backward":
            model.backward(loss)
        elif c == "model.step":
            model.step()
            # E:10-17 get_loss_scale_for_deepspeed
            assert model.optimizer.loss_scale == 1.0 and float(model.optimizer._global_grad_norm) == 1.25
    assert student.seen == [(torch.bfloat16, torch.bool)] * FIX["steps"]
    want = ["zero_grad"]                                                                 # the loop's one model.zero_grad()
    for i in range(FIX["steps"]):
        want += ["zero_grad", "begin_step", "backward", "finish_reduce", ("optimizer_step", FIX["lr_schedule"][i], FIX["wd_schedule"][i])]
    got = [x if not (isinstance(x, tuple) and x[0] == "backward") else "backward" for x in core.log]
    assert got == want, (got, want)
    assert model.micro_steps == FIX["steps"] and model.global_steps == FIX["steps"]


def test_adapter_checkpoints_round_trip_the_client_state(tmp_path):
    """utils.py:500-519 / :688-700: save_checkpoint(save_dir, tag, client_state) on every rank, load_checkpoint(dir, tag) -> (path, client_state)"""
    class Core(_Core):
        def state_dict(self):
            return {"step": 7}

        def load_state_dict(self, sd):
            self.loaded = sd

        def consolidate(self):
            self.log.append("consolidate")
    model, _, _, _ = ds_compat.initialize(model=_Student(), engine_cls=Core)
    model.global_steps = 7
    assert model.save_checkpoint(save_dir=str(tmp_path), tag="checkpoint-latest", client_state={"epoch": 3})
    m2, _, _, _ = ds_compat.initialize(model=_Student(), engine_cls=Core)
    path, client = m2.load_checkpoint(str(tmp_path), tag="checkpoint-latest")
    assert os.path.isfile(path) and client == {"epoch": 3} and m2.engine.loaded == {"step": 7} and m2.global_steps == 7
    assert m2.load_checkpoint(str(tmp_path))[1] == {"epoch": 3}                          # tag from `latest`
    assert model.engine.log[-1] == "consolidate"


# ---- gradient accumulation: `--update_freq 2` (run_pretraining.py:42,375) ------------------------------------------------------------------
FIX2 = json.load(open(os.path.join(ROOT, "tests", "golden", "step_protocol_gas2.json")))


class _AccCore(_Core):
    def accumulate(self):
        self.log.append("accumulate")

    def optimizer_step(self, lr=None, weight_decay=None, accumulated=False):
        self.log.append(("optimizer_step", lr, weight_decay, accumulated))
        self.grad_norm = torch.tensor(1.25)


def test_update_freq_2_trace_of_the_reference_loop_drives_the_adapter():
    """tests/golden/step_protocol_gas2.json: the reference's own train_one_epoch for four iterations with a DeepSpeed-shaped engine whose
    gradient_accumulation_steps is 2 (make_golden_step_protocol.py --update-freq 2: the stand-in restates deepspeed==0.10.1's contract -- loss
    scaled by 1 / 2 in backward, step() a no-op between boundaries).  The loop is oblivious: it calls backward and step every iteration and
    writes EVERY iteration's schedule values.  Replayed on ds_compat: each micro-step's backward sees loss / 2 and is followed by accumulate();
    only iterations 2 and 4 reach the optimizer, with the schedule values of THOSE iterations and accumulated=True."""
    from types import SimpleNamespace
    assert FIX2["update_freq"] == 2 and FIX2["steps"] == 4
    calls = [e["call"] for e in FIX2["trace"]]
    assert calls[2:] == ["clip_teacher", "mae_teacher", "model.__call__", "model.backward", "model.step"] * 4
    assert [e["boundary"] for e in FIX2["trace"] if e["call"] == "model.step"] == [False, True, False, True]
    args = SimpleNamespace(lr=FIX2["lr"], weight_decay=FIX2["weight_decay"], opt_betas=FIX2["betas"], opt_eps=FIX2["eps"], clip_grad=FIX2["clip"], update_freq=2)
    student = _Student()
    model, optimizer, _, _ = ds_compat.initialize(args=args, model=student, model_parameters=None, dist_init_required=False, engine_cls=_AccCore)
    assert model.gradient_accumulation_steps() == 2                                   # run_pretraining.py:373-375 asserts exactly this
    assert model.engine.kw.get("overlap") is False                                   # nothing goes to the wire before the boundary
    it, seen = -1, []
    for e in FIX2["trace"]:
        c = e["call"]
        if c == "model.train":
            model.train()
        elif c == "model.zero_grad":
            model.zero_grad(); model.micro_steps = 0
        elif c == "clip_teacher":
            it += 1
            for group in optimizer.param_groups:
                group["lr"] = FIX2["lr_schedule"][it] * group["lr_scale"]
                if group["weight_decay"] > 0:
                    group["weight_decay"] = FIX2["wd_schedule"][it]
        elif c == "model.__call__":
            out = model(torch.zeros(e["videos"]["shape"]).bfloat16(), torch.zeros(e["mask"]["shape"], dtype=torch.bool))
            loss = sum(o.float().mean() for o in out)
            seen.append(float(loss.detach()))
        elif c == "model.backward":
            model.backward(loss)
        elif c == "model.step":
            model.step()
    want = ["zero_grad"]
    for i in range(4):
        want += ["zero_grad", "begin_step", "backward", "accumulate"]
        if i % 2 == 1:
            want += [("optimizer_step", FIX2["lr_schedule"][i], FIX2["wd_schedule"][i], True)]
    got = [x if not (isinstance(x, tuple) and x[0] == "backward") else "backward" for x in model.engine.log]
    assert got == want, (got, want)
    scaled = [x[1] for x in model.engine.log if isinstance(x, tuple) and x[0] == "backward"]
    assert all(abs(a - b / 2) < 1e-6 * abs(b) for a, b in zip(scaled, seen))        # _scale_loss_by_gas
    assert model.micro_steps == 4 and model.global_steps == 2


def test_checkpoint_holds_the_named_module_weights_and_a_layout_fingerprint(tmp_path):
    """what the reference's NEXT stage reads is DeepSpeed's `mp_rank_00_model_states.pt` with the named weights under 'module'
    (run_finetuning.py:385-388 loads --finetune through model_key 'model|module'); the engine's own flat buffers are only valid under the layout
    they were written with -- a fingerprint (names, offsets, sizes) is saved with them and checked on load."""
    from internvideo_amd.engine import IVTrainEngine

    class Tower(torch.nn.Module):
        def __init__(self, widths=(8, 8)):
            super().__init__()
            self.blocks = torch.nn.ModuleList([torch.nn.Linear(w, w) for w in widths])
            self.head = torch.nn.Linear(widths[-1], 4)

        def no_weight_decay(self):
            return set()

    m1 = Tower()
    model, _, _, _ = ds_compat.initialize(model=m1)
    assert isinstance(model.engine, IVTrainEngine)
    model.global_steps = 3
    assert model.save_checkpoint(save_dir=str(tmp_path), tag="checkpoint-3", client_state={"epoch": 1})
    named = torch.load(os.path.join(str(tmp_path), "checkpoint-3", ds_compat.MODEL_STATES), map_location="cpu", weights_only=False)
    assert set(named["module"]) == set(m1.state_dict()) and named["epoch"] == 1     # the reference reads checkpoint['module'] (utils.py:568-647)
    for k, v in m1.state_dict().items():
        assert torch.equal(named["module"][k], v.detach().cpu())
    # same architecture: the flat state resumes
    m2 = Tower()
    e2, _, _, _ = ds_compat.initialize(model=m2)
    path, client = e2.load_checkpoint(str(tmp_path), tag="checkpoint-3")
    assert path.endswith("ivh_engine_states.pt") and client == {"epoch": 1} and e2.global_steps == 3
    for k, v in m1.state_dict().items():
        assert torch.equal(m2.state_dict()[k], v)
    # another parameter set of the SAME total size per region would load silently wrong: refused
    m3 = Tower()
    m3.blocks[0], m3.head = m3.head, m3.blocks[0]                                    # swaps names <-> shapes; flat sizes may even coincide
    e3, _, _, _ = ds_compat.initialize(model=m3)
    with pytest.raises(RuntimeError, match="laid out for another parameter order"):
        e3.load_checkpoint(str(tmp_path), tag="checkpoint-3")
    # a directory that only holds DeepSpeed's file (a checkpoint written by the reference itself): named weights, fresh optimizer state
    os.remove(os.path.join(str(tmp_path), "checkpoint-3", "ivh_engine_states.pt"))
    m4 = Tower()
    e4, _, _, _ = ds_compat.initialize(model=m4)
    path, client = e4.load_checkpoint(str(tmp_path), tag="checkpoint-3")
    assert path.endswith(ds_compat.MODEL_STATES) and client == {"epoch": 1}
    for k, v in m1.state_dict().items():
        assert torch.equal(m4.state_dict()[k], v)


def test_layer_decay_groups_reach_the_engine_and_inconsistent_groups_are_refused():
    """parameter groups with differing lr_scale (optim_factory.get_parameter_groups under layer decay) become the engine's name ->
    scale table; groups whose lr / lr_scale disagree (not written by the reference loop) raise instead of training at the first group's rate."""
    student = _Student()
    groups = [dict(params=[student.w.weight], weight_decay=0.05, lr_scale=0.5), dict(params=[student.w.bias], weight_decay=0.0, lr_scale=1.0)]
    model, optimizer, _, _ = ds_compat.initialize(model=student, model_parameters=groups, engine_cls=_Core)
    f = model.engine.kw["lr_scales"]
    assert f("w.weight") == 0.5 and f("w.bias") == 1.0 and f("anything else") == 1.0
    for g in optimizer.param_groups:                                                   # E:56-58
        g["lr"] = 2e-4 * g["lr_scale"]
    assert abs(optimizer.base_lr() - 2e-4) < 1e-12
    optimizer.param_groups[0]["lr"] = 7e-4
    with pytest.raises(RuntimeError, match="disagree on lr / lr_scale"):
        optimizer.base_lr()
    from types import SimpleNamespace
    class Tower(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = torch.nn.ModuleList([torch.nn.Linear(8, 8)])

    m2, _, _, _ = ds_compat.initialize(args=SimpleNamespace(zero_stage=1), model=Tower())       # zero_stage 1 -> the engine's ZeRO-1 mode
    assert m2.engine.reduce_mode == "zero1"


# ---- the distillation loop with the InternVideo2 teacher: clip-level attention map (engine_for_distill.py:89-98) ------------------------------
FIXD = json.load(open(os.path.join(ROOT, "tests", "golden", "distill_protocol.json")))


def test_clip_level_attention_masks_equal_the_ones_the_reference_distill_loop_drew():
    """tests/golden/distill_protocol.json / .npz (make_golden_distill_protocol.py): the reference's engine_for_distill.train_one_epoch around the
    reference's own InternVideo2 teacher, whose attention map is (B, T*H*W) -- ONE multinomial draw per clip, N_vis = N - int(N * ratio) kept
    over the whole clip (frames keep different counts).  masking.attention_guided_mask on the stored maps, under the fixture's seed, reproduces
    the loop's masks bit for bit; masking.visible_tokens gives the per-clip count from the map's shape for both teacher layouts."""
    if torch.__version__ != FIXD["torch_version"]:
        pytest.skip("the multinomial draw is pinned to the torch build that generated the fixture")
    g = np.load(os.path.join(ROOT, "tests", "golden", "distill_protocol.npz"))
    B = FIXD["batch"]
    calls = [e for e in FIXD["trace"] if e["call"] == "model.__call__"]
    assert len(calls) == FIXD["steps"] == 2
    torch.manual_seed(FIXD["mask_rng_seed"])
    for i, e in enumerate(calls):
        attn = torch.from_numpy(g[f"attn:{i}"])
        assert tuple(attn.shape) == (B, 64)                                              # clip-level: 4 frames x 16 patches in one row
        mask = masking.attention_guided_mask(attn, B, FIXD["mask_ratio"])
        assert np.packbits(mask.numpy(), axis=1).tolist() == e["mask"]["packed"], f"step {i}"
        L = masking.visible_tokens(attn.shape, B, FIXD["mask_ratio"])
        assert e["mask"]["visible_per_sample"] == [L] * B and L == 17
        per_frame = (~mask[:, 1:]).reshape(B, 4, 16).sum(-1)
        assert (per_frame.sum(1) == 16).all()
    assert any(len(set((~torch.from_numpy(np.unpackbits(np.array(e["mask"]["packed"], dtype=np.uint8), axis=1)[:, 1:65].astype(bool))).reshape(B, 4, 16).sum(-1)[0].tolist())) > 1
               for e in calls), "frames of one clip keep different counts under a clip-level draw"
    assert masking.visible_tokens((B * 8, 256), B, 0.8) == 1 + 8 * 52                    # the per-frame layout of the 1B recipe
    with pytest.raises(ValueError):
        masking.visible_tokens((7, 64), 4, 0.75)
