"""End-to-end parity of the HIP student (internvideo_amd.internvideo2_pretrain) on a real MI355X:
  * against the committed golden fixtures produced by the REFERENCE's own code (tests/golden/student_*.npz),
  * against the CPU oracle on the BASELINE configs (S/14, B/14, 1B) with the same seeded weights and inputs.
Stated tolerances: gather indices bit-exact; head outputs rel-L2 <= 1e-2 (bf16 compute vs fp32
oracle); loss <= 1e-3 relative; parameter gradients rel-L2 <= 3e-2 (bf16 backward)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from internvideo_amd import internvideo2_pretrain as M  # noqa: E402
from oracle import internvideo2_oracle as O  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


def rel(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


ZERO_GRAD = ("clip_projector.cross_attn.k_bias", "clip_projector.norm1_k.bias")


def grad_errors(named_grads, ref_grads):
    """rel-L2 per parameter; the two softmax-shift-invariant biases are compared on an absolute scale."""
    scale = float(torch.as_tensor(ref_grads["clip_projector.cross_attn.q_bias"]).double().norm())
    errs = {}
    for k, g in named_grads.items():
        if k in ZERO_GRAD:
            errs[k] = float(g.double().norm()) / scale * 1e-1        # must stay < 3e-2  <=>  |g| < 0.3 |dq_bias|
            continue
        errs[k] = rel(g, ref_grads[k])
    return errs


def grad_tol(k: str) -> float:
    """rel-L2 tolerance of a parameter gradient vs the fp32 oracle: 3e-2, except everything in front of the 1-query attention pool
    (its q / k projections and the LayerNorms feeding them): those gradients pass through a softmax over ALL tokens of a single mean
    query, and the reference's OWN bf16 run is already 1.8-2.4 % off its fp32 run there (bf16err:clip_projector.* in
    tests/golden/student_*.npz) -> 3 x that, the same rule the golden-fixture tests apply (max(3e-2, 3 x the reference's discrepancy))"""
    if k.startswith(("clip_projector.norm1_", "clip_projector.cross_attn.q", "clip_projector.cross_attn.k")):
        return 6e-2
    return 3e-2


def build(cfg: O.StudentConfig, params, drop_path_rate=0.0, **kw):
    m = M.PretrainInternVideo2(
        img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
        mlp_ratio=cfg.mlp_ratio, num_frames=cfg.num_frames, tubelet_size=cfg.tubelet_size, drop_path_rate=drop_path_rate,
        attn_pool_num_heads=cfg.attn_pool_num_heads, clip_embed_dim=cfg.clip_embed_dim,
        clip_teacher_embed_dim=cfg.clip_teacher_embed_dim, clip_teacher_final_dim=cfg.clip_teacher_final_dim,
        clip_return_layer=cfg.clip_return_layer, mae_teacher_embed_dim=cfg.mae_teacher_embed_dim,
        mae_return_layer=cfg.mae_return_layer, **kw)
    m.load_state_dict(params, strict=True)
    return m.to(DEV).train()


def losses(out, targets):
    oc, of, om = out
    tc, tf, tm = (t.to(oc.device) for t in targets)
    l1 = (2 - 2 * (oc.float() * tc).sum(-1)).mean()
    l2 = (2 - 2 * (of.float() * tf).sum(-1)).mean()
    l3 = (2 - 2 * (om.float() * tm).sum(-1)).mean()
    return l1 + l2 + l3, (l1, l2, l3)


@pytest.mark.parametrize("residual", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["tiny64", "tiny88"])
def test_student_matches_reference_golden(name, residual):
    """outputs, loss and gradients against the reference's own CPU run (tests/golden/student_*.npz).  residual = "bf16": the residual stream
    of the reference's bf16 recipe (model.residual_dtype); same tolerances -- they are calibrated on the reference's own bf16-vs-fp32
    discrepancy (`bf16err:*` in the fixture), which already contains a bf16 residual stream."""
    g = np.load(os.path.join(GOLD, f"student_{name}.npz"))
    B, n_vis, seed = (int(v) for v in g["meta"])
    cfg = O.named_config(name)
    params = O.synthetic_params(cfg, seed=seed)
    video, mask, targets = O.synthetic_batch(cfg, B, n_vis, seed=seed)
    model = build(cfg, params)
    model.residual_dtype = residual
    vis, inv = M.build_gather_indices(torch.from_numpy(mask), DEV)
    assert np.array_equal(vis.cpu().numpy(), g["vis_idx"])                                   # bit exact
    out = model(video.to(DEV), torch.from_numpy(mask))
    assert out[0].dtype == torch.bfloat16
    e = [rel(out[0].float(), g["x_clip_align"]), rel(out[1].float(), g["x_align"]), rel(out[2].float(), g["x_mae_align"])]
    assert max(e) < (2e-2 if residual == "bf16" else 1e-2), e      # stated tolerance of the bf16 stream: 2e-2 on head outputs, loss still 1e-3
    total, parts = losses(out, targets)
    ref = g["losses"]
    assert abs(total.item() - ref[0]) / abs(ref[0]) < 1e-3, (total.item(), ref[0])
    total.backward()
    sd = dict(model.named_parameters())
    worst = {}
    for key in g.files:
        if key.startswith("grad:"):
            k = key[5:]
            worst[k] = rel(sd[k].grad, g[key])
        elif key.startswith("gradnorm:"):
            k = key[9:]
            gr = sd[k].grad
            g2 = gr.reshape(gr.shape[0], -1) if gr.dim() == 5 else (gr.reshape(-1, gr.shape[-1]) if gr.dim() != 2 else gr)
            worst["corner:" + k] = rel(g2[:16, :16], g["gradcorner:" + k])
            worst["norm:" + k] = abs(gr.double().norm().item() - g[key][0]) / g[key][0]
    # full tensors: max(3e-2, 3 x the reference's own bf16-vs-fp32 discrepancy); 16x16 corners (256-element samples of a
    # bf16 gradient, noisier than a whole-tensor norm): floor 5e-2
    def tol(k):
        floor = 5e-2 if k.startswith("corner:") else 3e-2
        if residual == "bf16":                           # stated tolerance of the bf16 stream: 1.5 x the fp32-stream floors
            floor *= 1.5
        return max(floor, 3.0 * float(g["bf16err:" + k][0])) if ("bf16err:" + k) in g.files else floor
    bad = {k: (v, tol(k)) for k, v in worst.items() if v > tol(k)}
    assert not bad, bad


def _oracle_run(cfg, B, n_vis, seed, want_grads):
    params = O.synthetic_params(cfg, seed=seed)
    video, mask, targets = O.synthetic_batch(cfg, B, n_vis, seed=seed)
    p = {k: v.clone().requires_grad_(want_grads) for k, v in params.items()}
    out = O.student_forward(p, video, mask, cfg)
    total, parts = O.distill_losses(out, targets)
    grads = None
    if want_grads:
        total.backward()
        grads = {k: v.grad for k, v in p.items()}
    return params, video, mask, targets, [o.detach() for o in out], total.item(), grads


def _check_against_reference_digest(name, out, loss, B, n_vis, tol):
    """... and against the REFERENCE's own run at this size (tests/golden/student_<name>_digest.npz, make_golden_fullsize.py: first rows in full
    + 16 random projections of every token row of its fp32 CPU outputs, its loss): the same bars as against the oracle"""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"student_{name}_digest.npz"))
    assert [int(x) for x in g["meta"]] == [B, n_vis, 0]
    for key, o in zip(("x_clip_align", "x_align", "x_mae_align"), out):
        rows = o.detach().float().cpu().double().numpy().reshape(-1, o.shape[-1])
        C = rows.shape[1]
        proj = np.random.Generator(np.random.PCG64(777 + C)).standard_normal((C, 16)).astype(np.float32) / np.sqrt(C).astype(np.float32)
        e_rows = np.linalg.norm(rows[:3] - g[key + ":rows"]) / np.linalg.norm(g[key + ":rows"])
        e_proj = np.linalg.norm(rows @ proj.astype(np.float64) - g[key + ":proj"]) / np.linalg.norm(g[key + ":proj"])
        assert e_rows < tol and e_proj < tol, (key, e_rows, e_proj)
    assert abs(loss - float(g["losses"][0])) / float(g["losses"][0]) < 1e-3, (loss, float(g["losses"][0]))


@pytest.mark.parametrize("name,B,n_vis,want_grads,residual", [("S14", 2, 16, True, "fp32"), ("B14", 1, 51, True, "fp32"), ("1B", 1, 52, False, "fp32"),
                                                              ("S14", 2, 16, True, "bf16"), ("1B", 1, 52, False, "bf16")])
def test_student_matches_oracle_on_baseline_configs(name, B, n_vis, want_grads, residual):
    """configs[0..2] of BASELINE.json.  1B: 8 x 224^2, 52 visible tokens per frame (mask 0.8) -> L = 417.  residual "bf16" = the residual
    stream of the reference's bf16 recipe (what bench.py runs by default), held to the same loss bar (1e-3 relative)."""
    cfg = O.named_config(name)
    from internvideo_amd.hostinfo import usable_cores
    torch.set_num_threads(min(usable_cores(), 32))
    params, video, mask, targets, ref_out, ref_loss, ref_grads = _oracle_run(cfg, B, n_vis, 0, want_grads)
    model = build(cfg, params)
    model.residual_dtype = residual
    out = model(video.to(DEV), torch.from_numpy(mask))
    e = [rel(o.float(), r) for o, r in zip(out, ref_out)]
    assert max(e) < (2e-2 if residual == "bf16" else 1e-2), e
    total, _ = losses(out, targets)
    assert abs(total.item() - ref_loss) / abs(ref_loss) < 1e-3, (total.item(), ref_loss)
    _check_against_reference_digest(name, out, total.item(), B, n_vis, 2e-2 if residual == "bf16" else 1e-2)      # S14, B14, 1B
    if want_grads:
        total.backward()
        errs = grad_errors({k: p.grad for k, p in model.named_parameters()}, ref_grads)
        # 3e-2, except the LayerNorms in front of the 1-query attention pool: their gradients pass through a softmax over all
        # tokens of a single mean query and the reference's OWN bf16 run is already 1.8-2.4 % off its fp32 run there
        # (bf16err:clip_projector.norm1_k.weight in tests/golden/student_*.npz) -> 2 x that
        bad = {k: v for k, v in errs.items() if v > grad_tol(k)}
        assert not bad, dict(sorted(bad.items(), key=lambda kv: -kv[1])[:10])


def test_reference_step_loop_protocol_on_the_hip_path():
    """VERDICT r4 next 7, device side.  tests/golden/step_protocol.json is the call trace AND the trajectory of the reference's own
    `train_one_epoch` (engine_for_pretraining.py:17-199) driving the reference's own student for three steps (bf16 weights, fp32 master, clip 3,
    AdamW; seeded teacher outputs; masks drawn by the loop).  The same call sequence is replayed here on what `ds_init` returns on this side --
    internvideo_amd.ds_compat.initialize(...) around the HIP student -- with the loop's own masks, the targets gathered by the HIP gather kernel
    (E:118-125), the schedule values written into the parameter groups before each forward (E:56-61), `model(videos.bfloat16(), mask)`,
    the loss built with torch ops as the loop builds it (E:131-148), `model.backward(loss)`, `model.step()`.  Held to the reference trajectory:
    loss of step 1 (identical weights) within 1e-3, of steps 2-3 (weights moved by each side's own clip + AdamW) within 3e-3, gradient norms
    within 3 %.  And the native step (IVTrainEngine.train_step: fused loss, same kernels) on a second copy follows the adapter's losses."""
    import json
    from types import SimpleNamespace
    from internvideo_amd import ds_compat, masking
    from internvideo_amd.engine import IVTrainEngine
    from tests.test_step_protocol import FIX, per_step, teacher_features
    cfg = O.named_config(FIX["config"])
    B, TD = FIX["batch"], FIX["td_ratio"]
    T, h, w = cfg.grid
    params = O.synthetic_params(cfg, seed=FIX["param_seed"])
    args = SimpleNamespace(lr=FIX["lr"], weight_decay=FIX["weight_decay"], opt_betas=FIX["betas"], opt_eps=FIX["eps"], clip_grad=FIX["clip"], update_freq=1)
    model, optimizer, _, _ = ds_compat.initialize(args=args, model=build(cfg, params), model_parameters=None, dist_init_required=False)
    native = IVTrainEngine(build(cfg, params), lr=FIX["lr"], betas=tuple(FIX["betas"]), eps=FIX["eps"], weight_decay=FIX["weight_decay"], max_grad_norm=FIX["clip"])
    gv = torch.Generator().manual_seed(FIX["video_seed"])
    loader = [torch.rand(B, 3, T * TD, cfg.img_size, cfg.img_size, generator=gv) for _ in range(FIX["steps"])]
    model.train(); model.zero_grad(); model.micro_steps = 0                               # E:34,44-45
    got, got_native, gn = [], [], []
    for it, st in enumerate(per_step()):
        for group in optimizer.param_groups:                                             # E:56-61
            group["lr"] = FIX["lr_schedule"][it] * group["lr_scale"]
            if group["weight_decay"] > 0:
                group["weight_decay"] = FIX["wd_schedule"][it]
        videos = loader[it].to(DEV)[:, :, ::TD]                                          # E:81
        clip_mid, clip_fin, _, mae = (t.to(DEV) for t in teacher_features(it, cfg, B))
        e = st["model.__call__"]
        mask = torch.from_numpy(np.unpackbits(np.array(e["mask"]["packed"], dtype=np.uint8), axis=1)[:, :e["mask"]["shape"][1]].astype(bool)).to(DEV)
        tg_mid = masking.gather_visible(clip_mid, mask)                                  # E:118-121
        tg_mae = masking.gather_visible(mae, mask, drop_cls=True)                        # E:123-125
        oc, of, om = model(videos.bfloat16(), mask)                                      # E:127-128
        assert [list(o.shape) for o in (oc, of, om)] == [o["shape"] for o in e["outputs"]]
        l_mid = (2 - 2 * (oc * tg_mid).sum(dim=-1)).mean()                               # E:131-148
        l_fin = (2 - 2 * (of * clip_fin).sum(dim=-1)).mean()
        l_mae = (2 - 2 * (om * tg_mae).sum(dim=-1)).mean()
        loss = l_mid * 1.0 + l_fin * 1.0 + l_mae * 1.0
        model.backward(loss); model.step()                                               # E:164-165
        got.append(loss.item()); gn.append(float(model.optimizer._global_grad_norm))
        ln, _ = native.train_step(videos.bfloat16().contiguous(), mask.to(torch.uint8), (tg_mid.to(torch.bfloat16), clip_fin.to(torch.bfloat16), tg_mae.to(torch.bfloat16)),
                                  lr=FIX["lr_schedule"][it], weight_decay=FIX["wd_schedule"][it])
        got_native.append(ln.item())
    want = [s["model.backward"]["loss"] for s in per_step()]
    want_gn = [s["model.step"]["grad_norm"] for s in per_step()]
    rel_l = [abs(a - b) / abs(b) for a, b in zip(got, want)]
    rel_g = [abs(a - b) / abs(b) for a, b in zip(gn, want_gn)]
    rel_n = [abs(a - b) / abs(b) for a, b in zip(got_native, got)]
    print("reference loop replay: loss dev", rel_l, "grad-norm dev", rel_g, "native-vs-adapter loss dev", rel_n)
    os.makedirs(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"), exist_ok=True)
    json.dump(dict(reference_losses=want, hip_losses=got, native_step_losses=got_native, reference_grad_norms=want_gn, hip_grad_norms=gn),
              open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "step_protocol_replay.json"), "w"))
    assert rel_l[0] < 1e-3 and max(rel_l) < 3e-3, rel_l
    assert max(rel_g) < 3e-2, rel_g
    assert max(rel_n) < 2e-3, rel_n
    assert model.micro_steps == FIX["steps"]


def test_reference_step_loop_with_update_freq_2_on_the_hip_path():
    """`--update_freq 2` (run_pretraining.py:42,375; VERDICT r5 missing 3): tests/golden/step_protocol_gas2.json is the reference's own
    train_one_epoch run for four iterations = two optimizer steps around a DeepSpeed-shaped engine with gradient_accumulation_steps 2 (the
    stand-in restates deepspeed==0.10.1's contract; the student, the loop, the masks and the losses are the reference's).  The same call
    sequence on ds_compat.initialize(args with update_freq=2): losses of the four micro-steps and the gradient norms of the two boundaries
    against the reference trajectory (iterations 1-2 run on identical weights: 1e-3; 3-4 after each side's own step: 3e-3; norms 3 %)."""
    import json
    from types import SimpleNamespace
    from internvideo_amd import ds_compat, masking
    from tests.test_step_protocol import FIX2, teacher_features
    cfg = O.named_config(FIX2["config"])
    B, TD = FIX2["batch"], FIX2["td_ratio"]
    T, h, w = cfg.grid
    params = O.synthetic_params(cfg, seed=FIX2["param_seed"])
    args = SimpleNamespace(lr=FIX2["lr"], weight_decay=FIX2["weight_decay"], opt_betas=FIX2["betas"], opt_eps=FIX2["eps"], clip_grad=FIX2["clip"], update_freq=2)
    model, optimizer, _, _ = ds_compat.initialize(args=args, model=build(cfg, params), model_parameters=None, dist_init_required=False)
    assert model.gradient_accumulation_steps() == 2
    gv = torch.Generator().manual_seed(FIX2["video_seed"])
    loader = [torch.rand(B, 3, T * TD, cfg.img_size, cfg.img_size, generator=gv) for _ in range(FIX2["steps"])]
    steps, cur = [], None
    for e in FIX2["trace"]:
        if e["call"] == "clip_teacher":
            cur = {}
            steps.append(cur)
        if cur is not None:
            cur[e["call"]] = e
    model.train(); model.zero_grad(); model.micro_steps = 0
    got, gn = [], []
    for it, st in enumerate(steps):
        for group in optimizer.param_groups:                                             # E:56-61, every iteration
            group["lr"] = FIX2["lr_schedule"][it] * group["lr_scale"]
            if group["weight_decay"] > 0:
                group["weight_decay"] = FIX2["wd_schedule"][it]
        videos = loader[it].to(DEV)[:, :, ::TD]
        clip_mid, clip_fin, _, mae = (t.to(DEV) for t in teacher_features(it, cfg, B))
        e = st["model.__call__"]
        mask = torch.from_numpy(np.unpackbits(np.array(e["mask"]["packed"], dtype=np.uint8), axis=1)[:, :e["mask"]["shape"][1]].astype(bool)).to(DEV)
        tg_mid = masking.gather_visible(clip_mid, mask)
        tg_mae = masking.gather_visible(mae, mask, drop_cls=True)
        oc, of, om = model(videos.bfloat16(), mask)
        loss = (2 - 2 * (oc * tg_mid).sum(dim=-1)).mean() + (2 - 2 * (of * clip_fin).sum(dim=-1)).mean() + (2 - 2 * (om * tg_mae).sum(dim=-1)).mean()
        model.backward(loss); model.step()
        got.append(loss.item())
        if st["model.step"]["boundary"]:
            gn.append(float(model.optimizer._global_grad_norm))
    want = [s["model.backward"]["loss"] for s in steps]
    want_gn = [s["model.step"]["grad_norm"] for s in steps if s["model.step"]["boundary"]]
    rel_l = [abs(a - b) / abs(b) for a, b in zip(got, want)]
    rel_g = [abs(a - b) / abs(b) for a, b in zip(gn, want_gn)]
    print("update_freq 2 replay: loss dev", rel_l, "grad-norm dev", rel_g)
    json.dump(dict(reference_losses=want, hip_losses=got, reference_grad_norms=want_gn, hip_grad_norms=gn),
              open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "step_protocol_gas2_replay.json"), "w"))
    assert max(rel_l[:2]) < 1e-3 and max(rel_l) < 3e-3, rel_l
    assert len(gn) == 2 and max(rel_g) < 3e-2, rel_g
    assert model.micro_steps == 4 and model.global_steps == 2


def test_bf16_parameters_and_tanh_gelu_and_droppath():
    """model.bfloat16() (the DeepSpeed bf16 recipe) goes through the same kernels; gelu='tanh' matches the oracle's tanh
    flavour; DropPath only rescales/zeroes whole-sample branches (rate 1.0 on the last block == that block removed)."""
    cfg = O.named_config("tiny88")
    params = O.synthetic_params(cfg, seed=2)
    video, mask, targets = O.synthetic_batch(cfg, 2, 5, seed=2)
    cfg_t = O.named_config("tiny88"); cfg_t.gelu = "tanh"
    ref = O.student_forward(params, video, mask, cfg_t)
    model = build(cfg, params, fused_mlp_act="tanh")
    out = model(video.to(DEV), torch.from_numpy(mask))
    assert max(rel(o.float(), r) for o, r in zip(out, ref)) < 1e-2
    mb = build(cfg, params).bfloat16()
    ob = mb(video.to(DEV).bfloat16(), torch.from_numpy(mask))
    ref_e = O.student_forward(params, video, mask, cfg)
    assert max(rel(o.float(), r) for o, r in zip(ob, ref_e)) < 2e-2
    tot, _ = losses(ob, targets); tot.backward()
    assert all(p.grad is not None and p.grad.dtype == torch.bfloat16 for p in mb.parameters())
    # eval mode == no drop path
    md = build(cfg, params, drop_path_rate=0.5).eval()
    oe = md(video.to(DEV), torch.from_numpy(mask))
    assert max(rel(o.float(), r) for o, r in zip(oe, ref_e)) < 1e-2
    # training mode: deterministic given the torch seed, finite, and different from eval
    md.train(); torch.manual_seed(0)
    o1 = md(video.to(DEV), torch.from_numpy(mask)); torch.manual_seed(0)
    o2 = md(video.to(DEV), torch.from_numpy(mask))
    assert all(torch.equal(a, b) for a, b in zip(o1, o2)) and all(torch.isfinite(a.float()).all() for a in o1)


def test_forward_rejects_cpu_tensors_and_ragged_masks():
    cfg = O.named_config("tiny64")
    model = build(cfg, O.synthetic_params(cfg, seed=0))
    video, mask, _ = O.synthetic_batch(cfg, 2, 4, seed=0)
    with pytest.raises(M.InternVideoHipError):
        model(video, torch.from_numpy(mask))
    bad = mask.copy(); bad[0, 1] = not bad[0, 1]
    with pytest.raises(RuntimeError):
        model(video.to(DEV), torch.from_numpy(bad))
    with pytest.raises(RuntimeError):
        model(video.to(DEV), torch.from_numpy(bad).to(DEV))


def test_engine_fused_loss_and_main_grads_match_dropin_path():
    """native engine mode (flat buffers, main_grad written by the kernels, fused decoder-tail loss) == drop-in autograd mode."""
    from internvideo_amd.engine import IVTrainEngine
    cfg = O.named_config("tiny88")
    params = O.synthetic_params(cfg, seed=4)
    video, mask, targets = O.synthetic_batch(cfg, 2, 5, seed=4)
    tg = tuple(t.to(DEV) for t in targets)
    m1 = build(cfg, params)
    out = m1(video.to(DEV), torch.from_numpy(mask))
    l1, _ = losses(out, targets)
    l1.backward()
    m2 = build(cfg, params)
    eng = IVTrainEngine(m2, lr=1e-3, max_grad_norm=3.0)
    eng.zero_grad()
    l2, parts = m2.forward_loss(video.to(DEV), torch.from_numpy(mask), tg)
    assert abs(l1.item() - l2.item()) / abs(l1.item()) < 2e-3
    l2.backward()
    for n2, p2 in m2.named_parameters():
        assert p2.grad is None, n2                       # everything went to main_grad
    errs = grad_errors({n: p.main_grad.float() for n, p in m2.named_parameters()}, {n: p.grad for n, p in m1.named_parameters()})
    bad = {k: v for k, v in errs.items() if v > 2e-2}
    assert not bad, bad
    # the same backward with the weight-gradient GEMMs on the engine's side stream: bitwise the same gradients
    snap = {n: p.main_grad.clone() for n, p in m2.named_parameters()}
    eng.zero_grad()
    l2b, _ = m2.forward_loss(video.to(DEV), torch.from_numpy(mask), tg)
    eng.wgrad_stream = torch.cuda.Stream()               # opt in (IVTrainEngine(wgrad_stream=True) creates it at construction)
    eng.backward(l2b)
    eng._finish_reduce()
    torch.cuda.synchronize()
    assert all(torch.equal(p.main_grad, snap[n]) for n, p in m2.named_parameters())
    # optimizer: fused AdamW on the flat buffers == torch.optim.AdamW on the same gradients with the same clip
    named = dict(m2.named_parameters())
    decay = [p for n, p in eng.mat_params]; no_decay = [p for n, p in eng.vec_params]
    ref_params = {n: p.detach().clone().requires_grad_(True) for n, p in named.items()}
    for n, p in ref_params.items():
        p.grad = named[n].main_grad.float().clone()
    total = torch.nn.utils.clip_grad_norm_(list(ref_params.values()), 3.0)
    opt = torch.optim.AdamW([{"params": [ref_params[n] for n, _ in eng.mat_params], "weight_decay": 0.05},
                             {"params": [ref_params[n] for n, _ in eng.vec_params], "weight_decay": 0.0}],
                            lr=1e-3, betas=(0.9, 0.98), eps=1e-6)
    opt.step()
    eng.optimizer_step()
    assert abs(eng.grad_norm.item() - total.item()) / total.item() < 1e-3
    worst = max(rel(named[n].detach(), ref_params[n].detach()) for n in named)
    assert worst < 1e-5, worst
    w = m2.blocks[0].attn.qkv.weight
    assert torch.equal(w._ivh_bf16, w.detach().to(torch.bfloat16))
    # a second full step through train_step runs and changes the loss
    l3, _ = eng.train_step(video.to(DEV), torch.from_numpy(mask), tg)
    assert torch.isfinite(l3).item() and l3.item() != l2.item()


def test_graphed_step_matches_eager_step():
    """engine.capture_step / train_step_graphed (HIP graph replay of forward + loss + backward on both streams, then AdamW) ==
    the eager train_step on the same inputs, bit for bit, and it follows the contents of its static input tensors."""
    from internvideo_amd.engine import IVTrainEngine
    cfg = O.named_config("tiny88")
    params = O.synthetic_params(cfg, seed=6)
    video, mask, targets = O.synthetic_batch(cfg, 2, 5, seed=6)
    vid = video.to(DEV); msk = torch.from_numpy(mask).to(DEV).to(torch.uint8); tg = tuple(t.to(DEV) for t in targets)
    L = 1 + cfg.num_frames * 5
    m1 = build(cfg, params); e1 = IVTrainEngine(m1, lr=1e-3, max_grad_norm=3.0)
    m2 = build(cfg, params); e2 = IVTrainEngine(m2, lr=1e-3, max_grad_norm=3.0)
    vis_inv = M.build_gather_indices(msk, DEV, L=L, check=False)
    l1, _ = e1.train_step(vid, msk, tg, vis_inv=vis_inv)
    sv, sm, st = vid.clone(), msk.clone(), tuple(t.clone() for t in tg)
    e2.capture_step(sv, sm, st, L=L)
    # capture ran the body eagerly twice (warm-up) without an optimizer step: parameters are still the initial ones
    l2, _ = e2.train_step_graphed()
    l2 = l2.clone()                                       # the graph's outputs are static tensors, rewritten by every replay
    torch.cuda.synchronize()
    assert torch.equal(l1, l2)
    assert torch.equal(e1.grad_mat, e2.grad_mat) and torch.equal(e1.grad_vec, e2.grad_vec)
    assert torch.equal(e1.master, e2.master)
    # second step: same data -> both engines continue identically; then new data in the static buffers changes the loss
    l1b, _ = e1.train_step(vid, msk, tg, vis_inv=vis_inv)
    l2b = e2.train_step_graphed()[0].clone()
    assert torch.equal(l1b, l2b) and not torch.equal(l2b, l2)
    sv.copy_(torch.rand_like(sv))
    l2c = e2.train_step_graphed()[0].clone()
    assert torch.isfinite(l2c).item() and not torch.equal(l2c, l2b)


def test_6B_width_bf16_stream_with_interior_taps_matches_oracle():
    """The 6B width (rows of 3200) on the bf16 residual stream WITH interior feature taps: the tap gradients join the stream gradient inside
    the residual norm backward (`dres_extra`), on the two-chunks-per-lane instantiation of that kernel -- the combination in which round 5
    found the gfx950 store-data hazard (profiles/r5_store_data_hazard_gfx950.txt: dbranch rows corrupted).  Every gradient vs the CPU oracle."""
    cfg = O.StudentConfig(img_size=56, embed_dim=3200, depth=3, num_heads=25, mlp_ratio=4.0, num_frames=4, attn_pool_num_heads=16,
                          clip_embed_dim=768, clip_teacher_embed_dim=3200, clip_teacher_final_dim=768, clip_return_layer=3,
                          mae_teacher_embed_dim=1408, mae_return_layer=2)
    from internvideo_amd.hostinfo import usable_cores
    torch.set_num_threads(min(usable_cores(), 32))
    params, video, mask, targets, ref_out, ref_loss, ref_grads = _oracle_run(cfg, 2, 6, 0, True)
    model = build(cfg, params)
    model.residual_dtype = "bf16"
    out = model(video.to(DEV), torch.from_numpy(mask))
    e = [rel(o.float(), r) for o, r in zip(out, ref_out)]
    assert max(e) < 1.5e-2, e
    total, _ = losses(out, targets)
    assert abs(total.item() - ref_loss) / abs(ref_loss) < 1e-3, (total.item(), ref_loss)
    total.backward()
    grads = {k: p.grad for k, p in model.named_parameters()}
    assert all(torch.isfinite(g.float()).all() for g in grads.values() if g is not None)
    errs = grad_errors(grads, ref_grads)
    pool_front = ("clip_projector.norm1_", "clip_projector.cross_attn.q", "clip_projector.cross_attn.k")
    bad = {k: v for k, v in errs.items() if v > (1.2e-1 if k.startswith(pool_front) else 1.5 * grad_tol(k))}
    assert not bad, dict(sorted(bad.items(), key=lambda kv: -kv[1])[:10])


def test_6B_shaped_student_matches_oracle():
    """BASELINE configs[4] geometry in bf16 (the fp8 GEMMs are a later round): width 3200, 25 heads x 128, MLP 12800, attention-pool
    heads 200 wide, depth cut to 2 -- the row kernels at D = 3200, flash attention at hd = 128 (fwd + bwd) and the wide-head pooling
    kernels inside a real student, vs the CPU oracle."""
    cfg = O.StudentConfig(img_size=56, embed_dim=3200, depth=2, num_heads=25, mlp_ratio=4.0, num_frames=4, attn_pool_num_heads=16,
                          clip_embed_dim=768, clip_teacher_embed_dim=3200, clip_teacher_final_dim=768, clip_return_layer=1,
                          mae_teacher_embed_dim=1408, mae_return_layer=1)
    from internvideo_amd.hostinfo import usable_cores
    torch.set_num_threads(min(usable_cores(), 32))
    params, video, mask, targets, ref_out, ref_loss, ref_grads = _oracle_run(cfg, 1, 6, 0, True)
    model = build(cfg, params)
    out = model(video.to(DEV), torch.from_numpy(mask))
    e = [rel(o.float(), r) for o, r in zip(out, ref_out)]
    assert max(e) < 1e-2, e
    total, _ = losses(out, targets)
    assert abs(total.item() - ref_loss) / abs(ref_loss) < 1e-3, (total.item(), ref_loss)
    _check_against_reference_digest("6Bshape", out, total.item(), 1, 6, 1e-2)
    total.backward()
    errs = grad_errors({k: p.grad for k, p in model.named_parameters()}, ref_grads)
    bad = {k: v for k, v in errs.items() if v > grad_tol(k)}
    assert not bad, dict(sorted(bad.items(), key=lambda kv: -kv[1])[:10])


def test_engine_checkpoint_resume_is_bit_exact():
    """SURVEY.md section 5 (checkpoint / resume): IVTrainEngine.state_dict() -> a fresh model + engine -> load_state_dict() continues
    the run exactly (same losses, same master weights) -- flat fp32 master / moments / step are the whole training state."""
    from internvideo_amd.engine import IVTrainEngine
    cfg = O.named_config("tiny88")
    params = O.synthetic_params(cfg, seed=1)
    video, mask, targets = O.synthetic_batch(cfg, 2, 5, seed=1)
    v, m, tg = video.to(DEV), torch.from_numpy(mask), tuple(t.to(DEV) for t in targets)

    def fresh():
        model = build(cfg, params)
        return model, IVTrainEngine(model, lr=1e-3)
    _, eng = fresh()
    for _ in range(2):
        eng.train_step(v, m, tg)
    sd = {k: (t.clone() if torch.is_tensor(t) else t) for k, t in eng.state_dict().items()}
    ref = [eng.train_step(v, m, tg)[0].item() for _ in range(2)]
    model2, eng2 = fresh()
    eng2.load_state_dict(sd)
    got = [eng2.train_step(v, m, tg)[0].item() for _ in range(2)]
    assert got == ref, (got, ref)
    assert torch.equal(eng.master, eng2.master) and eng.step_count == eng2.step_count == 4
    # parameters are views of the flat master buffer: the module's state_dict is current without a copy
    assert torch.equal(model2.state_dict()["blocks.0.attn.qkv.weight"], eng2.master[eng2.mat_off[[n for n, _ in eng2.mat_params].index("blocks.0.attn.qkv.weight")]:][:model2.blocks[0].attn.qkv.weight.numel()].view_as(model2.blocks[0].attn.qkv.weight))


def test_distill_base_config_matches_oracle():
    """BASELINE configs[1] geometry through the distillation class: distill_internvideo2_base_patch14_224 (ViT-B/14, 8 x 224^2,
    clip_return_layer 6, teacher width 1408), global mask keeping 410 of 2048 patches (L = 411), vs the CPU oracle, forward + loss."""
    from internvideo_amd import internvideo2_distill as D
    from internvideo_amd.hostinfo import usable_cores
    torch.set_num_threads(min(usable_cores(), 32))
    cfg = O.StudentConfig(embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, num_frames=8, clip_teacher_embed_dim=1408, clip_return_layer=6,
                          has_mae=False)
    params = O.synthetic_params(cfg, seed=0)
    rng = np.random.Generator(np.random.PCG64(21))
    video = torch.from_numpy(rng.random((1, 3, 8, 224, 224), dtype=np.float32))
    mask = np.ones((1, 2048), dtype=bool)
    mask[0, rng.permutation(2048)[:410]] = False                                 # engine_for_distill.py:89-98 (global N_vis = N - int(N * 0.8))
    mask = np.concatenate([np.zeros((1, 1), dtype=bool), mask], axis=1)
    with torch.no_grad():
        ref = O.encoder_forward(params, video, mask, cfg)
    m = D.distill_internvideo2_base_patch14_224(clip_return_layer=6, clip_teacher_embed_dim=1408, drop_path_rate=0.0)
    m.load_state_dict(params, strict=True)
    m = m.to(DEV).train()
    oc, of = m(video.to(DEV), torch.from_numpy(mask))
    assert tuple(oc.shape) == (6, 1, 411, 1408) and rel(oc.float(), ref["x_clip_align"]) < 1e-2 and rel(of.float(), ref["x_align"]) < 1e-2


# (the 1-rank RCCL test of the multi-rank step modes lives in tests/test_fullsize_gpu.py: it runs in its own subprocess, because tearing the
# RCCL communicator down inside the pytest process aborted -- SIGABRT in destroy_process_group -- in one of ~10 runs)


@pytest.mark.parametrize("n_cp,residual", [(1, "fp32"), (3, "fp32"), (2, "bf16")])
def test_activation_recompute_is_bit_identical_and_saves_memory(n_cp, residual):
    """`use_checkpoint=True, checkpoint_num=n` (P:296,323-327; the 6B recipe's setting): the first n blocks keep only their outputs and
    are recomputed in backward.  DropPath's per-sample scales are an input of the block kernels, so loss and EVERY gradient are
    bit-identical to the run that keeps all activations -- with drop_path on -- and the saved-activation footprint shrinks."""
    cfg = O.named_config("tiny88")
    params = O.synthetic_params(cfg, seed=5)
    video, mask, targets = O.synthetic_batch(cfg, 4, 6, seed=5)
    res = {}
    for tag, kw in (("plain", {}), ("cp", dict(use_checkpoint=True, checkpoint_num=n_cp))):
        model = build(cfg, params, drop_path_rate=0.2, **kw)
        model.residual_dtype = residual
        torch.manual_seed(11)                                   # same DropPath draws in both runs
        import gc
        gc.collect()                                            # earlier tests' cyclic garbage must not be freed inside the measured window
        torch.cuda.synchronize()
        base = torch.cuda.memory_allocated()
        out = model(video.to(DEV), torch.from_numpy(mask))
        held = torch.cuda.memory_allocated() - base             # activations alive between forward and backward
        loss, _ = losses(out, targets)
        loss.backward()
        res[tag] = (loss.detach().clone(), {k: v.grad.detach().clone() for k, v in model.named_parameters() if v.grad is not None}, held)
        del model, out, loss
        gc.collect()
    assert torch.equal(res["plain"][0], res["cp"][0])
    assert res["plain"][1].keys() == res["cp"][1].keys()
    for k, g in res["plain"][1].items():
        assert torch.equal(g, res["cp"][1][k]), k
    assert res["cp"][2] < res["plain"][2]


def _variant_model(cfg, params, **kw):
    m = M.PretrainInternVideo2(
        img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
        mlp_ratio=cfg.mlp_ratio, num_frames=cfg.num_frames, tubelet_size=cfg.tubelet_size, drop_path_rate=0.0,
        attn_pool_num_heads=cfg.attn_pool_num_heads, clip_embed_dim=cfg.clip_embed_dim, clip_teacher_embed_dim=cfg.clip_teacher_embed_dim,
        clip_teacher_final_dim=cfg.clip_teacher_final_dim, clip_return_layer=cfg.clip_return_layer,
        mae_teacher_embed_dim=cfg.mae_teacher_embed_dim, mae_return_layer=cfg.mae_return_layer, **kw)
    sd = m.state_dict()
    m.load_state_dict({k: v for k, v in params.items() if k in sd}, strict=False)
    return m


@pytest.mark.parametrize("engine_mode", [False, True])
def test_sep_pos_embed_matches_reference_golden(engine_mode):
    """`sep_pos_embed=True` (P:479-495, 639-655, 696-712, 726-734; a constructor kwarg of the B1 contract no shipped recipe sets): outputs,
    loss and the gradients of the separable tables against the reference's own CPU run (tests/golden/variants.npz, generator
    make_golden_variants.py).  engine_mode: the same through IVTrainEngine (the tables' gradients reach them through autograd and are
    folded into the engine's buffers)."""
    g = np.load(os.path.join(GOLD, "variants.npz"))
    B, n_vis, seed = (int(v) for v in g["meta"])
    cfg = O.named_config("tiny64")
    params = dict(O.synthetic_params(cfg, seed=seed))
    sep = ["pos_embed_spatial", "pos_embed_temporal", "pos_embed_cls", "clip_pos_embed_spatial", "clip_pos_embed_temporal", "clip_pos_embed_cls",
           "mae_pos_embed_spatial", "mae_pos_embed_temporal"]
    for k in sep:
        params[k] = torch.from_numpy(g["sep:in:" + k])
    video, mask, targets = O.synthetic_batch(cfg, B, n_vis, seed=seed)
    model = _variant_model(cfg, params, sep_pos_embed=True)
    assert set(sep) <= set(model.state_dict()) and "pos_embed" not in model.state_dict() and set(sep) <= model.no_weight_decay()
    model = model.to(DEV).train()
    if engine_mode:
        from internvideo_amd.engine import IVTrainEngine
        eng = IVTrainEngine(model, lr=0.0, weight_decay=0.0, max_grad_norm=0.0)
        tg = tuple(t.to(DEV) for t in targets)
        eng.zero_grad()
        loss, _ = model.forward_loss(video.to(DEV), torch.from_numpy(mask), tg)
        eng.backward(loss)
        grads = {k: p.main_grad.float() for k, p in model.named_parameters()}
    else:
        out = model(video.to(DEV), torch.from_numpy(mask))
        e = [rel(out[0].float(), g["sep:x_clip_align"]), rel(out[1].float(), g["sep:x_align"]), rel(out[2].float(), g["sep:x_mae_align"])]
        assert max(e) < 1e-2, e
        loss, _ = losses(out, targets)
        loss.backward()
        grads = {k: p.grad for k, p in model.named_parameters()}
    assert abs(loss.item() - g["sep:loss"][0]) < 1e-3 * g["sep:loss"][0]
    worst = {k[9:]: rel(grads[k[9:]], g[k]) for k in g.files if k.startswith("sep:grad:")}
    assert set(sep) <= set(worst)
    bad = {k: v for k, v in worst.items() if v > (8e-2 if k.startswith("clip_projector") else 4e-2)}
    assert not bad, bad


def test_norm_type_none_decoders_match_reference_golden():
    """`clip_norm_type = mae_norm_type = 'none'` (P:358-363, 396-401): the decoders return the LayerNorm output without the l2 step; drop-in
    forward and the fused-loss path against the reference's own CPU run."""
    g = np.load(os.path.join(GOLD, "variants.npz"))
    B, n_vis, seed = (int(v) for v in g["meta"])
    cfg = O.named_config("tiny64")
    params = O.synthetic_params(cfg, seed=seed)
    video, mask, targets = O.synthetic_batch(cfg, B, n_vis, seed=seed)
    model = _variant_model(cfg, params, clip_norm_type="none", mae_norm_type="none").to(DEV).train()
    out = model(video.to(DEV), torch.from_numpy(mask))
    e = [rel(out[0].float(), g["none:x_clip_align"]), rel(out[1].float(), g["none:x_align"]), rel(out[2].float(), g["none:x_mae_align"])]
    assert max(e) < 1e-2, e
    assert 5.0 < out[0].float().norm(dim=-1).mean().item() < 20.0          # not l2-normalised: ~sqrt(C) rows
    loss, _ = losses(out, targets)
    assert abs(loss.item() - g["none:loss"][0]) < 2e-3 * abs(g["none:loss"][0])
    loss.backward()
    got = {k: p.grad.clone() for k, p in model.named_parameters()}
    worst = {k[10:]: rel(got[k[10:]], g[k]) for k in g.files if k.startswith("none:grad:")}
    bad = {k: v for k, v in worst.items() if v > 4e-2}
    assert not bad, bad
    # the fused-loss path (what the engine runs) gives the same loss and gradients
    model.zero_grad(set_to_none=True)
    l2, _ = model.forward_loss(video.to(DEV), torch.from_numpy(mask), tuple(t.to(DEV) for t in targets))
    assert abs(l2.item() - g["none:loss"][0]) < 2e-3 * abs(g["none:loss"][0])
    l2.backward()
    for k in ("clip_decoder.0.norm.weight", "mae_decoder.0.norm.bias", "blocks.0.norm1.weight", "final_clip_decoder.norm.weight"):
        assert rel(model.get_parameter(k).grad, got[k]) < 2e-2, k
    with pytest.raises(NotImplementedError):
        M.Linear_Decoder(norm_type="l1")


@pytest.mark.parametrize("residual", ["fp32", "bf16"])
def test_drop_path_matches_the_reference_on_its_own_draws(residual):
    """DropPath > 0 in train mode (what the recipes and bench.py run: 0.25 / 0.3) against the REFERENCE's run on the same draws: the fixture
    (tests/golden/variants.npz `dp:*`, make_golden_variants.py) holds the uniform numbers timm's DropPath drew per (block, branch, sample) and
    the reference's outputs / loss / gradients; fed the same numbers (`model._dp_uniform`), the per-sample keep / (1 - p) factors that the
    residual kernels apply give the same result -- six of the sixteen (block, branch, sample) branches are dropped in this fixture."""
    g = np.load(os.path.join(GOLD, "variants.npz"))
    B, n_vis, seed = (int(v) for v in g["dp:meta"])
    cfg = O.named_config("tiny64")
    params = O.synthetic_params(cfg, seed=int(g["meta"][2]))
    video, mask, targets = O.synthetic_batch(cfg, B, n_vis, seed=seed)
    model = build(cfg, params, drop_path_rate=float(g["dp:rate"][0]))
    model.residual_dtype = residual
    U = torch.from_numpy(g["dp:uniform"])
    assert tuple(U.shape) == (cfg.depth, 2, B)
    model._dp_uniform = U
    keep = 1.0 - torch.tensor(model.drop_path_rates).view(-1, 1, 1)
    assert int((torch.floor(keep + U) == 0).sum()) >= 4                     # the fixture really drops branches
    out = model(video.to(DEV), torch.from_numpy(mask))
    e = [rel(out[0].float(), g["dp:x_clip_align"]), rel(out[1].float(), g["dp:x_align"]), rel(out[2].float(), g["dp:x_mae_align"])]
    assert max(e) < (2e-2 if residual == "bf16" else 1e-2), e
    loss, _ = losses(out, targets)
    assert abs(loss.item() - g["dp:loss"][0]) < 1e-3 * g["dp:loss"][0], (loss.item(), g["dp:loss"][0])
    loss.backward()
    got = {k: p.grad for k, p in model.named_parameters()}
    worst = {k[8:]: rel(got[k[8:]], g[k]) for k in g.files if k.startswith("dp:grad:")}
    bad = {k: v for k, v in worst.items() if v > (6e-2 if residual == "bf16" else 4e-2)}
    assert not bad, bad
    # without the override the draws are fresh: another mask pattern, another loss
    model._dp_uniform = None
    torch.manual_seed(5)
    out2 = model(video.to(DEV), torch.from_numpy(mask))
    assert rel(out2[0].float(), g["dp:x_clip_align"]) > 1e-3
