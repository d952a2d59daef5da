"""Full-size parity on a real MI355X ("next round" item 1): the configurations BASELINE.json names, at their own
geometry, with gradients -- not only the fixture-sized models of test_model_gpu.py / test_flavours_gpu.py.

  * configs[2]: InternVideo2-1B stage-1 student, 8 x 224^2, mask 0.8 -> L = 417, B = 2: outputs, loss and EVERY parameter gradient
    vs the CPU oracle (fp32); and one step at the bench batch (B = 128) replayed from the HIP graph == the same step issued eagerly.
  * configs[3]: the stage-2 vision encoder `pretrain_internvideo2_1b_patch14_224(config)` built from the reference's stage-2 config
    values (multi_modality/scripts/pretraining/stage2/1B/config.py:19-21,43-74: 4 frames, random mask 0.8 -> L = 206, batch 64) +
    `Stage2VisionTextHeads` VTC loss with synthetic text [CLS] features.  The CPU oracle cannot run 64 clips of a 1B tower in test
    time: the tower is held to the oracle on the first ORACLE_CLIPS clips (outputs + pooled feature), the contrastive loss to the
    oracle's `vtc_loss` on all 64 projected features, and the backward to finiteness + the n = 64 kernel parity of test_kernels_gpu.
  * configs[1]: distill_internvideo2_base_patch14_224 (B/14, L = 411) forward AND backward vs the oracle.
Diagnostics (worst gradient errors per test) are written to gpurun_out/parity_fullsize.json when that directory exists.
Tolerances as everywhere: outputs rel-L2 <= 1e-2, loss <= 1e-3 relative, gradients rel-L2 <= 3e-2 (6e-2 in front of
the 1-query attention pool, where the reference's own bf16 run is 2 % off its fp32 run: tests/test_model_gpu.py::grad_tol)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from internvideo_amd import internvideo2_pretrain as M  # noqa: E402
from oracle import internvideo2_oracle as O  # noqa: E402
from tests.test_model_gpu import build, grad_errors, grad_tol, losses, rel, _oracle_run  # noqa: E402

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_CLIPS = 2


@pytest.fixture(autouse=True)
def _release_graph_pools():
    """the B = 128 tests capture HIP graphs whose private pools hold ~120 GB each; engine <-> model <-> hook closures are reference cycles, so the
    pools outlive the test function until the collector runs -- and the next full-size model (6B: 24 GB of weights) finds the device full"""
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    yield
    gc.collect()
    torch.cuda.empty_cache()


def _note(key, value):
    """measured errors of a run, kept only when the caller asks for them (IVH_PARITY_NOTES=<file>): the tests themselves write nothing
    into the repository"""
    path = os.environ.get("IVH_PARITY_NOTES")
    if not path:
        return
    data = {}
    if os.path.isfile(path):
        try:
            data = json.load(open(path))
        except Exception:
            data = {}
    data[key] = value
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)


def _threads():
    from internvideo_amd.hostinfo import usable_cores
    torch.set_num_threads(min(usable_cores(), 32))


_ORACLE_1B = {}


def _oracle_1B_B2():
    """one fp32 oracle run (forward + backward over 1.07 G parameters, ~40 s of host time) shared by both residual settings"""
    if not _ORACLE_1B:
        _threads()
        cfg = O.named_config("1B")
        _ORACLE_1B["run"] = (cfg,) + tuple(_oracle_run(cfg, 2, 52, 0, True))
    return _ORACLE_1B["run"]


# Stated bounds of the two residual settings at the 1B model's own size (B = 2, L = 417, every one of the 1.07 G gradients).
#   fp32 stream: SURVEY 8(c) bars (outputs 1e-2, gradients 3e-2, 8e-2 in front of the 1-query pool).
#   bf16 stream (what bench.py times; the reference's own recipe: DropoutAddRMSNorm with residual_in_fp32 False, P:283-286, 467): the
#   stream is rounded to 8 mantissa bits once per block, 80 roundings on the way to the last tap.  Head outputs 2e-2 (the reference's own
#   bf16 run is 0.4-0.7e-2 off its fp32 run at depth 12; rounding noise of independent roundings grows ~sqrt(depth)), block
#   gradients 4.5e-2 = 1.5 x the fp32 floor, 1.2e-1 in front of the pool.  The loss bar stays 1e-3 for both.
_BOUNDS = {"fp32": dict(out=1e-2, grad=1.0, pool=8e-2), "bf16": dict(out=2e-2, grad=1.5, pool=1.2e-1)}


@pytest.mark.parametrize("residual", ["fp32", "bf16"])
def test_1B_student_gradients_match_oracle_at_full_size(residual):
    """BASELINE configs[2] geometry, B = 2, every parameter gradient (1.07 G values) against the fp32 oracle, for the fp32 residual stream
    (parity default) AND the bf16 stream the headline bench runs"""
    cfg, params, video, mask, targets, ref_out, ref_loss, ref_grads = _oracle_1B_B2()
    model = build(cfg, params)
    model.residual_dtype = residual
    bd = _BOUNDS[residual]
    out = model(video.to(DEV), torch.from_numpy(mask))
    assert tuple(out[0].shape) == (6, 2, 417, 3200) and tuple(out[2].shape) == (4, 2, 416, 1408)
    e = [rel(o.float(), r) for o, r in zip(out, ref_out)]
    per_tap = [rel(out[0][i].float(), ref_out[0][i]) for i in range(out[0].shape[0])]      # clip decoder k reads block 34 + k's stream
    total, _ = losses(out, targets)
    loss_err = abs(total.item() - ref_loss) / abs(ref_loss)
    total.backward()
    errs = grad_errors({k: p.grad for k, p in model.named_parameters()}, ref_grads)
    worst = dict(sorted(errs.items(), key=lambda kv: -kv[1])[:12])
    by_block = {}
    for k, v in errs.items():
        if k.startswith("blocks."):
            i = int(k.split(".")[1])
            by_block[i] = max(by_block.get(i, 0.0), v)
    _note(f"1B_B2_L417_{residual}", dict(output_rel=e, per_tap_output_rel=per_tap, loss_rel=loss_err, worst_grad_rel=worst,
                                         worst_per_block=[by_block[i] for i in sorted(by_block)]))
    del model
    torch.cuda.empty_cache()
    assert max(e) < bd["out"], e
    assert loss_err < 1e-3, (total.item(), ref_loss)
    # In front of the 1-query attention pool the bound is 8e-2 at this depth (6e-2 on the fixture-sized models): these gradients are a
    # softmax Jacobian of ONE mean query over all 417 tokens of 2 clips, fed by the 40-block stack's output (itself 0.5 % off the fp32
    # oracle); across kernel revisions that leave every other number unchanged they move between 5.3 % and 6.4 %.
    pool_front = ("clip_projector.norm1_", "clip_projector.cross_attn.q", "clip_projector.cross_attn.k")
    bad = {k: v for k, v in errs.items() if v > (bd["pool"] if k.startswith(pool_front) else bd["grad"] * grad_tol(k))}
    assert not bad, dict(sorted(bad.items(), key=lambda kv: -kv[1])[:10])


@pytest.mark.parametrize("residual", ["fp32", "bf16"])
def test_1B_graph_replayed_step_equals_eager_step_at_the_bench_batch(residual):
    """the N = 1 bench path (HIP-graph replay of forward + loss + backward, eager AdamW) against the eager step at the bench batch:
    same loss, same gradient norm, same updated weights -- the grouped weight-gradient launches and the multi-round persistent GEMMs
    are exercised at the shapes the headline number is measured on"""
    from internvideo_amd.engine import IVTrainEngine
    B = int(os.environ.get("IV_FULLSIZE_BATCH", "128"))
    L, n_vis = 417, 52
    torch.manual_seed(0)
    model = M.pretrain_internvideo2_1B_patch14_224(clip_return_layer=6, mae_return_layer=4, drop_path_rate=0.0, num_frames=8).to(DEV).train()
    model.residual_dtype = residual                      # "bf16" = bench.py's default (--residual bf16)
    eng = IVTrainEngine(model, lr=1e-4, max_grad_norm=3.0)
    g = torch.Generator(device="cpu").manual_seed(0)
    video = torch.rand((B, 3, 8, 224, 224), generator=g).to(DEV).to(torch.bfloat16)
    mask = torch.ones((B, 8, 256), dtype=torch.bool)
    for b in range(B):
        for t in range(8):
            mask[b, t, torch.randperm(256, generator=g)[:n_vis]] = False
    mask = torch.cat([torch.zeros((B, 1), dtype=torch.bool), mask.reshape(B, -1)], dim=1).to(DEV).to(torch.uint8)
    unit = lambda *s: torch.nn.functional.normalize(torch.randn(*s, device=DEV), dim=-1).to(torch.bfloat16)   # noqa: E731
    tg = (unit(6, B, L, 3200), unit(B, 768), unit(4, B, L - 1, 1408))
    start = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in eng.state_dict().items()}
    loss_e, _ = eng.train_step(video, mask, tg)
    loss_e = loss_e.clone(); gn_e = eng.grad_norm.clone(); master_e = eng.master.clone()
    eng.load_state_dict(start)
    del start
    torch.cuda.empty_cache()
    eng.capture_step(video, mask, tg, L=L)
    loss_g = eng.train_step_graphed()[0].clone()
    torch.cuda.synchronize()
    _note(f"1B_graph_vs_eager_{residual}", dict(B=B, loss_eager=loss_e.item(), loss_graph=loss_g.item(), grad_norm_eager=gn_e.item(), grad_norm_graph=eng.grad_norm.item(),
                                    peak_mem_gb=torch.cuda.max_memory_allocated() / 2 ** 30))
    assert torch.isfinite(loss_e).item() and torch.equal(loss_e, loss_g), (loss_e.item(), loss_g.item())
    assert torch.equal(gn_e, eng.grad_norm), (gn_e.item(), eng.grad_norm.item())
    assert torch.equal(master_e, eng.master)


def test_6B_encoder_at_full_depth_matches_the_reference_digest():
    """BASELINE configs[4]'s encoder at its real depth and width (48 blocks x 3200, 25 heads of 128, mlp_ratio 4, six CLIP taps, four MAE taps)
    against the REFERENCE's own fp32 CPU forward of the same 5.9 G weights (tests/golden/student_6B_fulldepth_digest.npz,
    make_golden_6b_fulldepth.py; VERDICT r3 next 8: beyond the depth-2 digest).  Weights are streamed tensor by tensor from the oracle's
    generator into a model built on the device (no 24 GB host dictionary); 4 frames of 224^2, 52 visible patches per frame (L = 209), B = 1,
    bf16 compute with the fp32 residual stream.  Bars: head outputs 1e-2 rel-L2 on the stored rows and projections, loss 1e-3."""
    from tests.test_model_gpu import _check_against_reference_digest
    g = np.load(os.path.join(ROOT, "tests", "golden", "student_6B_fulldepth_digest.npz"))
    cfg = O.StudentConfig(embed_dim=3200, depth=48, num_heads=25, mlp_ratio=4.0, num_frames=4, attn_pool_num_heads=16, clip_embed_dim=768,
                          clip_teacher_embed_dim=3200, clip_teacher_final_dim=768, clip_return_layer=6, mae_teacher_embed_dim=1408, mae_return_layer=4)
    with torch.device(DEV):
        model = M.pretrain_internvideo2_6B_patch14_224(num_frames=4, drop_path_rate=0.0, clip_return_layer=6, mae_return_layer=4)
    sd = dict(model.named_parameters())
    n = 0
    with torch.no_grad():
        for k, t in O.iter_synthetic_params(cfg, seed=0, gamma=float(g["gamma"][0])):
            sd[k].copy_(t.reshape(sd[k].shape))
            n += t.numel()
    assert n == sum(p.numel() for p in model.parameters())
    model.train()
    video, mask, targets = O.synthetic_batch(cfg, 1, 52, seed=0)
    with torch.no_grad():
        out = model(video.to(DEV), torch.from_numpy(mask))
    assert tuple(out[0].shape) == (6, 1, 209, 3200) and tuple(out[2].shape) == (4, 1, 208, 1408)
    total, _ = losses(out, targets)
    _note("6B_fulldepth_L209", dict(loss=total.item(), loss_reference=float(g["losses"][0])))
    _check_against_reference_digest("6B_fulldepth", out, total.item(), 1, 52, 1e-2)


def test_6B_encoder_at_its_own_shape_bf16_and_fp8_match_the_reference_digest():
    """BASELINE configs[4] at ITS OWN shape: the 6B encoder, 48 blocks x 3200, on a 16 x 224^2 clip with 52 visible patches
    per frame -> L = 833 (single_modality/scripts/pretraining/6B_pt.sh:47-50), against the REFERENCE's own fp32 CPU forward of the same 5.9 G
    weights at that length (tests/golden/student_6B_fulldepth_16f_digest.npz, make_golden_6b_fulldepth.py --frames 16).  Both arithmetic paths
    of the bench line are held to the ORACLE-side digest, not to each other: bf16 GEMMs (outputs 1e-2 rel-L2, loss 1e-3; measured 5.0e-3 /
    6.4e-5) and the e4m3 block GEMMs of `bench.py --model 6B --fp8` (per-tensor current scaling): the LOSS meets north_star's 1e-3 as well
    (measured 1.0e-4), the head OUTPUTS do not meet the bf16 bar -- a 3-mantissa-bit format through 192 chained GEMMs at LayerScale 0.3
    measures 3.3-5.3e-2 rel-L2 on the stored rows / projections; the stated bar is 7e-2, and the bench line's `dtype` note says so."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "student_6B_fulldepth_16f_digest.npz"))
    assert int(g["frames"][0]) == 16
    cfg = O.StudentConfig(embed_dim=3200, depth=48, num_heads=25, mlp_ratio=4.0, num_frames=16, attn_pool_num_heads=16, clip_embed_dim=768,
                          clip_teacher_embed_dim=3200, clip_teacher_final_dim=768, clip_return_layer=6, mae_teacher_embed_dim=1408, mae_return_layer=4)
    with torch.device(DEV):
        model = M.pretrain_internvideo2_6B_patch14_224(num_frames=16, drop_path_rate=0.0, clip_return_layer=6, mae_return_layer=4)
    sd = dict(model.named_parameters())
    n = 0
    with torch.no_grad():
        for k, t in O.iter_synthetic_params(cfg, seed=0, gamma=float(g["gamma"][0])):
            sd[k].copy_(t.reshape(sd[k].shape))
            n += t.numel()
    assert n == sum(p.numel() for p in model.parameters())
    model.train()
    video, mask, targets = O.synthetic_batch(cfg, 1, 52, seed=0)
    ref_loss = float(g["losses"][0])

    def digest_errors(out):
        e = {}
        for key, o in zip(("x_clip_align", "x_align", "x_mae_align"), out):
            rows = o.detach().float().cpu().double().numpy().reshape(-1, o.shape[-1])
            C = rows.shape[1]
            proj = np.random.Generator(np.random.PCG64(777 + C)).standard_normal((C, 16)).astype(np.float32) / np.sqrt(C).astype(np.float32)
            e[key] = (float(np.linalg.norm(rows[:3] - g[key + ":rows"]) / np.linalg.norm(g[key + ":rows"])),
                      float(np.linalg.norm(rows @ proj.astype(np.float64) - g[key + ":proj"]) / np.linalg.norm(g[key + ":proj"])))
        return e

    res = {}
    for tag, fp8 in (("bf16", False), ("fp8_e4m3", True)):
        model.fp8_gemm, model.fp8_scaling, model.fp8_weight_scales = fp8, "current", "tensor"
        with torch.no_grad():
            out = model(video.to(DEV), torch.from_numpy(mask))
        assert tuple(out[0].shape) == (6, 1, 833, 3200) and tuple(out[2].shape) == (4, 1, 832, 1408)
        total, _ = losses(out, targets)
        res[tag] = dict(loss=total.item(), loss_rel=abs(total.item() - ref_loss) / ref_loss, out_rel=digest_errors(out))
        del out
    model.fp8_gemm = False
    _note("6B_fulldepth_L833_own_shape", dict(loss_reference=ref_loss, **res))
    print("6B at 16 x 224^2 (L = 833) vs the reference's fp32 CPU forward:", json.dumps(res))
    assert max(max(v) for v in res["bf16"]["out_rel"].values()) < 1e-2 and res["bf16"]["loss_rel"] < 1e-3, res["bf16"]
    assert max(max(v) for v in res["fp8_e4m3"]["out_rel"].values()) < 7e-2 and res["fp8_e4m3"]["loss_rel"] < 1e-3, res["fp8_e4m3"]


def _stage2_config():
    # multi_modality/scripts/pretraining/stage2/1B/config.py:43-74 (vision_encoder block; pretrained checkpoint not available offline)
    return dict(vision_encoder=dict(
        name="pretrain_internvideo2_1b_patch14_224", img_size=224, num_frames=4, tubelet_size=1, patch_size=14, d_model=1408, clip_embed_dim=768,
        clip_teacher_embed_dim=3200, clip_teacher_final_dim=768, clip_norm_type='l2', clip_return_layer=6, clip_student_return_interval=1,
        pretrained=None, use_checkpoint=False, checkpoint_num=40, use_flash_attn=True, use_fused_rmsnorm=True, use_fused_mlp=True,
        sep_image_video_pos_embed=True))


def test_stage2_1B_vision_tower_and_vtc_loss_at_config_size():
    """BASELINE configs[3]: L = 206 (4 x 224^2, random mask 0.8 of 1024 patches), per-GPU batch 64, temp 0.07, idx = arange"""
    _threads()
    from internvideo_amd import mm_internvideo2 as V
    from internvideo_amd.stage2 import Stage2VisionTextHeads
    B, n_keep = 64, 1024 - int(1024 * 0.8)
    cfg = O.StudentConfig(embed_dim=1408, depth=40, num_heads=16, mlp_ratio=48 / 11, num_frames=4, clip_return_layer=6, has_mae=False,
                          sep_image_video_pos_embed=True)
    params = O.synthetic_params(cfg, seed=3)
    model = V.pretrain_internvideo2_1b_patch14_224(_stage2_config())
    model.load_state_dict(params, strict=True)
    model.drop_path_rates = [0.0] * len(model.drop_path_rates)              # parity runs without stochastic depth
    model = model.to(DEV).train()
    rng = np.random.Generator(np.random.PCG64(33))
    video = torch.from_numpy(rng.random((B, 3, 4, 224, 224), dtype=np.float32))
    mask = np.ones((B, 1024), dtype=bool)
    for b in range(B):
        mask[b, rng.permutation(1024)[:n_keep]] = False                        # multi_modality/models/mask.py:24-37 (random masking), cls kept
    mask = np.concatenate([np.zeros((B, 1), dtype=bool), mask], axis=1)
    assert int((~mask[0]).sum()) == 206
    x_vis, x_pool, x_clip, x_align = model(video.to(DEV).to(torch.bfloat16), torch.from_numpy(mask), False)
    assert tuple(x_vis.shape) == (B, 206, 1408) and tuple(x_pool.shape) == (B, 768) and tuple(x_clip.shape) == (6, B, 206, 3200)
    with torch.no_grad():
        ref = O.encoder_forward(params, video[:ORACLE_CLIPS].to(torch.bfloat16).float(), mask[:ORACLE_CLIPS], cfg)
    e = dict(x_vis=rel(x_vis[:ORACLE_CLIPS].float(), ref["x_vis"]), x_pool_vis=rel(x_pool[:ORACLE_CLIPS].float(), ref["x_pool_vis"]),
             x_clip_align=rel(x_clip[:, :ORACLE_CLIPS].float(), ref["x_clip_align"]), x_align=rel(x_align[:ORACLE_CLIPS].float(), ref["x_align"]))
    heads = Stage2VisionTextHeads(vision_width=768, text_width=1024, embed_dim=512, temp=0.07, loss_weight=dict(vtc=1.0, uta=0.0)).to(DEV)
    text_cls = torch.from_numpy(rng.standard_normal((B, 1024)).astype(np.float32)).to(DEV)
    idx = torch.arange(B, device=DEV)
    out = heads(x_pool, text_cls.to(torch.bfloat16), idx)
    W = {k: v.detach().cpu().float() for k, v in heads.named_parameters()}
    v_ref = x_pool.detach().float().cpu() @ W["vision_proj.weight"].t() + W["vision_proj.bias"]
    t_ref = text_cls.to(torch.bfloat16).float().cpu() @ W["text_proj.weight"].t() + W["text_proj.bias"]
    want = O.vtc_loss(v_ref, t_ref, idx.cpu(), O.clamp_temperature(W["temp"])).item()
    loss_err = abs(out["loss_vtc"].item() - want) / abs(want)
    out["loss_vtc"].backward()
    named = dict(model.named_parameters())
    finite = all(torch.isfinite(p.grad).all().item() for p in named.values() if p.grad is not None)
    trunk = all(p.grad is not None for n, p in named.items() if n.startswith(("blocks.", "patch_embed.", "clip_projector.")))
    qkv0 = model.blocks[0].attn.qkv.weight.grad
    _note("stage2_1B_L206_B64", dict(tower_rel=e, vtc_loss=out["loss_vtc"].item(), vtc_loss_oracle=want, vtc_rel=loss_err,
                                     grad_norm_block0_qkv=float(qkv0.double().norm()), grad_norm_vision_proj=float(heads.vision_proj.weight.grad.double().norm())))
    assert max(e.values()) < 1e-2, e
    # ... and against the REFERENCE's own encoder at this size (tests/golden/stage2_vision_1B_digest.npz, make_golden_stage2_fullsize.py):
    # the same bar on the digest (first rows + 16 random projections of every row) of its outputs for the first ORACLE_CLIPS clips
    gd = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stage2_vision_1B_digest.npz"))
    assert int(gd["meta"][1]) == ORACLE_CLIPS
    for key, tt in (("x_vis", x_vis[:ORACLE_CLIPS]), ("x_pool_vis", x_pool[:ORACLE_CLIPS]), ("x_clip_align", x_clip[:, :ORACLE_CLIPS]), ("x_align", x_align[:ORACLE_CLIPS])):
        rows = tt.detach().float().cpu().double().numpy().reshape(-1, tt.shape[-1])
        C = rows.shape[1]
        proj = np.random.Generator(np.random.PCG64(777 + C)).standard_normal((C, 16)).astype(np.float32) / np.sqrt(C).astype(np.float32)
        e_rows = np.linalg.norm(rows[:3] - gd[key + ":rows"]) / np.linalg.norm(gd[key + ":rows"])
        e_proj = np.linalg.norm(rows @ proj.astype(np.float64) - gd[key + ":proj"]) / np.linalg.norm(gd[key + ":proj"])
        assert e_rows < 1e-2 and e_proj < 1e-2, (key, e_rows, e_proj)
    assert loss_err < 2e-3, (out["loss_vtc"].item(), want)                    # bf16 projections of 64 x 768 features
    assert finite and trunk and float(qkv0.double().norm()) > 0.0


def test_distill_B14_forward_and_backward_match_oracle():
    """BASELINE configs[1] through the distillation class, forward + loss + every parameter gradient"""
    _threads()
    from internvideo_amd import internvideo2_distill as D
    cfg = O.StudentConfig(embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, num_frames=8, clip_teacher_embed_dim=1408, clip_return_layer=6,
                          has_mae=False)
    params = O.synthetic_params(cfg, seed=0)
    rng = np.random.Generator(np.random.PCG64(21))
    video = torch.from_numpy(rng.random((1, 3, 8, 224, 224), dtype=np.float32))
    mask = np.ones((1, 2048), dtype=bool)
    mask[0, rng.permutation(2048)[:410]] = False                                 # engine_for_distill.py:89-98 (global N_vis = N - int(N * 0.8))
    mask = np.concatenate([np.zeros((1, 1), dtype=bool), mask], axis=1)
    unit = lambda shape: torch.nn.functional.normalize(torch.from_numpy(rng.standard_normal(shape).astype(np.float32)), dim=-1)   # noqa: E731
    tc, tf = unit((6, 1, 411, 1408)), unit((1, 768))
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.encoder_forward(p, video, mask, cfg)
    ref_loss = (2 - 2 * (ref["x_clip_align"] * tc).sum(-1)).mean() + (2 - 2 * (ref["x_align"] * tf).sum(-1)).mean()   # engine_for_distill.py:107-121
    ref_loss.backward()
    m = D.distill_internvideo2_base_patch14_224(clip_return_layer=6, clip_teacher_embed_dim=1408, drop_path_rate=0.0)
    m.load_state_dict(params, strict=True)
    m = m.to(DEV).train()
    loss, (lm, lf) = m.forward_loss(video.to(DEV), torch.from_numpy(mask), (tc.to(DEV), tf.to(DEV)))
    assert abs(loss.item() - ref_loss.item()) / abs(ref_loss.item()) < 1e-3, (loss.item(), ref_loss.item())
    # ... and the loss / gradient norms against the REFERENCE's own DistInternVideo2 at this size (tests/golden/distill_B14_digest.npz,
    # make_golden_distill_b14.py; the oracle is held to the same digest on the CPU)
    gd = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "distill_B14_digest.npz"))
    assert abs(loss.item() - float(gd["loss"][0])) / float(gd["loss"][0]) < 1e-3
    loss.backward()
    named_ = dict(m.named_parameters())
    for key in gd.files:
        if key.startswith("grad:") and key.endswith(":norm"):
            k_ = key[5:-5]
            want_ = float(gd[key][0])
            assert abs(named_[k_].grad.double().norm().item() - want_) < 3e-2 * want_, k_
    errs = grad_errors({k: q.grad for k, q in m.named_parameters()}, {k: v.grad for k, v in p.items()})
    _note("distill_B14_L411", dict(loss=loss.item(), loss_oracle=ref_loss.item(), worst_grad_rel=dict(sorted(errs.items(), key=lambda kv: -kv[1])[:8])))
    bad = {k: v for k, v in errs.items() if v > grad_tol(k)}
    assert not bad, dict(sorted(bad.items(), key=lambda kv: -kv[1])[:10])


def test_final_feature_not_distilled_is_a_zero_loss_term():
    """clip_teacher_final_dim = 0 / clip_loss_ratio[1] = 0 (engine_for_pretraining.py:135-138) must train, not raise"""
    from internvideo_amd.engine import IVTrainEngine
    cfg = O.named_config("tiny88")
    params = O.synthetic_params(cfg, seed=5)
    video, mask, targets = O.synthetic_batch(cfg, 2, 5, seed=5)
    tg = tuple(t.to(DEV) for t in targets)
    model = build(cfg, params)
    full, (lc, lf, lmae) = model.forward_loss(video.to(DEV), torch.from_numpy(mask), tg)
    eng = IVTrainEngine(build(cfg, params), lr=1e-3, clip_loss_ratio=(1.0, 0.0))
    l0, parts = eng.train_step(video.to(DEV), torch.from_numpy(mask), tg)
    assert parts[1].item() == 0.0 and abs(l0.item() - (lc.item() + lmae.item())) < 1e-4 * abs(l0.item())
    l1, parts = model.forward_loss(video.to(DEV), torch.from_numpy(mask), (tg[0], None, tg[2]))
    assert parts[1].item() == 0.0 and torch.isfinite(l1).item()


_RCCL_MODES = r"""
import faulthandler, json, os, sys, socket, torch, torch.distributed as dist
faulthandler.enable()
sys.path.insert(0, {root!r})
from internvideo_amd.engine import IVTrainEngine
from oracle import internvideo2_oracle as O
from tests.test_model_gpu import build
mode = sys.argv[1]
DEV = "cuda"
cfg = O.named_config("tiny88")
params = O.synthetic_params(cfg, seed=1)
video, mask, targets = O.synthetic_batch(cfg, 2, 5, seed=1)
v, m, tg = video.to(DEV), torch.from_numpy(mask).to(DEV).to(torch.uint8), tuple(t.to(DEV) for t in targets)
L = int((~torch.from_numpy(mask)[0]).sum())
base = IVTrainEngine(build(cfg, params), lr=1e-3)
ref = [base.train_step(v, m, tg)[0].item() for _ in range(3)]
print("STEP ref done", flush=True)
with socket.socket() as sock:
    sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{{port}}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
print("STEP group up", flush=True)
def finish(res):
    # the result is complete here.  The RCCL communicator is NOT torn down: dist.destroy_process_group() aborted (SIGABRT inside the
    # communicator's destruction) in one of ~10 runs of the in-process version of this test on this RCCL / runtime pair, which takes the
    # whole pytest process with it.  A hard exit skips the teardown.
    torch.cuda.synchronize()
    print("RESULT " + json.dumps(res), flush=True)
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(0)
if mode == "basic":
    # eager launches with the bucketed all-reduce issued from the per-block hook on the side stream, one graph followed by the bucketed
    # reduction, and the chain of graph segments with eager collectives between them -- same losses as the plain single-GPU engine
    res = dict(ref=ref)
    eager = IVTrainEngine(build(cfg, params), lr=1e-3, force_comm=True, bucket_bytes=1 << 18)
    res["eager"] = [eager.train_step(v, m, tg)[0].item() for _ in range(3)]
    res["eager_comm"], res["eager_buckets"] = bool(eager.comm), len(eager.reduce_log)
    graph = IVTrainEngine(build(cfg, params), lr=1e-3, force_comm=True, bucket_bytes=1 << 18)
    try:
        graph.capture_step(v, m, tg, L=L)                                        # collectives are never captured implicitly
        res["implicit_capture_refused"] = False
    except RuntimeError:
        res["implicit_capture_refused"] = True
    graph.capture_step(v, m, tg, L=L, defer_reduce=True)
    res["graph"] = [graph.train_step_graphed()[0].item() for _ in range(3)]
    seg = IVTrainEngine(build(cfg, params), lr=1e-3, force_comm=True, bucket_bytes=1 << 18, check_finite=True)
    seg.capture_step(v, m, tg, L=L, segmented=True)
    res["segments"] = len(seg._segments)
    res["last_segment_reduces_vectors"] = bool(seg._segments[-1][2])
    res["every_bucket_once_in_order"] = [list(b) for _, bs, _ in seg._segments for b in bs] == [list(b) for b in seg.buckets]
    res["seg"] = [seg.train_step_graphed()[0].item() for _ in range(3)]
    res["seg_reduce_log_ok"] = [list(b) for b in seg.reduce_log] == [list(b) for b in seg.buckets]
    res["seg_all_loss_mean_err"] = abs(seg.all_loss_mean - res["seg"][-1])
    relm = lambda e_: ((e_.master.double() - base.master.double()).norm() / base.master.double().norm()).item()     # noqa: E731
    res["master_rel"] = dict(eager=relm(eager), graph=relm(graph), seg=relm(seg))
    finish(res)
if mode == "skip":
    # round 6: DropPath skipping (device-side row counts in every launch of the dropped branches) under the multi-rank step modes.  The draws are a
    # static device tensor refreshed before every step, so the plain engine, the eager step with bucketed RCCL all-reduces from the per-block hook
    # and the chain of graph segments see the SAME three draws (three different amounts of work, one branch with every sample dropped).
    from tests.test_droppath_skip_gpu import _build
    B = 8
    video, mask, targets = O.synthetic_batch(cfg, B, 6, seed=9)
    vd, tg8 = video.to(DEV).to(torch.bfloat16), tuple(t.to(DEV).to(torch.bfloat16) for t in targets)
    mk = torch.from_numpy(mask).to(DEV).to(torch.uint8)
    Lv = int((~torch.from_numpy(mask)[0]).sum())
    gen = torch.Generator().manual_seed(77)
    draws = [torch.rand((cfg.depth, 2, B), generator=gen) for _ in range(3)]
    draws[1][cfg.depth - 1, 0] = 0.0
    out = {{}}
    for tag, kw_, how in (("plain", dict(), "eager"), ("eager_comm", dict(force_comm=True, bucket_bytes=1 << 18), "eager"),
                          ("segments", dict(force_comm=True, bucket_bytes=1 << 18), "seg")):
        model = _build(cfg, params, 0.4, True)
        model.residual_dtype = "bf16"
        U = draws[0].clone().to(DEV)
        model._dp_uniform = U
        eng = IVTrainEngine(model, lr=1e-3, weight_decay=0.05, max_grad_norm=1.0, **kw_)
        start = {{k: (v_.clone() if torch.is_tensor(v_) else v_) for k, v_ in eng.state_dict().items()}}
        if how == "seg":
            eng.capture_step(vd, mk, tg8, L=Lv, segmented=True)
            eng.load_state_dict(start)
        losses = []
        for k in range(3):
            U.copy_(draws[k])
            losses.append(float((eng.train_step_graphed() if how == "seg" else eng.train_step(vd, mk, tg8))[0]))
        out[tag] = (losses, eng.master.clone(), len(eng._segments) if how == "seg" else 0, len(eng.reduce_log))
        print("STEP " + tag, flush=True)
    rel = lambda a_, b_: ((a_.double() - b_.double()).norm() / b_.double().norm()).item()     # noqa: E731
    finish(dict(ref=ref, losses={{k: v_[0] for k, v_ in out.items()}}, segments=out["segments"][2], buckets=out["eager_comm"][3],
                seg_vs_eager_comm=rel(out["segments"][1], out["eager_comm"][1]), eager_comm_vs_plain=rel(out["eager_comm"][1], out["plain"][1])))
kw = dict(allreduce_fp32=dict(reduce_dtype="fp32"), zero1=dict(reduce_mode="zero1"), graph_overlap_allreduce=dict(), graph_overlap_zero1=dict(reduce_mode="zero1"),
          segments_zero1=dict(reduce_mode="zero1"), segments_fp32=dict(reduce_dtype="fp32"))[mode]
e = IVTrainEngine(build(cfg, params), lr=1e-3, force_comm=True, bucket_bytes=1 << 18, **kw)
print("STEP engine built", flush=True)
res = dict(ref=ref)
if mode.startswith("graph_overlap"):
    try:
        e.capture_step(v, m, tg, L=L, capture_comm=True)
        print("STEP captured", flush=True)
        res["losses"] = [e.train_step_graphed()[0].item() for _ in range(3)]
    except Exception as ex:                     # an RCCL / runtime combination that cannot capture collectives
        res["losses"] = "capture failed: " + repr(ex)[:300]
elif mode.startswith("segments"):
    e.capture_step(v, m, tg, L=L, segmented=True)
    print("STEP captured", flush=True)
    res["losses"] = [e.train_step_graphed()[0].item() for _ in range(3)]
    res["buckets"] = len(e.reduce_log)
    res["segments"] = len(e._segments)
else:
    res["losses"] = [e.train_step(v, m, tg)[0].item() for _ in range(3)]
    res["buckets"] = len(e.reduce_log)
torch.cuda.synchronize()
print("STEP steps done", flush=True)
n = base.n_mat + base.n_vec
res["master_rel"] = ((e.master[:e.n_mat][:base.n_mat].double() - base.master[:base.n_mat].double()).norm() / base.master[:base.n_mat].double().norm()).item()
finish(res)
"""


def test_bert_large_text_and_fusion_tower_match_oracle_at_config_size():
    """BASELINE configs[3]'s text side: BERT-large, fusion_layer 19, cross-attention to 1408-wide vision tokens
    (multi_modality/scripts/pretraining/stage2/1B/config.py; max_txt_l = 32, 4 frames x 256 patches at mask 0.8 -> 206 vision tokens),
    B = 2 against the fp32 CPU oracle: text-mode states, fusion-mode states, the MLM loss and a sample of its gradients; then the
    stage-2 batch (B = 64: MLM on 64 x 32 tokens + the 192-pair VTM fusion pass) runs forward + backward with finite results."""
    from types import SimpleNamespace
    from internvideo_amd import xbert
    from internvideo_amd.stage2 import MLMLoss, VTC_VTM_Loss
    _threads()
    cfg = O.named_bert_config("bert_large_1B")
    p = O.synthetic_bert_params(cfg, seed=0, std=0.02)
    pc = xbert.BertConfig(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, fusion_layer=cfg.fusion_layer, encoder_width=cfg.encoder_width)
    model = xbert.BertForMaskedLM(pc)
    model.load_state_dict(p, strict=False)
    model = model.to(DEV).train()
    B, L, LV = 2, 32, 206
    ids, mask = O.synthetic_text_batch(cfg, B, L, seed=1)
    g = torch.Generator().manual_seed(2)
    vision = torch.randn(B, LV, cfg.encoder_width, generator=g)
    rng = np.random.RandomState(3)
    draws = (rng.rand(B, L) < 0.5, rng.rand(B, L) < 0.8, rng.rand(B, L) < 0.5, rng.randint(0, cfg.vocab_size, size=(B, L)).astype(np.int64))
    m_ids, m_labels = O.mlm_mask_tokens(ids, *draws, cfg)
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    t_ref = O.bert_model(pr, cfg, input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), mode="text")
    f_ref = O.bert_model(pr, cfg, encoder_embeds=t_ref, attention_mask=torch.from_numpy(mask), encoder_hidden_states=vision, mode="fusion")
    l_ref = O.mlm_loss(pr, cfg, torch.from_numpy(m_ids), torch.from_numpy(m_labels), torch.from_numpy(mask), vision)
    l_ref.backward()
    d_ids, d_mask, d_vis = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV), vision.to(DEV)
    with torch.no_grad():
        t = model.bert(d_ids, attention_mask=d_mask, mode="text").last_hidden_state
        f = model.bert(encoder_embeds=t, attention_mask=d_mask, encoder_hidden_states=d_vis, mode="fusion").last_hidden_state
    e_text, e_fused = rel(t.float().cpu(), t_ref.detach()), rel(f.float().cpu(), f_ref.detach())
    tok = SimpleNamespace(pad_token_id=cfg.pad_token_id, cls_token_id=cfg.cls_token_id, mask_token_id=cfg.mask_token_id)
    crit = MLMLoss(0.5, tok)
    text = SimpleNamespace(input_ids=d_ids, attention_mask=d_mask)
    loss = crit.mlm_loss(model, text, d_vis, None, draws=tuple(torch.from_numpy(np.asarray(d)) for d in draws))
    loss.backward()
    named = dict(model.named_parameters())
    keys = ["bert.embeddings.position_embeddings.weight", "bert.encoder.layer.0.attention.self.query.weight", "bert.encoder.layer.11.output.dense.weight",
            "bert.encoder.layer.19.crossattention.self.key.weight", "bert.encoder.layer.23.crossattention.output.dense.weight",
            "bert.encoder.layer.23.output.LayerNorm.weight", "cls.predictions.transform.dense.weight", "cls.predictions.bias",
            "bert.embeddings.word_embeddings.weight"]
    gerr = {k: rel(named[k].grad.float().cpu(), pr[k].grad) for k in keys}
    e_loss = abs(loss.item() - l_ref.item()) / abs(l_ref.item())
    _note("bert_large_B2", dict(text_rel=e_text, fused_rel=e_fused, mlm_loss_rel=e_loss, mlm_loss=loss.item(), grad_rel=gerr))
    assert e_text < 1.5e-2 and e_fused < 1.5e-2, (e_text, e_fused)
    assert e_loss < 5e-3, (loss.item(), l_ref.item())
    assert max(gerr.values()) < 4e-2, gerr
    # ... and against the REFERENCE's own BertForMaskedLM at this size (tests/golden/bert_large_digest.npz, make_golden_bert_large.py): the
    # same bars on the digest of its states (first rows + 16 random projections of every row), its MLM loss and its gradient norms
    gd = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bert_large_digest.npz"))
    assert [int(x) for x in gd["meta"]] == [B, L, LV]
    for key, tt in (("text", t), ("fused", f)):
        rows = tt.float().cpu().double().numpy().reshape(-1, tt.shape[-1])
        C = rows.shape[1]
        proj = np.random.Generator(np.random.PCG64(777 + C)).standard_normal((C, 16)).astype(np.float32) / np.sqrt(C).astype(np.float32)
        e_rows = np.linalg.norm(rows[:3] - gd[key + ":rows"]) / np.linalg.norm(gd[key + ":rows"])
        e_proj = np.linalg.norm(rows @ proj.astype(np.float64) - gd[key + ":proj"]) / np.linalg.norm(gd[key + ":proj"])
        assert e_rows < 1.5e-2 and e_proj < 1.5e-2, (key, e_rows, e_proj)
    assert abs(loss.item() - float(gd["mlm_loss"][0])) / float(gd["mlm_loss"][0]) < 5e-3
    for k in keys:
        want = float(gd["grad:" + k + ":norm"][0])
        assert abs(named[k].grad.float().norm().item() - want) < 4e-2 * want, k
    # ---- VTM at config size, by VALUE (only ranges were asserted): the fusion pass over [pos | (neg video, text) | (video, neg
    # text)] with the negatives fixed, against the oracle (pinned to the reference's vtm_loss at the tiny size: tests/golden/bert_tiny.npz);
    # loss 5e-3.  Gradients: at initialisation the six fused rows' [CLS] states are nearly equal and (p - y) sums to ~0 over them, so the head /
    # vision gradients are differences of nearly equal vectors -- a bf16 pass through 24 layers shows up amplified there.  The bar is therefore
    # calibrated like the student's: max(4e-2, 3 x the ORACLE's own bf16-vs-fp32 deviation), measured here on the same inputs (6.1e-2 / 5.3e-2)
    gi = torch.Generator().manual_seed(11)
    itm_w, itm_b = torch.randn(2, cfg.hidden_size, generator=gi) * 0.02, torch.randn(2, generator=gi) * 0.02
    vneg_c, tneg_c = torch.tensor([1, 0]), torch.tensor([1, 0])
    vis_ref = vision.clone().requires_grad_(True)
    w_ref, b_ref = itm_w.clone().requires_grad_(True), itm_b.clone().requires_grad_(True)
    with torch.no_grad():
        pd = {k: v.detach() for k, v in pr.items()}
    lv_ref = O.vtm_loss_given_negatives(pd, cfg, w_ref, b_ref, vis_ref, t_ref.detach(), torch.from_numpy(mask), vneg_c, tneg_c)
    lv_ref.backward()
    head_c = torch.nn.Linear(cfg.hidden_size, 2).to(DEV)
    with torch.no_grad():
        head_c.weight.copy_(itm_w); head_c.bias.copy_(itm_b)
    model.zero_grad(set_to_none=True)
    dv = d_vis.bfloat16().requires_grad_(True)
    vpn = torch.nn.functional.normalize(torch.randn(B, 512, generator=gi), dim=-1).to(DEV)
    tpn = torch.nn.functional.normalize(torch.randn(B, 512, generator=gi), dim=-1).to(DEV)
    lv = VTC_VTM_Loss(True).vtm_loss(model.bert, head_c, torch.tensor(0.07, device=DEV), dv, t.detach(), vpn, tpn, d_mask, torch.arange(B, device=DEV),
                                     neg_indices=(vneg_c.to(DEV), tneg_c.to(DEV)))
    lv.backward()
    e_vtm = abs(lv.item() - lv_ref.item()) / abs(lv_ref.item())
    e_gv, e_gw = rel(dv.grad.float().cpu(), vis_ref.grad), rel(head_c.weight.grad.float().cpu(), w_ref.grad)
    bf = torch.bfloat16
    vis_b, w_b, b_b = (x.detach().clone().to(bf).requires_grad_(True) for x in (vision, itm_w, itm_b))
    lv_b = O.vtm_loss_given_negatives({k: v.to(bf) for k, v in pd.items()}, cfg, w_b, b_b, vis_b, t_ref.detach().to(bf), torch.from_numpy(mask), vneg_c, tneg_c)
    lv_b.backward()
    cal_gv, cal_gw = rel(vis_b.grad.float(), vis_ref.grad), rel(w_b.grad.float(), w_ref.grad)
    _note("bert_large_B2_vtm", dict(vtm_loss=lv.item(), vtm_loss_oracle=lv_ref.item(), rel=e_vtm, grad_vision_rel=e_gv, grad_itm_w_rel=e_gw,
                                    oracle_bf16_twin=dict(loss_rel=abs(lv_b.item() - lv_ref.item()) / abs(lv_ref.item()), grad_vision_rel=cal_gv, grad_itm_w_rel=cal_gw)))
    assert e_vtm < 5e-3, (lv.item(), lv_ref.item())
    assert e_gv < max(4e-2, 3.0 * cal_gv) and e_gw < max(4e-2, 3.0 * cal_gw), (e_gv, e_gw, cal_gv, cal_gw)
    del pr, t_ref, f_ref, pd
    # ---- the stage-2 batch: 64 texts x 32 tokens, 206 vision tokens per clip ----
    model.zero_grad(set_to_none=True)
    B = 64
    ids, mask = O.synthetic_text_batch(cfg, B, L, seed=4)
    d_ids, d_mask = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    d_vis = torch.randn(B, LV, cfg.encoder_width, device=DEV).bfloat16().requires_grad_(True)
    text = SimpleNamespace(input_ids=d_ids, attention_mask=d_mask)
    head = torch.nn.Linear(cfg.hidden_size, 2).to(DEV)
    vtm = VTC_VTM_Loss(True)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(2):
        model.zero_grad(set_to_none=True)
        if it == 1:
            ev[0].record()
        t = model.bert(d_ids, attention_mask=d_mask, mode="text").last_hidden_state
        vp = torch.nn.functional.normalize(torch.randn(B, 512, device=DEV), dim=-1)
        tp = torch.nn.functional.normalize(torch.randn(B, 512, device=DEV), dim=-1)
        l_vtm = vtm.vtm_loss(model.bert, head, torch.tensor(0.07, device=DEV), d_vis, t, vp, tp, d_mask, torch.arange(B, device=DEV))
        l_mlm = crit.mlm_loss(model, text, d_vis, None)
        (l_vtm + l_mlm).backward()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1])
    assert torch.isfinite(l_vtm) and torch.isfinite(l_mlm) and 0.3 < l_vtm.item() < 2.0 and 8.0 < l_mlm.item() < 13.0, (l_vtm.item(), l_mlm.item())
    assert torch.isfinite(d_vis.grad.float()).all() and float(d_vis.grad.float().abs().max()) > 0
    for k in keys:
        assert torch.isfinite(named[k].grad.float()).all(), k
    _note("bert_large_B64_text_side_ms", dict(ms=ms, vtm=l_vtm.item(), mlm=l_mlm.item(), what="encode_text + VTM (192 pairs) + MLM, fwd + bwd, eager"))



def _run_rccl_mode(mode):
    import subprocess
    import sys
    import tempfile
    fd, script = tempfile.mkstemp(prefix="ivh_rccl_modes_", suffix=".py")      # a scratch file of this test run, outside the repository
    with os.fdopen(fd, "w") as f:
        f.write(_RCCL_MODES.format(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, script, mode], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    steps = [l for l in r.stdout.splitlines() if l.startswith("STEP ")]
    _note("rccl_1rank_" + mode, dict(returncode=r.returncode, steps=steps, result=(json.loads(line[0][7:]) if line else None), stderr_tail=r.stderr[-1500:]))
    return r, (json.loads(line[0][7:]) if line else None), steps


def test_one_rank_rccl_eager_overlap_and_graph_deferred_reduce_agree():
    """the multi-rank step modes on real RCCL (1-rank group on this GPU, own subprocess): eager launches with the bucketed all-reduce issued
    from the per-block hook on the side stream, graph replay followed by the bucketed reduction, and bench.py's default -- a chain of graphs
    cut at the buckets with eager RCCL all-reduces between them: same losses as the plain single-GPU engine."""
    r, res, steps = _run_rccl_mode("basic")
    assert r.returncode == 0 and res, (r.returncode, steps, r.stderr[-3000:])
    ref = res["ref"]
    assert res["eager"] == ref, res
    assert res["eager_comm"] and res["eager_buckets"] >= 2                          # several buckets went through RCCL
    assert res["implicit_capture_refused"]
    assert max(abs(a - b) / abs(b) for a, b in zip(res["graph"], ref)) < 1e-5, res
    assert res["segments"] >= 3 and res["last_segment_reduces_vectors"] and res["every_bucket_once_in_order"], res
    assert max(abs(a - b) / abs(b) for a, b in zip(res["seg"], ref)) < 1e-5, res
    assert res["seg_reduce_log_ok"] and res["seg_all_loss_mean_err"] < 1e-6, res
    assert res["master_rel"]["eager"] == 0.0 and res["master_rel"]["graph"] < 1e-4 and res["master_rel"]["seg"] < 1e-4, res


@pytest.mark.parametrize("mode", ["allreduce_fp32", "zero1", "segments_zero1", "segments_fp32"])
def test_one_rank_rccl_reduce_modes(mode):
    """real RCCL (1-rank group on this GPU, own subprocess): the fp32-accumulating all-reduce and the ZeRO-1 path (all-to-all + fp32 shard
    sum + sharded AdamW + all-gather) reproduce the plain engine's losses and weights -- issued from eager launches and from the segmented
    graph chain (capture_step(segmented=True): the collectives stay ordinary RCCL calls between the graph segments)"""
    r, res, steps = _run_rccl_mode(mode)
    assert r.returncode == 0 and res, (r.returncode, steps, r.stderr[-3000:])
    assert max(abs(a - b) / abs(b) for a, b in zip(res["losses"], res["ref"])) < 1e-5, res
    assert res["buckets"] >= 2 and res["master_rel"] < (1e-4 if mode.startswith("segments") else 1e-6), res


def test_one_rank_rccl_step_modes_with_droppath_skipping_on_the_same_draws():
    """round 6: the multi-rank step modes with DropPath skipping, on draws that are the same for every mode (a static device tensor refreshed per
    step): the eager step with bucketed RCCL all-reduces and bench.py's default chain of graph segments follow the plain single-GPU engine over
    three steps that keep different numbers of samples (one branch keeps none) -- no count may be frozen into a segment."""
    r, res, steps = _run_rccl_mode("skip")
    assert r.returncode == 0 and res, (r.returncode, steps, r.stderr[-3000:])
    ls = res["losses"]
    assert len(set(ls["plain"])) == 3, ls
    assert ls["eager_comm"] == ls["plain"], ls                                   # same kernels, same order; a 1-rank all-reduce is the identity
    assert max(abs(a - b) / abs(b) for a, b in zip(ls["segments"], ls["plain"])) < 1e-5, ls
    assert res["segments"] >= 3 and res["buckets"] >= 2, res
    assert res["eager_comm_vs_plain"] == 0.0 and res["seg_vs_eager_comm"] < 1e-4, res


@pytest.mark.parametrize("mode", ["graph_overlap_allreduce", "graph_overlap_zero1"])
def test_one_rank_rccl_step_captured_with_its_collectives(mode):
    """HIP-graph capture of the step INCLUDING its bucketed collectives (overlap without per-step host work).  Whether collectives can be
    captured depends on the RCCL / runtime pair: when capture is refused (or the process dies in it) the engine's documented fallback is
    capture_step(defer_reduce=True) (test_model_gpu.py) and this test xfails with the evidence recorded in gpurun_out/parity_fullsize.json."""
    r, res, steps = _run_rccl_mode(mode)
    if r.returncode != 0 or not res or isinstance(res["losses"], str):
        pytest.xfail(f"{mode}: rc {r.returncode}, progress {steps}, {res['losses'] if res else r.stderr[-400:]}")
    assert max(abs(a - b) / abs(b) for a, b in zip(res["losses"], res["ref"])) < 1e-5, res


def test_shipped_stage2_1B_config_runs_one_training_step(tmp_path):
    """BASELINE configs[3] built from the reference's own config keys (scripts/pretraining/stage2/1B/config.py: 4 x 224^2 frames, mask 0.8,
    BERT-large with `gradient_checkpointing = True`, vtc + vtm + mlm) at B = 4: one forward + backward + AdamW step of the assembled
    InternVideo2_Stage2_visual with BERT's default dropout on; finite losses in the expected ranges, gradients in both towers and every head,
    and the checkpointed text tower gives the same VTC loss as the stored-activation run on the same draws."""
    from types import SimpleNamespace
    from internvideo_amd import xbert
    from internvideo_amd.stage2 import InternVideo2_Stage2_visual
    from tests.test_bert_host import BERT_LARGE_JSON, shipped_stage2_1B_config
    path = tmp_path / "config_bert_large.json"
    path.write_text(json.dumps(BERT_LARGE_JSON))
    tok = SimpleNamespace(pad_token_id=0, cls_token_id=101, mask_token_id=103)
    torch.manual_seed(0)
    with torch.device(DEV):
        model = InternVideo2_Stage2_visual(shipped_stage2_1B_config(str(path)), tok, True)
    model.train()
    assert model.text_encoder.config.gradient_checkpointing is True
    B, Lt = 4, 32
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(1000, 30000, (B, Lt), generator=g)
    ids[:, 0] = 101
    mask = torch.ones(B, Lt, dtype=torch.long)
    mask[1, 20:] = 0
    ids[1, 20:] = 0
    text = SimpleNamespace(input_ids=ids.to(DEV), attention_mask=mask.to(DEV))
    image = torch.randn(B, 4, 3, 224, 224, generator=g).to(DEV)
    idx = torch.arange(B, device=DEV)
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-5)
    losses_seen = {}
    for cp in (True, False):
        model.text_encoder.config.gradient_checkpointing = cp
        opt.zero_grad(set_to_none=True)
        np.random.seed(0); torch.manual_seed(1); xbert._DROP_CALLS = 5000
        out = model(image, text, idx, media_type="video")
        assert set(out) == {"loss_uta", "loss_vtc", "loss_vtm", "loss_mlm"}
        sum(out.values()).backward()
        losses_seen[cp] = {k: float(v) for k, v in out.items()}
        for k in ("loss_vtc", "loss_vtm", "loss_mlm"):
            assert np.isfinite(losses_seen[cp][k]) and losses_seen[cp][k] > 0, (cp, losses_seen)
        named = dict(model.named_parameters())
        for k in ("vision_encoder.blocks.0.attn.qkv.weight", "vision_encoder.blocks.39.mlp.fc2.weight", "text_encoder.bert.encoder.layer.0.attention.self.query.weight",
                  "text_encoder.bert.encoder.layer.23.crossattention.self.key.weight", "text_encoder.cls.predictions.transform.dense.weight",
                  "vision_proj.weight", "text_proj.weight", "itm_head.weight", "temp"):
            gr = named[k].grad
            assert gr is not None and torch.isfinite(gr.float()).all() and float(gr.float().abs().max()) > 0, (cp, k)
    assert 0.5 < losses_seen[True]["loss_vtc"] < 3.0 and 8.0 < losses_seen[True]["loss_mlm"] < 13.0, losses_seen
    # forward values do not depend on whether activations are stored or recomputed (same seeds, same masks)
    assert losses_seen[True]["loss_vtc"] == losses_seen[False]["loss_vtc"] and losses_seen[True]["loss_mlm"] == losses_seen[False]["loss_mlm"], losses_seen
    opt.step()
    _note("stage2_shipped_config_B4", losses_seen)
