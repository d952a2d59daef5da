"""MI355X parity of the model flavours around the pre-training student, against fixtures made by the REFERENCE's own code
(tests/golden/flavours.npz): the distillation student (internvideo_amd.internvideo2_distill), the stage-2 vision encoder
(internvideo_amd.mm_internvideo2: masked video, mask=None, image mode, early exit) and the teacher-target gather.
Tolerances as tests/test_model_gpu.py: indices / copies bit-exact; outputs rel-L2 <= 1e-2; loss <= 1e-3 relative; gradients
<= max(3e-2, 3 x the reference's own bf16-vs-fp32 discrepancy) (16x16 corners: floor 5e-2)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from internvideo_amd import functional as Fn, internvideo2_distill as D, internvideo2_pretrain as M, masking, mm_internvideo2 as V, ops  # noqa: E402
from oracle import internvideo2_oracle as O  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "flavours.npz")
DEV = "cuda"


def rel(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def check_grads(g, pre, model, errpre):
    sd = dict(model.named_parameters())
    worst = {}
    for key in g.files:
        if key.startswith(pre + "grad:"):
            k = key[len(pre) + 5:]
            worst[k] = rel(sd[k].grad, g[key])
        elif key.startswith(pre + "gradnorm:"):
            k = key[len(pre) + 9:]
            gr = sd[k].grad
            g2 = gr.reshape(gr.shape[0], -1) if gr.dim() == 5 else gr.reshape(-1, gr.shape[-1])
            worst["corner:" + k] = rel(g2[:16, :16], g[pre + "gradcorner:" + k])
            worst["norm:" + k] = abs(gr.double().norm().item() - g[key][0]) / g[key][0]

    def tol(k):
        floor = 5e-2 if k.startswith("corner:") else 3e-2
        e = errpre + k
        return max(floor, 3.0 * float(g[e][0])) if e in g.files else floor
    bad = {k: (v, tol(k)) for k, v in worst.items() if v > tol(k)}
    assert not bad, bad
    return len(worst)


def build_dist(cfg, params):
    m = D.DistInternVideo2(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
                           num_frames=cfg.num_frames, drop_path_rate=0.0, attn_pool_num_heads=cfg.attn_pool_num_heads,
                           clip_embed_dim=cfg.clip_embed_dim, clip_teacher_embed_dim=cfg.clip_teacher_embed_dim,
                           clip_teacher_final_dim=cfg.clip_teacher_final_dim, clip_return_layer=cfg.clip_return_layer,
                           clip_student_return_index=list(cfg.clip_return_index_override),
                           clip_student_decoder={"linear": "Linear_Decoder", "mlp": "MLP_Decoder"}[cfg.clip_decoder_kind])
    m.load_state_dict(params, strict=True)
    return m.to(DEV).train()


def build_mm(cfg, params):
    m = V.PretrainInternVideo2(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
                               num_frames=cfg.num_frames, drop_path_rate=0.0, attn_pool_num_heads=cfg.attn_pool_num_heads,
                               clip_embed_dim=cfg.clip_embed_dim, clip_teacher_embed_dim=cfg.clip_teacher_embed_dim,
                               clip_teacher_final_dim=cfg.clip_teacher_final_dim, clip_return_layer=cfg.clip_return_layer,
                               sep_image_video_pos_embed=cfg.sep_image_video_pos_embed)
    m.load_state_dict(params, strict=True)
    return m.to(DEV).train()


def test_distill_student_matches_reference_golden():
    g = np.load(GOLD)
    cfg = O.named_config("dist64")
    params = O.synthetic_params(cfg, seed=2)
    video, mask, targets = O.synthetic_batch(cfg, 2, 4, seed=2)
    model = build_dist(cfg, params)
    oc, of = model(video.to(DEV), torch.from_numpy(mask))
    assert oc.dtype == torch.bfloat16 and tuple(oc.shape) == tuple(g["dist:x_clip_align"].shape)
    e = [rel(oc.float(), g["dist:x_clip_align"]), rel(of.float(), g["dist:x_align"])]
    assert max(e) < 1e-2, e
    tc, tf = targets[0].to(DEV), targets[1].to(DEV)
    loss = (2 - 2 * (oc.float() * tc).sum(-1)).mean() + (2 - 2 * (of.float() * tf).sum(-1)).mean()
    ref = g["dist:losses"]
    assert abs(loss.item() - ref[0]) / abs(ref[0]) < 1e-3, (loss.item(), ref[0])
    loss.backward()
    assert check_grads(g, "dist:", model, "dist:bf16err:") >= 8
    # fused-loss path (engines/engine_for_distill.py:107-121) == the drop-in forward + torch loss
    model.zero_grad(set_to_none=True)
    l2, (lm, lf) = model.forward_loss(video.to(DEV), torch.from_numpy(mask), (tc, tf))
    assert abs(l2.item() - ref[0]) / abs(ref[0]) < 1e-3 and abs(lm.item() - ref[1]) / abs(ref[1]) < 1e-3 and abs(lf.item() - ref[2]) / abs(ref[2]) < 1e-3
    l2.backward()
    assert check_grads(g, "dist:", model, "dist:bf16err:") >= 8


def test_distill_student_trains_through_the_native_engine():
    from internvideo_amd.engine import IVTrainEngine
    cfg = O.named_config("dist64")
    params = O.synthetic_params(cfg, seed=2)
    video, mask, targets = O.synthetic_batch(cfg, 2, 4, seed=2)
    model = build_dist(cfg, params)
    eng = IVTrainEngine(model, lr=1e-3)
    tg = (targets[0].to(DEV), targets[1].to(DEV))
    losses = []
    for _ in range(4):
        l, _ = eng.train_step(video.to(DEV), torch.from_numpy(mask), tg)
        losses.append(l.item())
    ref = np.load(GOLD)["dist:losses"][0]
    assert abs(losses[0] - ref) / abs(ref) < 1e-3 and losses[-1] < losses[0], losses


@pytest.mark.parametrize("name,seed", [("mm88", 4), ("mm64", 5)])
def test_stage2_vision_encoder_matches_reference_golden(name, seed):
    g = np.load(GOLD)
    cfg = O.named_config(name)
    pre = name + ":"
    params = O.synthetic_params(cfg, seed=seed)
    video, mask, targets = O.synthetic_batch(cfg, 2, 5, seed=seed)
    rng = np.random.Generator(np.random.PCG64(77 + seed))
    image = torch.from_numpy(rng.random((2, cfg.in_chans, 1, cfg.img_size, cfg.img_size), dtype=np.float32))
    img_mask = torch.from_numpy(g[pre + "img_mask"])
    # (1) masked video: the 4-tuple, and gradients through all four outputs
    model = build_mm(cfg, params)
    x_vis, x_pool, x_clip, x_align = model(video.to(DEV), torch.from_numpy(mask), False)
    assert x_vis.dtype == torch.bfloat16 and tuple(x_vis.shape) == tuple(g[pre + "video:x_vis"].shape)
    e = {k: rel(v.float(), g[pre + "video:" + k]) for k, v in dict(x_vis=x_vis, x_pool_vis=x_pool, x_clip_align=x_clip, x_align=x_align).items()}
    assert max(e.values()) < 1e-2, e
    w = torch.from_numpy(np.random.Generator(np.random.PCG64(5)).standard_normal(tuple(x_vis.shape)).astype(np.float32)).to(DEV)
    loss = ((2 - 2 * (x_clip.float() * targets[0].to(DEV)).sum(-1)).mean() + (2 - 2 * (x_align.float() * targets[1].to(DEV)).sum(-1)).mean()
            + (x_vis.float() * w).mean() + x_pool.float().square().mean())
    ref = g[pre + "video:loss"][0]
    assert abs(loss.item() - ref) / abs(ref) < 1e-3, (loss.item(), ref)
    loss.backward()
    assert check_grads(g, pre + "video:", model, pre + "bf16err:") >= 8
    # (2) mask=None: every token kept
    model.eval()
    with torch.no_grad():
        out = model(video.to(DEV), None, False)
    e = {k: rel(v.float(), g[pre + "nomask:" + k]) for k, v in zip(("x_vis", "x_pool_vis", "x_clip_align", "x_align"), out)}
    assert max(e.values()) < 1e-2, e
    # (4) early exit, x_vis only
    with torch.no_grad():
        xv = model(video.to(DEV), torch.from_numpy(mask), False, x_vis_return_idx=-2, x_vis_only=True)
    assert rel(xv.float(), g[pre + "early:x_vis"]) < 1e-2
    # (3) image mode: T = 1, image positional tables (separate for mm88, frame-averaged for mm64), gradients reach the tables
    model = build_mm(cfg, params)
    x_vis, x_pool, x_clip, x_align = model(image.to(DEV), img_mask, True)
    e = {k: rel(v.float(), g[pre + "image:" + k]) for k, v in dict(x_vis=x_vis, x_clip_align=x_clip, x_align=x_align).items()}
    assert max(e.values()) < 1e-2, e
    loss = x_clip.float().sum(-1).mean() + x_align.float().sum(-1).mean() + x_vis.float().square().mean()
    ref = g[pre + "image:loss"][0]
    # this probe loss sums signed components of l2-normalised bf16 vectors (cancellation: |loss| ~ 0.4 from terms of size ~10), so
    # bf16 output rounding alone moves it by several 1e-3 relative; the outputs themselves are held to 1e-2 above
    assert abs(loss.item() - ref) / abs(ref) < 2e-2, (loss.item(), ref)
    loss.backward()
    assert check_grads(g, pre + "image:", model, pre + "bf16err:") >= 4
    if cfg.sep_image_video_pos_embed:
        assert model.pos_embed.grad is None and model.img_pos_embed.grad is not None
    else:
        assert model.pos_embed.grad is not None


def test_clip_teacher_matches_reference_golden():
    """the frozen CLIP teacher on the student's kernels (hd = 128, per-frame sequences) + frame merge / l2 / attention-map tails
    vs the reference's InternVL_CLIP.  Tolerance: 1e-2 rel-L2 (the reference's own bf16 run is 5e-3 off its fp32 run)."""
    from internvideo_amd.internvl_clip_vision import InternVL_CLIP
    g = np.load(GOLD)
    cfg = O.named_config("teach128")
    params = O.synthetic_teacher_params(cfg, seed=6)
    rng = np.random.Generator(np.random.PCG64(66))
    video = torch.from_numpy(rng.random((2, cfg.in_chans, cfg.num_frames, cfg.img_size, cfg.img_size), dtype=np.float32))
    m = InternVL_CLIP(img_size=cfg.img_size, embed_dim=cfg.embed_dim, num_heads=cfg.num_heads, depth=cfg.depth, mlp_ratio=cfg.mlp_ratio,
                      attn_pool_num_heads=cfg.attn_pool_num_heads, clip_embed_dim=cfg.clip_embed_dim, clip_return_layer=2)
    m.load_state_dict(params, strict=True)
    m = m.to(DEV).eval()
    z, x, attn = m(video.to(DEV))
    assert z.dtype == torch.bfloat16 and tuple(z.shape) == tuple(g["teach:z"].shape) and tuple(attn.shape) == tuple(g["teach:attn"].shape)
    e = dict(z=rel(z.float(), g["teach:z"]), x=rel(x.float(), g["teach:x"]), attn=rel(attn, g["teach:attn"]))
    assert max(e.values()) < 1e-2, e
    assert torch.allclose(attn.sum(1).cpu(), torch.from_numpy(g["teach:attn"].sum(1)), atol=5e-3)   # probability mass on the patch keys
    assert torch.allclose(z.float().norm(dim=-1), torch.ones_like(z[..., 0], dtype=torch.float32), atol=1e-2)
    # the attention map drives the mask exactly as engines/engine_for_pretraining.py:105-125: mask -> gather of the teacher targets
    mask = masking.attention_guided_mask(attn, 2, 0.75)
    vis, _ = M.build_gather_indices(mask, DEV)
    tg = masking.gather_visible(z, vis_idx=vis)
    want = z[~mask.unsqueeze(0).repeat(z.shape[0], 1, 1)].reshape(z.shape[0], 2, -1, z.shape[-1])
    assert torch.equal(tg, want)
    # bf16 module (the reference recipe runs the teacher under bf16 autocast) and the 'none' normalisation branch
    mb = InternVL_CLIP(img_size=cfg.img_size, embed_dim=cfg.embed_dim, num_heads=cfg.num_heads, depth=cfg.depth, mlp_ratio=cfg.mlp_ratio,
                       attn_pool_num_heads=cfg.attn_pool_num_heads, clip_embed_dim=cfg.clip_embed_dim, clip_return_layer=2,
                       clip_norm_type='none', return_attn=False)
    mb.load_state_dict(params, strict=True)
    mb = mb.to(DEV).bfloat16().eval()
    z2, x2 = mb(video.to(DEV).bfloat16())
    assert tuple(z2.shape) == (2, 8, 17, 256) and tuple(x2.shape) == (8, 64)
    zz = z2.float().view(2, 2, 4, 17, 256)
    merged = torch.cat([zz[:, :, :, :1].mean(2), zz[:, :, :, 1:].reshape(2, 2, 64, 256)], 2)
    merged = merged / merged.norm(dim=-1, keepdim=True)
    assert rel(merged, g["teach:z"]) < 1.5e-2


def test_clip_teacher_with_fp8_block_gemms_tracks_the_bf16_teacher():
    """`teacher.fp8_gemm = True` (opt-in; bench.py --teacher-fp8): the frozen CLIP teacher's block GEMMs on the e4m3 MFMA path, weights
    quantised once with per-channel scales.  Stated tolerance against the reference's golden outputs: the l2-normalised targets z and the
    pooled feature x within 4e-2 rel-L2 (bf16 teacher: 1e-2), per-token cosine to the bf16 teacher's targets > 0.998, and the
    top quarter of the attention map (what the attention-guided mask favours) overlaps the bf16 teacher's by >= 90 % on average."""
    from internvideo_amd.internvl_clip_vision import InternVL_CLIP
    g = np.load(GOLD)
    cfg = O.named_config("teach128")
    params = O.synthetic_teacher_params(cfg, seed=6)
    rng = np.random.Generator(np.random.PCG64(66))
    video = torch.from_numpy(rng.random((2, cfg.in_chans, cfg.num_frames, cfg.img_size, cfg.img_size), dtype=np.float32))
    m = InternVL_CLIP(img_size=cfg.img_size, embed_dim=cfg.embed_dim, num_heads=cfg.num_heads, depth=cfg.depth, mlp_ratio=cfg.mlp_ratio,
                      attn_pool_num_heads=cfg.attn_pool_num_heads, clip_embed_dim=cfg.clip_embed_dim, clip_return_layer=2)
    m.load_state_dict(params, strict=True)
    m = m.to(DEV).eval()
    z16, x16, attn16 = m(video.to(DEV))
    m.fp8_gemm = True
    z8, x8, attn8 = m(video.to(DEV))
    assert not torch.equal(z8, z16)                                          # it really is another path
    e = dict(z=rel(z8.float(), g["teach:z"]), x=rel(x8.float(), g["teach:x"]), attn=rel(attn8, g["teach:attn"]))
    assert e["z"] < 4e-2 and e["x"] < 4e-2 and e["attn"] < 8e-2, e
    cos = (z8.float() * z16.float()).sum(-1)
    assert cos.min().item() > 0.998, cos.min().item()
    # masks: compare the kept sets through the sampling weights themselves (multinomial draws differ run to run): the top quarter by weight
    k = attn16.shape[1] // 4
    top16 = attn16.topk(k, dim=1).indices
    top8 = attn8.topk(k, dim=1).indices
    overlap = torch.stack([torch.isin(top8[i], top16[i]).float().mean() for i in range(top16.shape[0])])
    assert overlap.mean().item() >= 0.9 and overlap.min().item() >= 0.75, overlap      # k = 4 of 16 keys here: one swap = 0.75
    wq, sw = Fn.frozen_fp8_weight(m.blocks[0].attn.qkv.weight)               # quantised once, per-channel scales, cached on the parameter
    assert sw.numel() == m.blocks[0].attn.qkv.weight.shape[0] and Fn.frozen_fp8_weight(m.blocks[0].attn.qkv.weight)[0] is wq


@pytest.mark.parametrize("fp8", [False, True])
def test_clip_teacher_at_real_width_matches_the_reference_digest(fp8):
    """The frozen CLIP teacher at InternVL-6B's width and sequence geometry (3200 wide, 25 heads of 128, 257-token per-frame sequences,
    8 frames, 16 pooling heads of 200; depth 2) against a digest of the REFERENCE's own InternVL_CLIP at that size
    (tests/golden/clip_teacher_fullwidth_digest.npz, make_golden_teacher_fullwidth.py; the oracle is held to the same digest on the CPU):
    bf16 path 1e-2 on targets / pooled feature / attention map; the opt-in fp8 block GEMMs 6e-2 / 6e-2 / 1e-1 (measured 4.9e-2 on the targets)."""
    from internvideo_amd.internvl_clip_vision import InternVL_CLIP
    from tests.test_flavours_oracle import _clip_teacher_fullwidth
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "clip_teacher_fullwidth_digest.npz"))
    cfg, p, video = _clip_teacher_fullwidth()
    m = InternVL_CLIP(img_size=cfg.img_size, embed_dim=cfg.embed_dim, num_heads=cfg.num_heads, depth=cfg.depth, mlp_ratio=cfg.mlp_ratio,
                      attn_pool_num_heads=cfg.attn_pool_num_heads, clip_embed_dim=cfg.clip_embed_dim, clip_return_layer=2)
    m.load_state_dict(p, strict=True)
    m = m.to(DEV).eval()
    m.fp8_gemm = fp8
    z, x, attn = m(video.to(DEV))
    tol = dict(z=6e-2, x=6e-2, attn=1e-1) if fp8 else dict(z=1e-2, x=1e-2, attn=1e-2)      # fp8 measured: z 4.9e-2 (LayerScale ~0.5 here: the blocks dominate the stream)
    for name, t in (("z", z), ("x", x), ("attn", attn)):
        assert tuple(t.shape) == tuple(int(i) for i in g[name + ":shape"]), name
        rows = t.detach().float().cpu().double().numpy().reshape(-1, t.shape[-1])
        C = rows.shape[1]
        proj = np.random.Generator(np.random.PCG64(777 + C)).standard_normal((C, 16)).astype(np.float32) / np.sqrt(C).astype(np.float32)
        e_rows = np.linalg.norm(rows[:3] - g[name + ":rows"]) / np.linalg.norm(g[name + ":rows"])
        e_proj = np.linalg.norm(rows @ proj.astype(np.float64) - g[name + ":proj"]) / np.linalg.norm(g[name + ":proj"])
        assert e_rows < tol[name] and e_proj < tol[name], (name, fp8, e_rows, e_proj)


def test_teacher_tail_kernels_vs_torch():
    gen = torch.Generator().manual_seed(1)
    B, T, L, C = 3, 4, 9, 200
    x = torch.randn(B * T * L, C, generator=gen)
    xx = x.view(B, T, L, C)
    want = torch.cat([xx[:, :, :1].mean(1), xx[:, :, 1:].reshape(B, T * (L - 1), C)], 1)
    wl2 = want / want.norm(dim=-1, keepdim=True)
    assert rel(ops.frames_merge_l2(x.to(DEV), B, T, L, l2=True, out_fp32=True), wl2) < 1e-6
    assert rel(ops.frames_merge_l2(x.to(DEV), B, T, L, l2=False, out_fp32=True), want) < 1e-6
    assert rel(ops.frames_merge_l2(x.to(DEV).bfloat16(), B, T, L, l2=True).float(), wl2) < 6e-3
    p = torch.randn(B * T, 96, generator=gen)
    wm = p.view(B, T, 96).mean(1)
    assert rel(ops.frames_merge_l2(p.to(DEV), B, T, 1, l2=True, out_fp32=True).view(B, 96), wm / wm.norm(dim=-1, keepdim=True)) < 1e-6
    S, Lk, H, hd = 5, 300, 4, 40                     # L > 256: more than one key per thread
    q = torch.randn(S, H, hd, generator=gen).bfloat16()
    k = torch.randn(S, Lk, H, hd, generator=gen).bfloat16()
    pr = torch.einsum("shd,slhd->shl", q.float(), k.float()).mul(hd ** -0.5).softmax(-1).mean(1)
    got = ops.pool_attn_map(q.to(DEV), k.to(DEV), skip=1)
    assert rel(got, pr[:, 1:]) < 1e-5
    got0 = ops.pool_attn_map(q.to(DEV), k.to(DEV), skip=0)
    assert rel(got0, pr) < 1e-5 and torch.allclose(got0.sum(1).cpu(), torch.ones(S), atol=1e-5)


def test_gather_rows_is_a_bit_exact_boolean_mask_gather():
    """`t[~mask].reshape(K, B, -1, C)` (engines/engine_for_pretraining.py:118-125) for bf16 / fp32 / fp16 rows, with and without cls."""
    gen = torch.Generator().manual_seed(0)
    B, N, K = 3, 40, 2
    mask = torch.ones(B, N, dtype=torch.bool)
    for b in range(B):
        mask[b, torch.randperm(N, generator=gen)[:11]] = False
    mask = masking.with_cls_column(mask)                                   # (B, 1+N), 12 kept per row
    vis, inv = M.build_gather_indices(mask, DEV)
    for dtype, C in ((torch.bfloat16, 3200), (torch.float32, 100), (torch.float16, 72)):
        t = torch.randn(K, B, N + 1, C, generator=gen).to(dtype)
        want = t[~mask.unsqueeze(0).repeat(K, 1, 1)].reshape(K, B, -1, C)
        got = masking.gather_visible(t.to(DEV), mask=mask.to(DEV))
        assert got.dtype == dtype and torch.equal(got.cpu(), want)
        got1 = masking.gather_visible(t[0].to(DEV), vis_idx=vis)             # (B, 1+N, C) input
        assert torch.equal(got1.cpu(), want[0])
        tm = t[:, :, 1:].contiguous()                                       # MAE flavour: no cls row, mask[:, 1:]
        want_m = tm[~mask[:, 1:].unsqueeze(0).repeat(K, 1, 1)].reshape(K, B, -1, C)
        got_m = masking.gather_visible(tm.to(DEV), vis_idx=vis, drop_cls=True)
        assert torch.equal(got_m.cpu(), want_m)
    with pytest.raises(Exception):
        ops.gather_rows(torch.zeros(2, 5, 3, dtype=torch.bfloat16, device=DEV), vis[:2])      # 6-byte rows: not a multiple of 16


def test_attention_guided_mask_on_device_feeds_the_student():
    """engine_for_pretraining.py:105-116 on the device: multinomial draw -> mask -> the student's index compaction accepts it."""
    B, T, N = 2, 4, 16
    attn = torch.rand(B * T, N, device=DEV) + 1e-3
    m = masking.attention_guided_mask(attn, B, 0.75)
    assert m.is_cuda and m.shape == (B, 1 + T * N) and (~m).sum(1).tolist() == [1 + T * 4] * B and not m[:, 0].any()
    vis, inv = M.build_gather_indices(m, DEV)
    want = O.visible_indices(m.cpu().numpy())
    assert np.array_equal(vis.cpu().numpy(), want)


def test_stage1_distiller_end_to_end_step():
    """engines/engine_for_pretraining.py:63-148 on the device: both frozen teachers -> attention-guided mask -> visible targets ->
    student step through the native engine.  Checks the wiring (shapes, mask counts, bit-exact target gathers, the step's loss ==
    the student's fused loss on the same mask / targets) with random-weight teachers."""
    from internvideo_amd.engine import IVTrainEngine
    from internvideo_amd.internvl_clip_vision import InternVL_CLIP
    from internvideo_amd.stage1 import Stage1Distiller
    from internvideo_amd.videomae_teacher import VisionTransformer
    torch.manual_seed(0)
    B, T16, img, p = 2, 8, 56, 14                                      # 8 loaded frames -> 4 for the student / CLIP teacher
    student = M.PretrainInternVideo2(img_size=img, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4.0, num_frames=4, drop_path_rate=0.0,
                                     attn_pool_num_heads=2, clip_embed_dim=64, clip_teacher_embed_dim=256, clip_teacher_final_dim=64,
                                     clip_return_layer=2, mae_teacher_embed_dim=96, mae_return_layer=2, init_values=1.0).to(DEV).train()
    clip_t = InternVL_CLIP(img_size=img, embed_dim=256, num_heads=2, depth=3, mlp_ratio=4, attn_pool_num_heads=4, clip_embed_dim=64,
                           clip_return_layer=2).to(DEV).eval()
    mae_t = VisionTransformer(img_size=img, patch_size=p, embed_dim=96, depth=3, num_heads=4, mlp_ratio=4, qkv_bias=True, all_frames=T16,
                              tubelet_size=2, mae_return_layer=2).to(DEV).eval()
    # the default table is sized for 16-frame 224^2 clips: give the tiny teacher a table of its own size
    mae_t.pos_embed = torch.nn.Parameter(torch.randn(1, (T16 // 2) * 16, 96, device=DEV) * 0.02)
    eng = IVTrainEngine(student, lr=1e-3)
    gen = torch.Generator(device=DEV).manual_seed(5)
    dist = Stage1Distiller(eng, clip_t, mae_t, mask_type="attention", mask_ratio=0.75, td_ratio=2, generator=gen)
    videos = torch.rand(B, 3, T16, img, img, device=DEV)
    clip_videos, mask, targets, (vis, inv) = dist.teacher_targets(videos)
    assert tuple(clip_videos.shape) == (B, 3, 4, img, img) and torch.equal(clip_videos, videos[:, :, ::2])
    n_vis = 16 - int(16 * 0.75)
    L = 1 + 4 * n_vis
    assert mask.shape == (B, 1 + 4 * 16) and (~mask).sum(1).tolist() == [L] * B
    tg_clip, tg_final, tg_mae = targets
    assert tuple(tg_clip.shape) == (2, B, L, 256) and tuple(tg_final.shape) == (B, 64) and tuple(tg_mae.shape) == (2, B, L - 1, 96)
    # the gathers are the reference's boolean-mask indexing (engine_for_pretraining.py:118-125), bit for bit
    gen2 = torch.Generator(device=DEV).manual_seed(5)
    z, xf, attn = clip_t(clip_videos)
    assert torch.equal(masking.attention_guided_mask(attn, B, 0.75, generator=gen2), mask)
    assert torch.equal(tg_clip, z[~mask.unsqueeze(0).repeat(2, 1, 1)].reshape(2, B, -1, 256))
    zm = mae_t(videos)
    assert torch.equal(tg_mae, zm[~mask[:, 1:].unsqueeze(0).repeat(2, 1, 1)].reshape(2, B, -1, 96))
    # one step; its loss is the fused student loss on exactly these inputs
    with torch.no_grad():
        want, _ = student.forward_loss(clip_videos, mask, targets, vis_inv=(vis, inv))
    gen.manual_seed(5)
    before = eng.master.clone()
    loss, parts = dist.step(videos)
    assert abs(loss.item() - want.item()) / abs(want.item()) < 1e-5 and not torch.equal(before, eng.master)
    assert 0 < loss.item() < 12.0 and len(parts) == 3            # three cosine losses, each in [0, 4]


def test_pool_attention_wide_heads_kernels_vs_torch():
    """hd = 200 (the 6B models' attention-pool projector: 16 heads over 3200): forward, lse-free backward vs torch fp32"""
    gen = torch.Generator().manual_seed(2)
    S, L, H, hd = 3, 417, 2, 200
    q = (torch.randn(S, H, hd, generator=gen) * 0.5).bfloat16()
    kv = (torch.randn(2, S, L, H, hd, generator=gen) * 0.5).bfloat16()
    qr, kr, vr = q.float().requires_grad_(True), kv[0].float().requires_grad_(True), kv[1].float().requires_grad_(True)
    p = torch.einsum("shd,slhd->shl", qr, kr).mul(hd ** -0.5).softmax(-1)
    ref = torch.einsum("shl,slhd->shd", p, vr)
    w = torch.randn(S, H, hd, generator=gen).bfloat16()
    (ref * w.float()).sum().backward()
    kd = kv.to(DEV)
    o, lse = ops.pool_attn_fwd(q.to(DEV), kd[0], kd[1])
    assert rel(o.float(), ref.detach()) < 6e-3
    dq, dkv = ops.pool_attn_bwd(q.to(DEV), kd[0], kd[1], w.to(DEV), lse)
    assert rel(dq.float(), qr.grad) < 1e-2 and rel(dkv[0].float(), kr.grad) < 1e-2 and rel(dkv[1].float(), vr.grad) < 1e-2


def test_student_with_wide_pool_heads_matches_oracle():
    """a student whose attention-pool heads are 200 wide (like pretrain_internvideo2_6B_patch14_224) vs the CPU oracle, fwd + bwd"""
    cfg = O.StudentConfig(img_size=56, embed_dim=400, depth=2, num_heads=5, mlp_ratio=4.0, num_frames=4, attn_pool_num_heads=2,
                          clip_embed_dim=64, clip_teacher_embed_dim=96, clip_teacher_final_dim=64, clip_return_layer=1, mae_teacher_embed_dim=96,
                          mae_return_layer=1)
    params = O.synthetic_params(cfg, seed=11)
    video, mask, targets = O.synthetic_batch(cfg, 2, 5, seed=11)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref_out = O.student_forward(p, video, mask, cfg)
    ref_loss, _ = O.distill_losses(ref_out, targets)
    ref_loss.backward()
    m = M.PretrainInternVideo2(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
                               num_frames=cfg.num_frames, drop_path_rate=0.0, attn_pool_num_heads=2, clip_embed_dim=64, clip_teacher_embed_dim=96,
                               clip_teacher_final_dim=64, clip_return_layer=1, mae_teacher_embed_dim=96, mae_return_layer=1)
    m.load_state_dict(params, strict=True)
    m = m.to(DEV).train()
    tg = tuple(t.to(DEV) for t in targets)
    loss, _ = m.forward_loss(video.to(DEV), torch.from_numpy(mask), tg)
    assert abs(loss.item() - ref_loss.item()) / abs(ref_loss.item()) < 1e-3
    loss.backward()
    for k in ("clip_projector.cross_attn.q_bias", "clip_projector.cross_attn.v_bias", "clip_projector.norm1_q.weight", "clip_projector.cross_attn.proj.bias",
              "clip_projector.cross_attn.q.weight", "clip_projector.cross_attn.v.weight", "blocks.1.mlp.fc2.bias", "final_clip_decoder.head.bias"):
        assert rel(dict(m.named_parameters())[k].grad, p[k].grad) < 5e-2, k


def test_finetune_classifier_matches_reference_golden():
    """fine-tuning classifier (internvideo_amd.internvideo2) at full sequence length vs the reference's InternVideo2: logits,
    cross-entropy, gradients; 10 classes -> the head GEMM runs zero-padded to 16 columns."""
    from internvideo_amd import internvideo2 as FT
    g = np.load(GOLD)
    cfg = O.named_config("tiny88")
    params = O.synthetic_finetune_params(cfg, 10, seed=12)
    video, _, _ = O.synthetic_batch(cfg, 2, 5, seed=12)
    m = FT.InternVideo2(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
                        num_frames=cfg.num_frames, drop_path_rate=0.0, attn_pool_num_heads=cfg.attn_pool_num_heads,
                        clip_embed_dim=cfg.clip_embed_dim, num_classes=10)
    m.load_state_dict(params, strict=True)
    m = m.to(DEV).train()
    logits = m(video.to(DEV))
    assert tuple(logits.shape) == (2, 10) and rel(logits.float(), g["ft:logits"]) < 1e-2
    loss = torch.nn.functional.cross_entropy(logits.float(), torch.tensor([3, 7], device=DEV))
    assert abs(loss.item() - g["ft:loss"][0]) / g["ft:loss"][0] < 1e-3
    loss.backward()
    assert check_grads(g, "ft:", m, "ft:bf16err:") >= 9


def test_finetune_classifier_at_full_sequence_length_and_real_width_matches_the_reference_digest():
    """The fine-tuning classifier at the 1B width on the UN-MASKED sequence (1408 wide, 16 heads of 88, 8 x 224^2 -> L = 2049: 33 key tiles per
    attention head; 400 classes; depth 4) against a digest of the REFERENCE's own InternVideo2 forward + cross-entropy + backward at that size
    (tests/golden/finetune_fullwidth_digest.npz, make_golden_finetune_fullwidth.py): logits 1e-2, loss 1e-3, gradient norms 3e-2, corners 5e-2."""
    from internvideo_amd import internvideo2 as FT
    from tests.test_flavours_oracle import _finetune_fullwidth
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "finetune_fullwidth_digest.npz"))
    classes, seed, label = (int(x) for x in g["meta"])
    cfg = _finetune_fullwidth()
    params = O.synthetic_finetune_params(cfg, classes, seed=seed)
    video, _, _ = O.synthetic_batch(cfg, 1, 52, seed=seed)
    m = FT.InternVideo2(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
                        num_frames=cfg.num_frames, drop_path_rate=0.0, attn_pool_num_heads=cfg.attn_pool_num_heads,
                        clip_embed_dim=cfg.clip_embed_dim, num_classes=classes)
    m.load_state_dict(params, strict=True)
    m = m.to(DEV).train()
    logits = m(video.to(DEV))
    assert tuple(logits.shape) == (1, classes) and rel(logits.float(), g["logits"]) < 1e-2, rel(logits.float(), g["logits"])
    loss = torch.nn.functional.cross_entropy(logits.float(), torch.tensor([label], device=DEV))
    assert abs(loss.item() - float(g["loss"][0])) / float(g["loss"][0]) < 1e-3
    loss.backward()
    sd = dict(m.named_parameters())
    worst = {}
    for key in g.files:
        if key.startswith("grad:") and key.endswith(":norm"):
            k = key[5:-5]
            gr = sd[k].grad
            g2 = gr.reshape(-1, gr.shape[-1])
            worst["norm:" + k] = abs(gr.double().norm().item() - float(g[key][0])) / float(g[key][0])
            worst["corner:" + k] = rel(g2[:16, :16].float(), g["grad:" + k + ":corner"])
    # in front of the 1-query attention pool the bound is 8e-2 as in the full-size student test: a softmax Jacobian of ONE mean query over
    # all 2049 tokens (measured 5.3e-2 on the 16 x 16 corner of cross_attn.k.weight; its norm is inside 3e-2)
    pool_front = ("clip_projector.norm1_", "clip_projector.cross_attn.q", "clip_projector.cross_attn.k")
    bad = {k: v for k, v in worst.items() if v > (8e-2 if k.split(":", 1)[1].startswith(pool_front) else (5e-2 if k.startswith("corner:") else 3e-2))}
    assert len(worst) >= 12 and not bad, bad


def test_stage2_heads_uta_and_vtc_losses_match_oracle():
    """`Stage2VisionTextHeads` (vision_proj / text_proj / clamped temperature / UTA + VTC losses of
    multi_modality/models/internvideo2_stage2_visual.py:103-120) vs the oracle's criterions restatement (pinned to the reference's
    get_sim / vtc_loss in tests/test_oracle_golden.py), forward and backward."""
    from internvideo_amd.stage2 import Stage2VisionTextHeads, new_UTA_Loss
    gen = torch.Generator().manual_seed(4)
    B, K, N, C = 24, 2, 9, 96
    heads = Stage2VisionTextHeads(vision_width=64, text_width=80, embed_dim=48, temp=0.9).to(DEV)      # 0.9 -> clamped to 0.5
    pv = torch.randn(B, 64, generator=gen); pt = torch.randn(B, 80, generator=gen)
    idx = torch.tensor([0, 1, 2, 3, 3, 5, 6, 7, 8, 9, 1, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 0])
    so = torch.nn.functional.normalize(torch.randn(K, B, N, C, generator=gen), dim=-1)
    tg = torch.nn.functional.normalize(torch.randn(K, B, N, C, generator=gen), dim=-1)
    sf = torch.nn.functional.normalize(torch.randn(B, 32, generator=gen), dim=-1)
    tf = torch.nn.functional.normalize(torch.randn(B, 32, generator=gen), dim=-1)
    # reference arithmetic in fp32 on the CPU
    W = {k: v.detach().cpu().float().requires_grad_(True) for k, v in heads.named_parameters()}
    so_r, sf_r = so.clone().requires_grad_(True), sf.clone().requires_grad_(True)
    temp_c = W["temp"].clamp(0.001, 0.5)
    v = pv @ W["vision_proj.weight"].t() + W["vision_proj.bias"]
    t = pt @ W["text_proj.weight"].t() + W["text_proj.bias"]
    want_vtc = O.vtc_loss(v, t, idx, temp_c)
    want_uta = (2 - 2 * (so_r * tg).sum(-1)).mean() + (2 - 2 * (sf_r * tf).sum(-1)).mean()
    (want_vtc + want_uta).backward()
    so_g, sf_g = so.to(DEV).bfloat16().requires_grad_(True), sf.to(DEV).bfloat16().requires_grad_(True)
    out = heads(pv.to(DEV).bfloat16(), pt.to(DEV).bfloat16(), idx.to(DEV), so_g, sf_g, tg.to(DEV), tf.to(DEV))
    assert abs(heads.temp.item() - 0.5) < 1e-7                                               # clamped in place (:291-294)
    assert abs(out["loss_vtc"].item() - want_vtc.item()) / abs(want_vtc.item()) < 2e-2           # bf16 inputs to the projections
    assert abs(out["loss_uta"].item() - want_uta.item()) / abs(want_uta.item()) < 2e-3
    (out["loss_vtc"] + out["loss_uta"]).backward()
    assert rel(heads.vision_proj.weight.grad, W["vision_proj.weight"].grad) < 5e-2
    assert rel(heads.text_proj.bias.grad, W["text_proj.bias"].grad) < 5e-2
    assert rel(so_g.grad.float(), so_r.grad) < 1e-2 and rel(sf_g.grad.float(), sf_r.grad) < 1e-2
    # final features not distilled -> zeros (criterions.py:483-484)
    l = new_UTA_Loss(distill_final_features=False).uta_loss(so_g.detach(), sf_g.detach(), tg.to(DEV), tf.to(DEV))
    assert abs(l.item() - (2 - 2 * (so * tg).sum(-1)).mean().item()) < 5e-3


def test_teachers_split_large_batches_into_passes():
    """the frozen teachers run clip groups back to back when one pass would exceed the GEMM's 2 GiB operand limit: same outputs"""
    from internvideo_amd.internvl_clip_vision import InternVL_CLIP
    from internvideo_amd.videomae_teacher import VisionTransformer
    torch.manual_seed(1)
    clip_t = InternVL_CLIP(img_size=56, embed_dim=128, num_heads=2, depth=2, mlp_ratio=4, attn_pool_num_heads=2, clip_embed_dim=64,
                           clip_return_layer=2).to(DEV).eval()
    v = torch.rand(5, 3, 4, 56, 56, device=DEV)
    want = clip_t(v)
    clip_t._clips_per_pass = lambda T: 2                                    # force 3 passes (2 + 2 + 1 clips)
    got = clip_t(v)
    assert all(torch.equal(a, b) for a, b in zip(want, got)) and tuple(got[0].shape) == (2, 5, 65, 128) and tuple(got[2].shape) == (20, 16)
    mae_t = VisionTransformer(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True, all_frames=16,
                              tubelet_size=2, mae_return_layer=2).to(DEV).eval()
    x = torch.rand(3, 3, 16, 32, 32, device=DEV)
    want = mae_t(x)
    import internvideo_amd.videomae_teacher as VT
    orig = VT.VisionTransformer.forward
    try:
        def small(self, x, mask=None):                                               # same code path with a 1-clip pass size
            self._bf16_weights()
            return torch.cat([self._forward_pass(x[b:b + 1], None if mask is None else mask[b:b + 1]) for b in range(x.shape[0])], dim=1)
        VT.VisionTransformer.forward = torch.no_grad()(small)
        got = mae_t(x)
    finally:
        VT.VisionTransformer.forward = orig
    assert torch.equal(want, got)


@pytest.mark.parametrize("zero1_shards", [0, 3])
def test_layer_wise_lr_decay_in_the_fused_adamw_matches_the_reference_groups(zero1_shards):
    """VERDICT r3 item 7 / SURVEY 8(f) row 4.  tests/golden/layer_decay.npz = torch.optim.AdamW over the groups the REFERENCE's
    optim_factory.get_parameter_groups + LayerDecayValueAssigner built for this classifier (lr * lr_scale per group, three steps of seeded
    gradients).  Here: IVTrainEngine(layer_decay=0.75) -- one flat buffer per region, the (segment end, scale) table consumed inside
    ivh_adamw_step_scaled.  zero1_shards > 0 additionally updates the matrix region in that many separate slices with seg_base offsets,
    the way the ZeRO-1 path launches one AdamW per bucket shard."""
    from internvideo_amd import internvideo2 as FT, ops
    from internvideo_amd.engine import IVTrainEngine
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "layer_decay.npz"))
    layer_decay, lr, wd, steps, seed = (float(x) for x in g["meta"])
    cfg = O.named_config("tiny88")
    m = FT.InternVideo2(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
                        num_frames=cfg.num_frames, drop_path_rate=0.0, attn_pool_num_heads=cfg.attn_pool_num_heads,
                        clip_embed_dim=cfg.clip_embed_dim, num_classes=10)
    m.load_state_dict(O.synthetic_finetune_params(cfg, 10, seed=12), strict=True)
    m = m.to(DEV)
    eng = IVTrainEngine(m, lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, max_grad_norm=0.0, layer_decay=layer_decay)
    assert eng._lr_seg_mat is not None and eng._lr_seg_mat[0].numel() == cfg.depth + 2
    mat_names = {n for n, _ in eng.mat_params}
    for step in range(1, int(steps) + 1):
        eng.zero_grad()
        for n, p in m.named_parameters():
            gen = torch.Generator().manual_seed(int(seed) * 1000 + step * 131 + (sum(map(ord, n)) % 997))
            gr = torch.randn(p.shape, generator=gen) * 0.05
            p.main_grad.copy_(gr.to(torch.bfloat16) if n in mat_names else gr)       # the generator rounded matrix gradients to bf16 too
        if zero1_shards:
            eng.step_count += 1
            cuts = [0] + [eng.n_mat * k // zero1_shards // 64 * 64 for k in range(1, zero1_shards)] + [eng.n_mat]
            for s0, s1 in zip(cuts, cuts[1:]):
                ops.adamw_step(eng.master[s0:s1], eng.exp_avg[s0:s1], eng.exp_avg_sq[s0:s1], eng.grad_mat[s0:s1], eng.shadow[s0:s1], lr, 0.9, 0.999,
                               1e-8, wd, eng.step_count, 1.0, None, lr_segments=eng._lr_seg_mat, seg_base=s0)
            ops.adamw_step(eng.master[eng.n_mat:], eng.exp_avg[eng.n_mat:], eng.exp_avg_sq[eng.n_mat:], eng.grad_vec, None, lr, 0.9, 0.999, 1e-8, 0.0,
                           eng.step_count, 1.0, None, lr_segments=eng._lr_seg_vec)
        else:
            eng.optimizer_step()
    torch.cuda.synchronize()
    worst = 0.0
    for n, p in m.named_parameters():
        flat = p.detach().float().reshape(-1).cpu()
        ref = torch.from_numpy(g["final:" + n])
        got = flat if flat.numel() <= 2048 else torch.cat([flat[:512], flat[-512:]])
        # three AdamW steps of |update| ~ lr * scale each: an element updated with the wrong scale is off by >= 25 % of that
        tol = 2e-2 * lr * eng.lr_scale_of(n)
        assert (got - ref).abs().max().item() < tol, (n, (got - ref).abs().max().item(), tol)
        worst = max(worst, abs(float(flat.double().norm()) - float(g["norm:" + n])) / max(float(g["norm:" + n]), 1e-12))
    assert worst < 1e-4, worst
    assert torch.equal(eng.shadow, eng.master[:eng.n_mat].to(torch.bfloat16))


def test_sep_pos_embed_in_the_distill_and_finetune_models_matches_reference_golden():
    """`sep_pos_embed=True` in DistInternVideo2 (internvideo2_distill.py:481-494, 551-563, 622-637, 677-692) and in the fine-tuning classifier
    (internvideo2.py:390-397, 454-465, 510-525) -- refused by the round-5 mirrors (VERDICT r5 missing 4): same state_dict keys, outputs, loss and
    the gradient of every separable table against the reference's own CPU run (tests/golden/sep_pos.npz, make_golden_sep_pos.py)."""
    from internvideo_amd import internvideo2 as FT
    g = np.load(os.path.join(os.path.dirname(GOLD), "sep_pos.npz"))
    # ---- distillation student
    cfg = O.named_config("dist64")
    params = dict(O.synthetic_params(cfg, seed=2))
    sep = ["pos_embed_spatial", "pos_embed_temporal", "pos_embed_cls", "clip_pos_embed_spatial", "clip_pos_embed_temporal", "clip_pos_embed_cls"]
    for k in ("pos_embed", "clip_pos_embed"):
        params.pop(k)
    for k in sep:
        params[k] = torch.from_numpy(g["dist:in:" + k])
    video, mask, targets = O.synthetic_batch(cfg, 2, 4, seed=2)
    m = D.DistInternVideo2(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
                           num_frames=cfg.num_frames, drop_path_rate=0.0, attn_pool_num_heads=cfg.attn_pool_num_heads,
                           clip_embed_dim=cfg.clip_embed_dim, clip_teacher_embed_dim=cfg.clip_teacher_embed_dim,
                           clip_teacher_final_dim=cfg.clip_teacher_final_dim, clip_return_layer=cfg.clip_return_layer,
                           clip_student_return_index=list(cfg.clip_return_index_override),
                           clip_student_decoder={"linear": "Linear_Decoder", "mlp": "MLP_Decoder"}[cfg.clip_decoder_kind], sep_pos_embed=True)
    assert "pos_embed" not in m.state_dict() and set(sep) <= set(m.state_dict()) and set(sep) <= m.no_weight_decay()
    m.load_state_dict(params, strict=True)
    m = m.to(DEV).train()
    oc, of = m(video.to(DEV), torch.from_numpy(mask))
    assert max(rel(oc.float(), g["dist:x_clip_align"]), rel(of.float(), g["dist:x_align"])) < 1e-2
    tc, tf = targets[0].to(DEV), targets[1].to(DEV)
    loss = (2 - 2 * (oc.float() * tc).sum(-1)).mean() + (2 - 2 * (of.float() * tf).sum(-1)).mean()
    assert abs(loss.item() - g["dist:losses"][0]) < 1e-3 * g["dist:losses"][0]
    loss.backward()
    named = dict(m.named_parameters())
    worst = {k[10:]: rel(named[k[10:]].grad, g[k]) for k in g.files if k.startswith("dist:grad:")}
    assert set(sep) <= set(worst)
    assert not {k: v for k, v in worst.items() if v > 4e-2}, worst
    # ---- fine-tuning classifier
    cfg = O.named_config("tiny88")
    params = dict(O.synthetic_finetune_params(cfg, 10, seed=12))
    fsep = ["pos_embed_spatial", "pos_embed_temporal", "pos_embed_cls"]
    params.pop("pos_embed")
    for k in fsep:
        params[k] = torch.from_numpy(g["ft:in:" + k])
    video, _, _ = O.synthetic_batch(cfg, 2, 5, seed=12)
    m = FT.InternVideo2(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
                        num_frames=cfg.num_frames, drop_path_rate=0.0, attn_pool_num_heads=cfg.attn_pool_num_heads,
                        clip_embed_dim=cfg.clip_embed_dim, num_classes=10, sep_pos_embed=True)
    assert "pos_embed" not in m.state_dict() and set(fsep) <= set(m.state_dict())
    m.load_state_dict(params, strict=True)
    m = m.to(DEV).train()
    logits = m(video.to(DEV))
    assert rel(logits.float(), g["ft:logits"]) < 1e-2
    loss = torch.nn.functional.cross_entropy(logits.float(), torch.tensor([3, 7], device=DEV))
    assert abs(loss.item() - g["ft:loss"][0]) < 1e-3 * g["ft:loss"][0]
    loss.backward()
    named = dict(m.named_parameters())
    worst = {k[8:]: rel(named[k[8:]].grad, g[k]) for k in g.files if k.startswith("ft:grad:")}
    assert set(fsep) <= set(worst)
    assert not {k: v for k, v in worst.items() if v > 5e-2}, worst


def _iv2_teacher():
    from internvideo_amd import internvideo2_teacher as T2
    fix = json.load(open(os.path.join(os.path.dirname(GOLD), "distill_protocol.json")))
    kw = fix["teacher"]
    cfg_t = O.StudentConfig(clip_teacher_embed_dim=96, clip_teacher_final_dim=64, clip_return_layer=2, has_mae=False, **kw)
    m = T2.InternVideo2(drop_path_rate=0.0, clip_norm_type='l2', return_attn=True, clip_return_layer=2, **kw)
    params = O.synthetic_params(cfg_t, seed=fix["teacher_param_seed"])
    sd = m.state_dict()
    m.load_state_dict({k: v for k, v in params.items() if k in sd}, strict=True)
    return fix, m.to(DEV).eval()


def test_internvideo2_teacher_matches_the_reference_and_returns_a_clip_level_attention_map():
    """internvideo2_teacher.InternVideo2 (the class behind `teacher_internvideo2_stage2_1B`, scripts/distillation/B14_dist_1B_stage2.sh:26)
    against the reference's own module (tests/golden/distill_protocol.npz: its three outputs on the two batches of the recorded distillation
    loop): normalised taps, pooled clip token, and the (B, T*H*W) attention map of the pooling query."""
    fix, teacher = _iv2_teacher()
    g = np.load(os.path.join(os.path.dirname(GOLD), "distill_protocol.npz"))
    cfg = O.named_config(fix["config"])
    T, h, w = cfg.grid
    gv = torch.Generator().manual_seed(fix["video_seed"])
    for i in range(fix["steps"]):
        video = torch.rand(fix["batch"], 3, T, cfg.img_size, cfg.img_size, generator=gv)
        z, x, attn = teacher(video.to(DEV))
        assert tuple(z.shape) == g[f"z:{i}"].shape and tuple(attn.shape) == (fix["batch"], T * h * w)
        assert rel(z.float(), g[f"z:{i}"]) < 1e-2 and rel(x.float(), g[f"x:{i}"]) < 1e-2 and rel(attn, g[f"attn:{i}"]) < 2e-2
        assert abs(float(attn.sum(1).mean()) - float(g[f"attn:{i}"].sum(1).mean())) < 1e-2


def test_reference_distillation_loop_with_the_internvideo2_teacher_on_the_hip_path():
    """tests/golden/distill_protocol.json: the reference's engine_for_distill.train_one_epoch (:20-199) around its own teacher and student for two
    steps.  Replayed on the HIP path: teacher mirror -> the loop's own masks (clip-level draws: frames keep different counts) -> targets gathered
    by the HIP gather kernel -> ds_compat engine around the HIP DistInternVideo2 -> the loss as the loop builds it (:107-118) -> backward, step.
    Losses against the reference trajectory: step 1 (same weights) 2e-3 incl. the bf16 teacher; step 2 within 3 x the REFERENCE's own
    bf16-vs-fp32 deviation there (recorded in the fixture: the first AdamW step moves every weight by ~lr sign(g), and gradient elements at the
    noise floor take either sign -- the reference's bf16 and fp32 runs differ by 6.6e-3 at step 2); gradient norms 4 %.  And
    Stage1Distiller drives the same teacher end to end (its kept-token count comes from the map's per-clip layout)."""
    from types import SimpleNamespace
    from internvideo_amd import ds_compat, masking
    from internvideo_amd.engine import IVTrainEngine
    from internvideo_amd.stage1 import Stage1Distiller
    fix, teacher = _iv2_teacher()
    cfg = O.named_config(fix["config"])
    B = fix["batch"]
    T, h, w = cfg.grid
    params = O.synthetic_params(cfg, seed=fix["param_seed"])
    args = SimpleNamespace(lr=fix["lr"], weight_decay=fix["weight_decay"], opt_betas=fix["betas"], opt_eps=fix["eps"], clip_grad=fix["clip"], update_freq=1)
    model, optimizer, _, _ = ds_compat.initialize(args=args, model=build_dist(cfg, params), model_parameters=None, dist_init_required=False)
    gv = torch.Generator().manual_seed(fix["video_seed"])
    loader = [torch.rand(B, 3, T, cfg.img_size, cfg.img_size, generator=gv) for _ in range(fix["steps"])]
    calls = [e for e in fix["trace"] if e["call"] == "model.__call__"]
    want = [e["loss"] for e in fix["trace"] if e["call"] == "model.backward"]
    want_gn = [e["grad_norm"] for e in fix["trace"] if e["call"] == "model.step"]
    model.train(); model.zero_grad(); model.micro_steps = 0
    got, gn = [], []
    for it, e in enumerate(calls):
        for group in optimizer.param_groups:
            group["lr"] = fix["lr_schedule"][it] * group["lr_scale"]
            if group["weight_decay"] > 0:
                group["weight_decay"] = fix["wd_schedule"][it]
        videos = loader[it].to(DEV)
        z, x, attn = teacher(videos)                                                     # DE:81-85
        assert tuple(attn.shape) == (B, T * h * w)
        mask = torch.from_numpy(np.unpackbits(np.array(e["mask"]["packed"], dtype=np.uint8), axis=1)[:, :e["mask"]["shape"][1]].astype(bool)).to(DEV)
        tg_mid = masking.gather_visible(z, mask)                                         # DE:100-103
        oc, of = model(videos.bfloat16(), mask)                                          # DE:107
        assert [list(o.shape) for o in (oc, of)] == [o["shape"] for o in e["outputs"]]
        loss = (2 - 2 * (oc * tg_mid).sum(dim=-1)).mean() + (2 - 2 * (of * x).sum(dim=-1)).mean()
        model.backward(loss); model.step()
        got.append(loss.item()); gn.append(float(model.optimizer._global_grad_norm))
    rel_l = [abs(a - b) / abs(b) for a, b in zip(got, want)]
    rel_g = [abs(a - b) / abs(b) for a, b in zip(gn, want_gn)]
    print("distill loop replay: loss dev", rel_l, "grad-norm dev", rel_g)
    bars = [max(2e-3, 3.0 * d) for d in fix["reference_bf16_vs_fp32_loss_dev"]]
    assert rel_l[0] < 2e-3 and all(r < b for r, b in zip(rel_l, bars)), (rel_l, bars)
    assert max(rel_g) < 4e-2, rel_g
    # the native distiller with the same teacher: one step, finite, per-clip kept count 17
    eng = IVTrainEngine(build_dist(cfg, params), lr=1e-3, max_grad_norm=3.0)
    dst = Stage1Distiller(eng, teacher, None, mask_type="attention", mask_ratio=fix["mask_ratio"], td_ratio=1)
    clip_videos, mask, targets, (vis_idx, _) = dst.teacher_targets(loader[0].to(DEV))
    assert tuple(vis_idx.shape) == (B, 17) and bool((~mask).sum(1).eq(17).all()) and tuple(targets[0].shape) == (2, B, 17, 96)
    loss, _ = dst.step(loader[0].to(DEV).bfloat16())
    assert torch.isfinite(loss).item() and 2.0 < loss.item() < 4.5
