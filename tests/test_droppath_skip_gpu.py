"""DropPath sample skipping (include/internvideo_hip.h "DropPath SAMPLE SKIPPING", functional.BlockStackFn): the samples a branch's
DropPath draw zeroes (reference: timm DropPath in Block.forward, InternVideo2/single_modality/models/internvideo2_pretrain.py:264,274,283-286)
are not computed.  Every kernel that takes a device-side count / a keep map is checked against the SAME kernel run on the kept rows only, with
NaN poison in everything it must not read and sentinels in everything it must not write; the block stack is checked against the
compute-then-multiply-by-zero path and against the reference's own run on its own draws.  Runs only on a real MI355X (`-m gpu`)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from internvideo_amd import functional as Fn  # noqa: E402
from internvideo_amd import internvideo2_pretrain as M  # noqa: E402
from internvideo_amd import lib as L_  # noqa: E402
from internvideo_amd import ops  # noqa: E402
from oracle import internvideo2_oracle as O  # noqa: E402

DEV = "cuda"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BF16, F32 = torch.bfloat16, torch.float32


def rel(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def randn(*shape, seed=0, scale=1.0, dtype=BF16):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV).to(dtype)


def cnt(v):
    return torch.tensor([v], dtype=torch.int32, device=DEV)


@pytest.fixture
def plain_tiles():
    """whole 256 x 256 tiles only (no K split of the tail round, no half-width tiles): the full-size launch then sums in the order of the
    device-count launch, and forward / dgrad results can be compared BITWISE"""
    lib = L_.load()
    lib.ivh_gemm256_debug_split(0); lib.ivh_gemm256_debug_half(0)
    ops.set_gemm_kernel(2)
    yield
    ops.set_gemm_kernel(0)
    lib.ivh_gemm256_debug_split(1); lib.ivh_gemm256_debug_half(1)


def test_abi_version_and_plan():
    assert L_.load().ivh_abi_version() == 2
    rng = np.random.default_rng(0)
    for B, Lr in ((1, 7), (5, 33), (64, 417), (128, 417), (200, 9)):
        rs = (rng.random((6, 2, B)) > 0.35).astype(np.float32) * 1.25
        rs[0, 0] = 1.0                                                       # nothing dropped
        rs[1, 1] = 0.0                                                       # everything dropped
        slot, count = ops.droppath_plan(torch.from_numpy(rs).to(DEV), Lr)
        slot, count = slot.cpu().numpy(), count.cpu().numpy()
        keep = rs != 0
        exp = np.where(keep, np.cumsum(keep, axis=-1) - 1, -1)
        assert np.array_equal(slot, exp)
        assert np.array_equal(count[..., 0], keep.sum(-1)) and np.array_equal(count[..., 1], keep.sum(-1) * Lr)


# ---- GEMM with device-side row counts --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(1251, 1408, 1408), (834, 4224, 1408), (2085, 6144, 1408), (1668, 1408, 6144), (417 * 128, 1408, 1408)])
@pytest.mark.parametrize("frac", [0.0, 0.31, 0.875, 1.0])
def test_gemm_forward_and_dgrad_follow_m_dev(M, N, K, frac, plain_tiles):
    """forward NT (bias), forward EPI 2 (GELU + gelu' copy), dgrad (rows-contiguous weight), dgrad EPI 3 (x gelu', bias-gradient partials): the
    first *m_dev rows equal the launch on those rows alone BITWISE, rows behind them keep their sentinel, NaN rows behind them in A and in the
    gelu' operand are never read (the column sums would show them)."""
    if M > 10000 and frac in (0.0, 0.31):
        pytest.skip("large case: two fractions are enough")
    Mk = int(round(M * frac))
    if 0 < Mk < M:
        Mk = Mk // 417 * 417 + (5 if frac < 0.5 else 0)                     # whole clips, and a ragged count
        Mk = max(1, min(Mk, M - 1))
    md = cnt(Mk)
    a = randn(M, K, seed=1)
    a[Mk:] = float("nan")
    w = randn(N, K, seed=2, scale=K ** -0.5)
    bias = randn(N, seed=3, dtype=F32)
    SENT = 7.0
    # forward, plain epilogue
    out = torch.full((M, N), SENT, dtype=BF16, device=DEV)
    ops.gemm(a, w, bias=bias, out=out, m_dev=md)
    if Mk:
        ref = ops.gemm(a[:Mk].contiguous(), w, bias=bias)
        assert torch.equal(out[:Mk], ref)
    assert bool((out[Mk:] == SENT).all())
    # forward, GELU + gelu' copy
    out = torch.full((M, N), SENT, dtype=BF16, device=DEV)
    g, u = ops.gemm(a, w, bias=bias, act="gelu_erf_d", want_preact=True, out=out, m_dev=md)
    if Mk:
        gr, ur = ops.gemm(a[:Mk].contiguous(), w, bias=bias, act="gelu_erf_d", want_preact=True)
        assert torch.equal(g[:Mk], gr) and torch.equal(u[:Mk], ur)
    assert bool((out[Mk:] == SENT).all())
    # dgrad: dy [M, N] x W [N, K] (rows-contiguous B)
    dy = randn(M, N, seed=4)
    dy[Mk:] = float("nan")
    out = torch.full((M, K), SENT, dtype=BF16, device=DEV)
    ops.gemm(dy, w, a_kc=True, b_kc=False, out=out, m_dev=md)
    if Mk:
        assert torch.equal(out[:Mk], ops.gemm(dy[:Mk].contiguous(), w, a_kc=True, b_kc=False))
    assert bool((out[Mk:] == SENT).all())
    # dgrad x gelu' with the column sums
    du_in = randn(M, K, seed=5)
    du_in[Mk:] = float("nan")
    out = torch.full((M, K), SENT, dtype=BF16, device=DEV)
    dx, part = ops.gemm(dy, w, a_kc=True, b_kc=False, dact_in=du_in, act="gelu_erf_d", want_colsum=True, out=out, m_dev=md)
    assert part is not None
    cs = ops.colsum_finish(part, m_dev=md)
    if Mk:
        dxr, partr = ops.gemm(dy[:Mk].contiguous(), w, a_kc=True, b_kc=False, dact_in=du_in[:Mk].contiguous(), act="gelu_erf_d", want_colsum=True)
        assert torch.equal(dx[:Mk], dxr)
        assert torch.equal(cs, ops.colsum_finish(partr))
    else:
        assert bool((cs == 0).all())
    assert bool((out[Mk:] == SENT).all())
    assert bool(torch.isfinite(cs).all())


@pytest.mark.parametrize("Kmax,frac", [(417 * 6, 0.0), (417 * 6, 0.5), (417 * 6, 1.0), (417 * 32, 0.81), (100, 0.37), (417 * 128, 0.9)])
def test_wgrad_follows_k_dev_single_and_grouped(Kmax, frac):
    """weight gradients dW[n, k] = sum_m dy[m, n] x[m, k] over the first *k_dev token rows: single launch and grouped launch with one count per
    problem (mixed with problems that have none); NaN rows behind the count are never read; a count of 0 writes zeros."""
    shapes = [(1408, 1408), (4224, 1408), (1408, 6144), (6144, 1408)] if Kmax < 20000 else [(1408, 1408), (4224, 1408)]
    probs, refs = [], []
    for j, (N, Kf) in enumerate(shapes):
        kk = int(round(Kmax * frac))
        if 0 < kk < Kmax:
            kk = max(1, kk - 3 * j)                                          # a different (ragged) count per problem
        dy = randn(Kmax, N, seed=10 + j, scale=0.1)
        x = randn(Kmax, Kf, seed=20 + j)
        ref = (dy[:kk].float().t() @ x[:kk].float()) if kk else torch.zeros((N, Kf), device=DEV)
        dy[kk:] = float("nan"); x[kk:] = float("nan")
        out = torch.full((N, Kf), 3.0, dtype=BF16, device=DEV)
        kd = cnt(kk) if j != 2 else None                                     # problem 2 of a group carries no count: all rows
        if kd is None:
            dy = torch.nan_to_num(dy, nan=0.0); x = torch.nan_to_num(x, nan=0.0)
        probs.append((dy, x, out, kd))
        refs.append(ref)
    # single launches
    for (dy, x, out, kd), ref in zip(probs, refs):
        got = ops.gemm(dy, x, a_kc=False, b_kc=False, k_dev=kd)
        assert bool(torch.isfinite(got).all())
        assert rel(got.float(), ref) < 6e-3 or float(ref.abs().max()) == 0.0 and float(got.abs().max()) == 0.0
    # one grouped launch
    ops.gemm_grouped(probs, a_kc=False, b_kc=False)
    for (dy, x, out, kd), ref in zip(probs, refs):
        assert bool(torch.isfinite(out).all())
        if float(ref.abs().max()) == 0.0:
            assert float(out.abs().max()) == 0.0
        else:
            assert rel(out.float(), ref) < 6e-3
        if kd is not None:
            assert torch.equal(out, ops.gemm(dy, x, a_kc=False, b_kc=False, k_dev=kd))     # grouped == single launch (same kernel), bitwise


# ---- row kernels with keep maps ---------------------------------------------------------------------------------------------------------
def _maps(B, Lr, p_b, p_y, seed):
    rng = np.random.default_rng(seed)
    kb = rng.random(B) >= p_b
    ky = rng.random(B) >= p_y
    rs = np.where(kb, 1.0 / max(1.0 - p_b, 1e-3), 0.0).astype(np.float32)
    sb = np.where(kb, np.cumsum(kb) - 1, -1).astype(np.int32)
    sy = np.where(ky, np.cumsum(ky) - 1, -1).astype(np.int32)
    to = lambda a: torch.from_numpy(a).to(DEV)
    return kb, ky, to(rs), to(sb), to(sy)


def _compact(full, keep, Lr):
    """rows of the kept samples moved to the front, NaN behind them"""
    M, D = full.shape
    rows = np.nonzero(np.repeat(keep, Lr))[0]
    out = torch.full_like(full, float("nan"))
    out[:len(rows)] = full[torch.from_numpy(rows).to(full.device)]
    return out, rows


@pytest.mark.parametrize("D", [1408, 384, 3200])
@pytest.mark.parametrize("rt", [BF16, F32])
@pytest.mark.parametrize("p_b,p_y", [(0.3, 0.4), (0.0, 1.0), (1.0, 0.0)])
def test_rmsnorm_add_skip_equals_the_rowscale_path(D, rt, p_b, p_y):
    """forward and backward of the residual kernel with keep maps against the plain kernel on full-size operands whose dropped samples are
    multiplied by rowscale 0 (forward) / carry dy = 0 (backward): stream, compacted y / dbranch rows, rstd of the kept rows and all three
    column sums BITWISE equal -- including the bytes-in-flight kernel of the bf16 stream and its tap-gradient variant."""
    B, Lr = 12, 33
    Mr = B * Lr
    kb, ky, rs, sb, sy = _maps(B, Lr, p_b, p_y, seed=D)
    res = randn(Mr, D, seed=1, dtype=rt)
    branch = randn(Mr, D, seed=2)
    gamma = randn(D, seed=3, dtype=F32)
    w = randn(D, seed=4, dtype=F32) + 1.0
    eps = 1e-6
    r_ref, y_ref, rstd_ref = ops.rmsnorm_add_fwd(res, branch, gamma, rs, Lr, w, eps)
    bc, brows = _compact(branch, kb, Lr)
    r_got, y_got, rstd_got = ops.rmsnorm_add_fwd(res, bc, gamma, rs, Lr, w, eps, branch_slot=sb, y_slot=sy)
    assert torch.equal(r_got, r_ref)
    yrows = torch.from_numpy(np.nonzero(np.repeat(ky, Lr))[0]).to(DEV)
    assert torch.equal(y_got[:len(yrows)], y_ref[yrows]) and torch.equal(rstd_got[yrows], rstd_ref[yrows])
    # backward
    dres = randn(Mr, D, seed=5, dtype=rt)
    dy_full = randn(Mr, D, seed=6)
    dy_full[torch.from_numpy(np.repeat(~ky, Lr)).to(DEV)] = 0.0                # what a dropped sample's dy IS on the multiply-by-zero path
    dyc, _ = _compact(dy_full, ky, Lr)
    rstd_poison = rstd_ref.clone()
    rstd_poison[torch.from_numpy(np.repeat(~ky, Lr)).to(DEV)] = float("nan")  # the skip forward never wrote these
    for extra in ((None, randn(Mr, D, seed=7)) if rt == BF16 else (None,)):
        a = ops.rmsnorm_add_bwd(dy_full, dres.clone(), r_ref, rstd_ref, w, branch, gamma, rs, Lr, want_dbias=True, dres_extra=extra)
        b = ops.rmsnorm_add_bwd(dyc, dres.clone(), r_ref, rstd_poison, w, bc, gamma, rs, Lr, want_dbias=True, dres_extra=extra,
                                y_slot=sy, branch_slot=sb)
        assert torch.equal(a[0], b[0]), "dres"
        assert torch.equal(b[1][:len(brows)], a[1][torch.from_numpy(brows).to(DEV)]), "dbranch"
        for k, nm in ((2, "dw"), (3, "dgamma"), (4, "dbias")):
            assert bool(torch.isfinite(b[k]).all()) and torch.equal(a[k], b[k]), nm


@pytest.mark.parametrize("D,H", [(1408, 16), (384, 6), (512, 4)])       # head dims 88, 64 and 128 (the 6B model's: its dK / dV runs on the 16x16 kernel)
@pytest.mark.parametrize("frac", [0.0, 0.6, 1.0])
def test_qk_norm_and_attention_follow_device_counts(D, H, frac):
    """q/k RMSNorm (fwd, bwd) and the 32x32 attention kernels (fwd, dQ, dK/dV) on the first *count clips: equal to the launch on those clips
    alone, bitwise; clips behind the count are neither read (NaN) nor written (sentinel)."""
    B, Lr = 6, 417 if D == 1408 else 97
    nb = int(round(B * frac))
    Mr, Mk = B * Lr, nb * Lr
    qkv0 = randn(Mr, 3 * D, seed=1)
    qkv0[Mk:] = float("nan")
    wq, wk = randn(D, seed=2, dtype=F32) + 1.0, randn(D, seed=3, dtype=F32) + 1.0
    qkv = qkv0.clone()
    rq, rk = ops.qk_rmsnorm_fwd(qkv, wq, wk, 1e-6, m_dev=cnt(Mk))
    assert bool(torch.isnan(qkv[Mk:]).all())
    if nb == 0:
        dq = ops.flash_attn_bwd_packed(qkv, torch.zeros((Mr, D), dtype=BF16, device=DEV), torch.zeros((Mr, D), dtype=BF16, device=DEV),
                                       torch.zeros((B, H, Lr), dtype=F32, device=DEV), B, Lr, H, nb_dev=cnt(0))
        return
    ref = qkv0[:Mk].clone()
    rq_r, rk_r = ops.qk_rmsnorm_fwd(ref, wq, wk, 1e-6)
    assert torch.equal(qkv[:Mk], ref) and torch.equal(rq[:Mk], rq_r) and torch.equal(rk[:Mk], rk_r)
    att, lse = ops.flash_attn_fwd_packed(qkv, B, Lr, H, nb_dev=cnt(nb))
    att_r, lse_r = ops.flash_attn_fwd_packed(ref, nb, Lr, H)
    assert torch.equal(att[:Mk], att_r) and torch.equal(lse[:nb], lse_r)
    datt = randn(Mr, D, seed=4, scale=0.1)
    datt[Mk:] = float("nan")
    dqkv = ops.flash_attn_bwd_packed(qkv, att, datt, lse, B, Lr, H, nb_dev=cnt(nb))
    dqkv_r = ops.flash_attn_bwd_packed(ref, att_r, datt[:Mk].contiguous(), lse_r, nb, Lr, H)
    assert torch.equal(dqkv[:Mk], dqkv_r)
    dqkv[Mk:] = float("nan")
    g = ops.qk_rmsnorm_bwd(qkv, dqkv, wq, wk, rq, rk, m_dev=cnt(Mk))
    g_r = ops.qk_rmsnorm_bwd(ref, dqkv_r, wq, wk, rq_r, rk_r)
    assert torch.equal(dqkv[:Mk], dqkv_r)
    for a, b in zip(g, g_r):
        assert bool(torch.isfinite(a).all()) and rel(a, b) < 1e-5            # (the partial sums are split over another number of workgroups)


# ---- the block stack --------------------------------------------------------------------------------------------------------------------
def _build(cfg, params, rate, skip, **kw):
    m = M.PretrainInternVideo2(
        img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
        mlp_ratio=cfg.mlp_ratio, num_frames=cfg.num_frames, tubelet_size=cfg.tubelet_size, drop_path_rate=rate,
        attn_pool_num_heads=cfg.attn_pool_num_heads, clip_embed_dim=cfg.clip_embed_dim,
        clip_teacher_embed_dim=cfg.clip_teacher_embed_dim, clip_teacher_final_dim=cfg.clip_teacher_final_dim,
        clip_return_layer=cfg.clip_return_layer, mae_teacher_embed_dim=cfg.mae_teacher_embed_dim,
        mae_return_layer=cfg.mae_return_layer, **kw)
    m.load_state_dict(params, strict=True)
    m.drop_path_skip = skip
    return m.to(DEV).train()


def _losses(out, targets):
    oc, of, om = out
    tc, tf, tm = (t.to(oc.device) for t in targets)
    return (2 - 2 * (oc.float() * tc).sum(-1)).mean() + (2 - 2 * (of.float() * tf).sum(-1)).mean() + (2 - 2 * (om.float() * tm).sum(-1)).mean()


def _run(model, video, mask, targets, U):
    model._dp_uniform = U
    out = model(video.to(DEV), torch.from_numpy(mask))
    loss = _losses(out, targets)
    loss.backward()
    return loss.detach().clone(), [o.detach().clone() for o in out], {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


def _cfg(name):
    if name == "hd128":                                                      # head dim 128, as the 6B model
        return O.StudentConfig(img_size=56, embed_dim=256, depth=3, num_heads=2, mlp_ratio=4.0, num_frames=4, attn_pool_num_heads=2, clip_embed_dim=64,
                               clip_teacher_embed_dim=96, clip_teacher_final_dim=64, clip_return_layer=2, mae_teacher_embed_dim=128, mae_return_layer=2)
    return O.named_config(name)


@pytest.mark.parametrize("residual,n_cp,cfg_name", [("fp32", 0, "tiny88"), ("bf16", 0, "tiny88"), ("fp32", 2, "tiny88"), ("bf16", 2, "tiny88"), ("bf16", 0, "hd128")])
def test_block_stack_skip_equals_multiply_by_zero(residual, n_cp, cfg_name, plain_tiles):
    """the whole student (tiny88: head dim 88 as the 1B model), DropPath 0.5 so that whole branches vanish: the skipping stack against the
    stack that computes every sample and multiplies by 0 -- on the same draws, one of which drops EVERY sample of a branch and one NONE.
    Same GEMM kernel on both sides (256^2, whole tiles): outputs and loss bitwise, every gradient that is not a sum over token rows
    bitwise, weight gradients to the rounding of their summation order.  n_cp: the same with activation recomputation (P:294-295)."""
    cfg = _cfg(cfg_name)
    params = O.synthetic_params(cfg, seed=5)
    B = 6
    video, mask, targets = O.synthetic_batch(cfg, B, 6, seed=5)
    g = torch.Generator().manual_seed(3)
    U = torch.rand((cfg.depth, 2, B), generator=g)
    U[1, 0] = 0.0                                                            # block 1, attention branch: every sample dropped (keep < 1)
    U[1, 1] = 0.999                                                          # block 1, MLP branch: none dropped
    kw = dict(use_checkpoint=True, checkpoint_num=n_cp) if n_cp else {}
    res = {}
    for skip in (False, True):
        model = _build(cfg, params, 0.5, skip, **kw)
        model.residual_dtype = residual
        res[skip] = _run(model, video, mask, targets, U)
        del model
    keep = 1.0 - torch.linspace(0, 0.5, cfg.depth).view(-1, 1, 1)
    dropped = int((torch.floor(keep + U) == 0).sum())
    assert dropped >= B + 3, dropped
    (l0, o0, g0), (l1, o1, g1) = res[False], res[True]
    assert torch.equal(l0, l1), (l0.item(), l1.item())
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    assert g0.keys() == g1.keys()
    worst = {k: rel(g1[k], g0[k]) for k in g0 if float(g0[k].abs().max()) > 0}
    bad = {k: v for k, v in worst.items() if v > 4e-3}
    assert not bad, bad
    exact = [k for k in g0 if k.endswith(("ls1.gamma", "ls2.gamma", "proj.bias", "fc2.bias", "norm1.weight", "norm2.weight")) and k.startswith("blocks.")]
    assert exact
    for k in exact:                                                          # row-kernel column sums: the same rows in the same order
        assert torch.equal(g0[k], g1[k]), k


@pytest.mark.parametrize("residual", ["fp32", "bf16"])
def test_skip_matches_the_reference_on_its_own_draws(residual):
    """the reference's own run with DropPath on (tests/golden/variants.npz `dp:*`, make_golden_variants.py: timm's draws recorded, six of the
    sixteen (block, branch, sample) branches dropped) against the SKIPPING stack fed the same draws: same bars as the multiply-by-zero path
    (tests/test_model_gpu.py::test_drop_path_matches_the_reference_on_its_own_draws)."""
    g = np.load(os.path.join(GOLD, "variants.npz"))
    B, n_vis, seed = (int(v) for v in g["dp:meta"])
    cfg = O.named_config("tiny64")
    params = O.synthetic_params(cfg, seed=int(g["meta"][2]))
    video, mask, targets = O.synthetic_batch(cfg, B, n_vis, seed=seed)
    model = _build(cfg, params, float(g["dp:rate"][0]), True)
    model.residual_dtype = residual
    U = torch.from_numpy(g["dp:uniform"])
    calls = []
    orig = ops.droppath_plan
    ops.droppath_plan = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        loss, out, got = _run(model, video, mask, targets, U)
    finally:
        ops.droppath_plan = orig
    assert calls, "the skipping path did not run"
    e = [rel(out[0].float(), g["dp:x_clip_align"]), rel(out[1].float(), g["dp:x_align"]), rel(out[2].float(), g["dp:x_mae_align"])]
    assert max(e) < (2e-2 if residual == "bf16" else 1e-2), e
    assert abs(loss.item() - g["dp:loss"][0]) < 1e-3 * g["dp:loss"][0], (loss.item(), g["dp:loss"][0])
    worst = {k[8:]: rel(got[k[8:]], g[k]) for k in g.files if k.startswith("dp:grad:")}
    bad = {k: v for k, v in worst.items() if v > (6e-2 if residual == "bf16" else 4e-2)}
    assert not bad, bad


def test_skip_in_the_engine_and_under_graph_replay():
    """the native training step (IVTrainEngine: main_grad buffers, grouped weight gradients with per-problem device counts, AdamW) with
    DropPath skipping: eager steps == replays of ONE captured HIP graph, bitwise, over three steps whose draws (a static device tensor the
    graph reads, refreshed between replays) keep different numbers of samples -- nothing of a count may be frozen into the graph -- and the
    skipping engine follows the multiply-by-zero engine."""
    from internvideo_amd.engine import IVTrainEngine
    cfg = O.named_config("tiny88")
    params = O.synthetic_params(cfg, seed=9)
    B = 8
    video, mask, targets = O.synthetic_batch(cfg, B, 6, seed=9)
    vd, tg = video.to(DEV).to(BF16), tuple(t.to(DEV).to(BF16) for t in targets)
    mk = torch.from_numpy(mask).to(DEV).to(torch.uint8)
    Lv = int((~mask[0]).sum())
    g = torch.Generator().manual_seed(77)
    draws = [torch.rand((cfg.depth, 2, B), generator=g) for _ in range(3)]
    draws[1][cfg.depth - 1, 0] = 0.0                                         # second step: a branch with every sample dropped (keep 0.6 + 0 < 1)
    keep = 1.0 - torch.linspace(0, 0.4, cfg.depth).view(-1, 1, 1)
    kept = [int(torch.floor(keep + u).sum()) for u in draws]
    assert len(set(kept)) == 3, kept                                         # three different amounts of work
    traj = {}
    for tag, skip, graph in (("zero", False, False), ("skip", True, False), ("skip_graph", True, True)):
        model = _build(cfg, params, 0.4, skip)
        model.residual_dtype = "bf16"
        U = draws[0].clone().to(DEV)
        model._dp_uniform = U
        eng = IVTrainEngine(model, lr=1e-3, weight_decay=0.05, max_grad_norm=1.0)
        start = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in eng.state_dict().items()}
        losses = []
        if graph:
            eng.capture_step(vd, mk, tg, L=Lv)
            eng.load_state_dict(start)                                       # the warm-up steps of the capture do not count
        for k in range(3):
            U.copy_(draws[k])
            if graph:
                losses.append(float(eng.train_step_graphed()[0]))
            else:
                losses.append(float(eng.train_step(vd, mk, tg)[0]))
        traj[tag] = (losses, eng.master.clone())
        eng.close()
        del eng, model
    assert traj["skip"][0] == traj["skip_graph"][0], (traj["skip"][0], traj["skip_graph"][0])
    assert torch.equal(traj["skip"][1], traj["skip_graph"][1])
    for a, b in zip(traj["zero"][0], traj["skip"][0]):
        assert abs(a - b) < 2e-3 * abs(a), (traj["zero"][0], traj["skip"][0])
    assert len(set(traj["skip"][0])) == 3


@pytest.mark.parametrize("M,N,K,flavour", [(417 * 32, 1408, 6144, "fwd"), (417 * 32, 1408, 4224, "dgrad"), (417 * 128, 1408, 6144, "fwd"), (417 * 40, 1408, 6144, "dgrad")])
@pytest.mark.parametrize("frac", [0.55, 0.8, 0.875, 0.97])
def test_device_count_launches_split_their_tail_round_along_k(M, N, K, flavour, frac):
    """long-K launches under a device-side row count cut a mostly empty last round into K slices -- planned INSIDE the kernel from the real tile
    count (gemm256_kernel, DYN && SPLIT), the launch being sized for the full row count: the first *m_dev rows equal the fp32 product to bf16
    rounding, two runs are bitwise identical (the slices are summed in slice order), nothing behind the count is read (NaN) or written, and the
    plan really engages for some of the counts (a launch with the split switched off gives another rounding of the same numbers)."""
    lib = L_.load()
    Mk = int(M * frac) // 417 * 417
    md = cnt(Mk)
    a = randn(M, K, seed=1)
    w = randn(N, K, seed=2, scale=K ** -0.5)
    a[Mk:] = float("nan")
    kw = dict(a_kc=True, b_kc=(flavour == "fwd"))
    wb = w if flavour == "fwd" else w.t().contiguous()          # dgrad: the weight as stored, [K][N] rows-contiguous
    out1 = torch.full((M, N), 5.0, dtype=BF16, device=DEV)
    ops.gemm(a, wb, out=out1, m_dev=md, **kw)
    out2 = torch.full((M, N), 5.0, dtype=BF16, device=DEV)
    ops.gemm(a, wb, out=out2, m_dev=md, **kw)
    assert torch.equal(out1, out2)
    assert bool((out1[Mk:] == 5.0).all()) and bool(torch.isfinite(out1[:Mk]).all())
    ref = a[:Mk].float() @ w.float().t()
    assert rel(out1[:Mk].float(), ref) < 3e-3
    lib.ivh_gemm256_debug_split(0)
    try:
        out3 = torch.full((M, N), 5.0, dtype=BF16, device=DEV)
        ops.gemm(a, wb, out=out3, m_dev=md, **kw)
    finally:
        lib.ivh_gemm256_debug_split(1)
    assert rel(out3[:Mk].float(), ref) < 3e-3
    tiles = -(-Mk // 256) * -(-N // 256)
    rem = tiles % 256
    planned = rem > 0 and 256 // rem >= 2
    if planned:
        assert not torch.equal(out1[:Mk], out3[:Mk]), "the tail split did not engage"
    else:
        assert torch.equal(out1[:Mk], out3[:Mk])


def test_1B_at_the_bench_batch_skip_equals_multiply_by_zero_and_graph_replay_equals_eager(plain_tiles):
    """VERDICT r5 next 1, at the size the headline is measured on: pretrain_internvideo2_1B_patch14_224, B = 128, L = 417, drop_path 0.25, bf16
    residual stream, the native engine (flat main_grad buffers, grouped weight gradients with one device count per problem).
    (1) The skipping stack against the multiply-by-zero stack on the SAME draws (a device tensor both read): loss bitwise (same GEMM kernel on
        both sides: whole 256 x 256 tiles), the gradient norm and EVERY gradient of the 1.07 G parameters (flat buffers) to the rounding of
        the weight gradients' summation order.
    (2) The skipping step captured once into a HIP graph and replayed on three different draws against eager steps on the same draws: loss,
        gradient norm and updated weights bitwise -- the kept counts (about 7 % apart between the draws) are read from device memory by every
        replay, nothing of them is frozen into the graph."""
    from internvideo_amd.engine import IVTrainEngine
    B = int(os.environ.get("IV_FULLSIZE_BATCH", "128"))
    L, n_vis, depth = 417, 52, 40
    g = torch.Generator(device="cpu").manual_seed(0)
    video = torch.rand((B, 3, 8, 224, 224), generator=g).to(DEV).to(BF16)
    mask = torch.ones((B, 8, 256), dtype=torch.bool)
    for b in range(B):
        for t in range(8):
            mask[b, t, torch.randperm(256, generator=g)[:n_vis]] = False
    mask = torch.cat([torch.zeros((B, 1), dtype=torch.bool), mask.reshape(B, -1)], dim=1).to(DEV).to(torch.uint8)
    gt = torch.Generator(device="cpu").manual_seed(1)
    unit = lambda *s: torch.nn.functional.normalize(torch.randn(*s, generator=gt), dim=-1).to(DEV).to(BF16)   # noqa: E731
    tg = (unit(6, B, L, 3200), unit(B, 768), unit(4, B, L - 1, 1408))
    draws = [torch.rand((depth, 2, B), generator=gt) for _ in range(3)]
    draws[1][depth - 1, 1] = 0.0                                             # one draw drops a whole branch (keep 0.75 + 0 < 1)
    keep = 1.0 - torch.linspace(0, 0.25, depth).view(-1, 1, 1)
    kept = [float(torch.floor(keep + u).mean()) for u in draws]
    assert max(kept) - min(kept) > 0.002 and all(0.8 < k < 0.95 for k in kept), kept

    def build(skip):
        torch.manual_seed(0)
        m = M.pretrain_internvideo2_1B_patch14_224(clip_return_layer=6, mae_return_layer=4, drop_path_rate=0.25, num_frames=8).to(DEV).train()
        m.residual_dtype = "bf16"
        m.drop_path_skip = skip
        m._dp_uniform = draws[0].clone().to(DEV)
        return m, IVTrainEngine(m, lr=1e-4, max_grad_norm=3.0)

    # ---- (1) skip vs multiply-by-zero, one step on draw 0
    res = {}
    for skip in (False, True):
        m, eng = build(skip)
        loss, _ = eng.train_step(video, mask, tg)
        res[skip] = (loss.clone(), eng.grad_norm.clone(), eng.grad_mat.float().clone(), eng.grad_vec.clone())
        if not skip:
            eng.close(); del m, eng
            torch.cuda.empty_cache()
    l0, n0, gm0, gv0 = res[False]
    l1, n1, gm1, gv1 = res[True]
    assert torch.isfinite(l0).item() and torch.equal(l0, l1), (l0.item(), l1.item())
    assert abs(n0.item() - n1.item()) < 1e-4 * n0.item(), (n0.item(), n1.item())
    assert rel(gm1, gm0) < 2e-3 and rel(gv1, gv0) < 1e-4, (rel(gm1, gm0), rel(gv1, gv0))
    del res, gm0, gm1, gv0, gv1
    # ---- (2) eager vs graph replay of the skipping step on three draws (the engine of the skip run continues)
    start = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in eng.state_dict().items()}
    eager = []
    for u in draws:
        m._dp_uniform.copy_(u)
        loss, _ = eng.train_step(video, mask, tg)
        eager.append((loss.clone(), eng.grad_norm.clone()))
    master_e = eng.master.clone()
    eng.load_state_dict(start)                                              # weights, moments and the step count of the eager run's start
    del start
    torch.cuda.empty_cache()
    m._dp_uniform.copy_(draws[0])
    eng.capture_step(video, mask, tg, L=L)                                  # (the capture's warm-up passes run no optimizer: the weights stand)
    for i, u in enumerate(draws):
        m._dp_uniform.copy_(u)
        loss = eng.train_step_graphed()[0]
        assert torch.equal(loss, eager[i][0]) and torch.equal(eng.grad_norm, eager[i][1]), (i, loss.item(), eager[i][0].item())
    assert torch.equal(eng.master, master_e)
    eng.close()


def test_grouped_weight_gradients_with_a_different_k_per_problem():
    """ivh_gemm_grouped_bf16 (ABI 2): the problems of one launch may contract over different numbers of token rows -- the decoders' weight
    gradients over B L rows (CLIP branch) and B (L - 1) rows (MAE branch, no cls row) go out as ONE launch that fills the CUs where two
    launches left a mostly empty round each.  Every problem equals its own single launch bitwise and the fp32 product to bf16 rounding; mixed
    with a problem that also carries a device-side count."""
    shapes = [(3200, 1408, 417 * 12), (1408, 1408, 416 * 12), (1408, 1408, 416 * 12), (768, 1408, 12), (3200, 1408, 417 * 12)]
    probs = []
    for j, (N, Kf, K) in enumerate(shapes):
        dy = randn(K, N, seed=30 + j, scale=0.1)
        x = randn(K, Kf, seed=40 + j)
        out = torch.full((N, Kf), 3.0, dtype=BF16, device=DEV)
        kd = cnt(K - 417) if j == 4 else None
        if kd is not None:
            dy[K - 417:] = float("nan"); x[K - 417:] = float("nan")
        probs.append((dy, x, out, kd))
    ops.gemm_grouped(probs, a_kc=False, b_kc=False)
    for (dy, x, out, kd), (N, Kf, K) in zip(probs, shapes):
        kk = K - 417 if kd is not None else K
        ref = dy[:kk].float().t() @ x[:kk].float()
        assert bool(torch.isfinite(out).all()) and rel(out.float(), ref) < 6e-3, (N, Kf, K)
        ops.set_gemm_kernel(2)
        try:
            single = ops.gemm(dy, x, a_kc=False, b_kc=False, k_dev=kd)
        finally:
            ops.set_gemm_kernel(0)
        assert torch.equal(out, single), (N, Kf, K)
