"""Parity of every HIP kernel (through the C ABI) against fp32 math on the same bf16-rounded inputs.
Runs only on a real MI355X (`-m gpu`)."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from internvideo_amd import ops  # noqa: E402
from oracle import internvideo2_oracle as O  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def bf(x):
    return x.to(torch.bfloat16)


def randn(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


# ----------------------------------------------------------------------------------------------------------------
def test_probe_tr16_semantics():
    """ds_read_b64_tr_b16: lane i of a 16-lane group receives column i (4 rows) of the 4x16 block its group addresses."""
    inp = torch.arange(256, dtype=torch.int16, device=DEV)      # in[row][col] = row*64 + col
    out = ops.probe_tr16(inp).cpu().numpy()
    exp = np.zeros((64, 4), dtype=np.int16)
    for lane in range(64):
        for j in range(4):
            exp[lane, j] = j * 64 + 16 * (lane >> 4) + (lane & 15)
    assert np.array_equal(out, exp), f"tr16 layout differs:\n{out[:20]}\nexpected\n{exp[:20]}"


def test_probe_mfma16_layout():
    a = bf(randn(16, 32, seed=1)); b = bf(randn(16, 32, seed=2))      # asymmetric operands
    c = ops.probe_mfma16(a, b)                                      # c[n][m] = sum_k a[m][k] b[n][k]
    ref = b.float() @ a.float().t()
    assert rel(c, ref) < 1e-5, (c[:4, :4], ref[:4, :4])


def test_probe_mfma32_layout():
    """v_mfma_f32_32x32x16_bf16 operand / accumulator layout the 32x32 attention kernels are built on (asymmetric operands)"""
    a = bf(randn(32, 16, seed=3)); b = bf(randn(32, 16, seed=4))
    c = ops.probe_mfma32(a, b)                                      # c[i][j] = sum_k a[i][k] b[j][k]
    ref = a.float() @ b.float().t()
    assert rel(c, ref) < 1e-5, (c[:4, :4], ref[:4, :4])


# ----------------------------------------------------------------------------------------------------------------
GEMM_SHAPES = [(128, 128, 64), (16, 8, 8), (200, 136, 72), (417 * 2, 1408, 1408), (130, 264, 6144), (1000, 96, 176)]


@pytest.fixture(params=[1, 2], ids=["k128", "k256"])
def gemm_kernel(request):
    """run the test once per GEMM kernel (128^2 4-wave, 256^2 8-wave ping-pong), then restore the heuristic"""
    ops.set_gemm_kernel(request.param)
    yield request.param
    ops.set_gemm_kernel(0)


# shapes that exercise the 256^2 kernel's edges: odd / even / single K-step counts (ghost step), K tails, ragged M and N,
# more tiles than CUs (several dispatch rounds)
GEMM_SHAPES += [(512, 512, 256), (520, 264, 192), (300, 256, 64), (256, 256, 8), (1336, 1408, 1336), (8192, 8192, 320)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("a_kc,b_kc", [(True, True), (True, False), (False, True), (False, False)])
def test_gemm_layouts(M, N, K, a_kc, b_kc, gemm_kernel):
    if not a_kc and M % 8:
        M = (M + 7) // 8 * 8
    A = bf(randn(M, K, seed=3)); Bm = bf(randn(N, K, seed=4))
    a = A if a_kc else A.t().contiguous()
    b = Bm if b_kc else Bm.t().contiguous()
    out = ops.gemm(a, b, a_kc=a_kc, b_kc=b_kc)
    ref = A.float() @ Bm.float().t()
    assert out.shape == (M, N)
    err = rel(out.float(), ref)
    assert err < 4e-3, f"rel {err}"
    # element-wise: bf16 output rounding only
    tol = 1e-2 * ref.abs() + 1e-2 * math.sqrt(K) * 0.05
    assert ((out.float() - ref).abs() <= tol).all()


def test_gemm_epilogues(gemm_kernel):
    M, N, K = 300, 264, 136
    A = bf(randn(M, K, seed=5)); W = bf(randn(N, K, seed=6, scale=0.1)); bias = randn(N, seed=7)
    pre_ref = A.float() @ W.float().t() + bias
    for act, fn in (("gelu_erf", lambda x: O.gelu(x, "erf")), ("gelu_tanh", lambda x: O.gelu(x, "tanh"))):
        out, pre = ops.gemm(A, W, bias=bias, act=act, want_preact=True)
        assert rel(pre.float(), pre_ref) < 4e-3
        assert rel(out.float(), fn(pre_ref)) < 5e-3
    # fp32 output + alpha
    o32 = ops.gemm(A, W, out_fp32=True, alpha=0.5)
    assert rel(o32, 0.5 * (A.float() @ W.float().t())) < 1e-5
    # dgrad with fused gelu' : dU = (dY @ W2) * gelu'(u)
    u = bf(randn(M, N, seed=8))
    dY = bf(randn(M, 72, seed=9)); W2 = bf(randn(72, N, seed=10, scale=0.1))     # fc2.weight [out=72, in=N]
    for act in ("gelu_erf", "gelu_tanh"):
        got = ops.gemm(dY, W2, a_kc=True, b_kc=False, dact_in=u, act=act)
        uu = u.float().requires_grad_(True)
        O.gelu(uu, "erf" if act == "gelu_erf" else "tanh").backward(dY.float() @ W2.float())
        assert rel(got.float(), uu.grad) < 5e-3
    # act = 3 ("gelu_erf_d"): the forward's second output is gelu'(pre-activation), the backward multiplies by it as it is
    g3, d3 = ops.gemm(A, W, bias=bias, act="gelu_erf_d", want_preact=True)
    pre = (A.float() @ W.float().t() + bias).requires_grad_(True)
    O.gelu(pre, "erf").sum().backward()
    assert rel(g3.float(), O.gelu(pre.detach(), "erf")) < 4e-3 and rel(d3.float(), pre.grad) < 4e-3
    got3 = ops.gemm(dY, W2, a_kc=True, b_kc=False, dact_in=bf(pre.grad), act="gelu_erf_d")
    assert rel(got3.float(), (dY.float() @ W2.float()) * bf(pre.grad).float()) < 5e-3
    # batched (decoders): out[z] = A[z] W[z]^T + bias[z]
    Ab = bf(randn(3, 100, 72, seed=11)); Wb = bf(randn(3, 40, 72, seed=12)); bb = randn(3, 40, seed=13)
    ob = ops.gemm(Ab, Wb, bias=bb)
    refb = torch.einsum("zmk,znk->zmn", Ab.float(), Wb.float()) + bb[:, None, :]
    assert rel(ob.float(), refb) < 4e-3


def _half_plan(M, N, K, cap, a_kc=True, b_kc=True, **kw):
    import ctypes as C
    from internvideo_amd import lib
    d = lib.GemmDesc()
    d.M, d.N, d.K, d.a_kc, d.b_kc, d.batch = M, N, K, int(a_kc), int(b_kc), 1
    d.lda, d.ldb, d.ldc = K, (K if b_kc else N), N
    for k, v in kw.items():
        setattr(d, k, v)
    out = (C.c_int32 * 4)()
    used = lib.load().ivh_gemm256_half_plan(C.byref(d), int(cap), out)
    return used, list(out)


# (M, N, K, workgroup cap): [whole tiles + half-width tiles per workgroup] x [N edge of 128 / ragged N edge / no edge but leftover tiles cut in
# two] x [odd / even / single K-step counts] -- the mixes the B = 128 and B = 32 steps launch on 256 CUs, scaled down by the cap
HALF_CASES = [(1336, 1408, 1336, 8), (1336, 1408, 200, 16), (2100, 1384, 264, 12), (1336, 1024, 136, 10), (2100, 1408, 64, 256), (700, 640, 8, 4)]


@pytest.mark.parametrize("mode", [1, 2], ids=["interleaved", "last"])
@pytest.mark.parametrize("M,N,K,cap", HALF_CASES)
def test_gemm_half_width_tiles(M, N, K, cap, mode):
    """gemm256.hip HALF: an output whose last column tile is at most 128 wide gets half-width tiles that skip their zero half, the leftover
    whole tiles of the last round are cut into two column halves, all scheduled after the whole tiles.  Every flavour the HALF kernels
    are built for, against fp32 products, with the workgroup count capped so that one workgroup runs whole tiles AND half tiles (mode 1: the
    half tile between its whole tiles, the shipped schedule; mode 2: after them); bit-identical to the same launch with half tiles
    switched off (same K order per output element)."""
    from internvideo_amd import lib
    L = lib.load()
    L.ivh_gemm256_debug_max_wg(cap)
    L.ivh_gemm256_debug_half(mode)
    ops.set_gemm_kernel(2)
    try:
        used, plan = _half_plan(M, N, K, cap)
        assert used == 1, plan
        tnf, hb, hs, ids = plan
        assert ids > hb and (N % 256 == 0 or N % 256 > 128 or tnf == N // 256)
        A = bf(randn(M, K, seed=3)); W = bf(randn(N, K, seed=4, scale=0.1)); bias = randn(N, seed=5)
        ref = A.float() @ W.float().t()
        for b_kc in (True, False):
            b = W if b_kc else W.t().contiguous()
            out = ops.gemm(A, b, a_kc=True, b_kc=b_kc, bias=bias)
            assert rel(out.float(), ref + bias) < 4e-3
            tol = 1e-2 * (ref + bias).abs() + 1e-2 * math.sqrt(K) * 0.05
            assert ((out.float() - (ref + bias)).abs() <= tol).all()
            L.ivh_gemm256_debug_half(0)
            try:
                plain = ops.gemm(A, b, a_kc=True, b_kc=b_kc, bias=bias)
            finally:
                L.ivh_gemm256_debug_half(mode)
            assert torch.equal(out, plain)
        # fc1 forward: gelu(x W^T + b) with the gelu' copy (EPI 2)
        g3, d3 = ops.gemm(A, W, bias=bias, act="gelu_erf_d", want_preact=True)
        pre = (ref + bias).requires_grad_(True)
        O.gelu(pre, "erf").sum().backward()
        assert rel(g3.float(), O.gelu(pre.detach(), "erf")) < 4e-3 and rel(d3.float(), pre.grad) < 4e-3
        # fc2 dgrad: (dY W2) * gelu' with the bias-gradient column sums of the layer in front (EPI 3)
        Kd = 72 if K > 72 else K
        dY = bf(randn(M, Kd, seed=9)); W2 = bf(randn(Kd, N, seed=10, scale=0.1)); dact = bf(pre.grad)
        assert _half_plan(M, N, Kd, cap, b_kc=False, act=3, dact_in=dact.data_ptr(), ldd=N)[0] == 1
        got, part = ops.gemm(dY, W2, a_kc=True, b_kc=False, dact_in=dact, act="gelu_erf_d", want_colsum=True)
        want = (dY.float() @ W2.float()) * dact.float()
        assert rel(got.float(), want) < 5e-3
        assert part is not None and rel(part.sum(0), want.sum(0)) < 2e-3
    finally:
        ops.set_gemm_kernel(0)
        L.ivh_gemm256_debug_max_wg(0)
        L.ivh_gemm256_debug_half(1)


def test_gemm_rejects_bad_arguments():
    A = bf(randn(16, 12, seed=1)); W = bf(randn(8, 12, seed=2))
    with pytest.raises(ops.InternVideoHipError):
        ops.gemm(A, W)                                  # K not a multiple of 8
    with pytest.raises(ops.InternVideoHipError):
        ops.gemm(A.float(), W)
    with pytest.raises(ops.InternVideoHipError):
        ops.gemm(A.cpu(), W.cpu())                      # no CPU path


# ----------------------------------------------------------------------------------------------------------------
def _rms_ref(res_in, branch, gamma, rowscale, rps, w, eps):
    r = res_in + rowscale.repeat_interleave(rps)[:, None] * gamma * branch.float()
    y = O.rmsnorm(r, w, eps)
    return r, y


@pytest.mark.parametrize("M,D,rps", [(34, 128, 17), (42, 176, 21), (834, 1408, 417), (20, 3200, 10), (64, 768, 8), (1666, 3200, 833), (39, 2560, 13),
                                     (21, 4096, 7)])
def test_rmsnorm_add_fwd_bwd(M, D, rps):
    res_in = randn(M, D, seed=1); branch = bf(randn(M, D, seed=2)); gamma = 1 + 0.1 * randn(D, seed=3)
    rowscale = (torch.rand(M // rps, device=DEV) > 0.3).float() / 0.7
    w = 1 + 0.1 * randn(D, seed=4)
    res_out, y, rstd = ops.rmsnorm_add_fwd(res_in, branch, gamma, rowscale, rps, w, 1e-6)
    ri = res_in.clone().requires_grad_(True); br = branch.float().requires_grad_(True)
    gm = gamma.clone().requires_grad_(True); ww = w.clone().requires_grad_(True)
    r_ref, y_ref = _rms_ref(ri, br, gm, rowscale, rps, ww, 1e-6)
    assert rel(res_out, r_ref.detach()) < 1e-6
    assert rel(y.float(), y_ref.detach()) < 4e-3
    dy = bf(randn(M, D, seed=5)); dres = randn(M, D, seed=6)
    (y_ref * dy.float()).sum().backward(retain_graph=True)
    (r_ref * dres).sum().backward()
    dres_in, dbranch, dw, dg = ops.rmsnorm_add_bwd(dy, dres.clone(), res_out, rstd, w, branch, gamma, rowscale, rps)
    assert rel(dres_in, ri.grad) < 1e-5
    assert rel(dbranch.float(), br.grad) < 4e-3
    assert rel(dw, ww.grad) < 1e-4
    assert rel(dg, gm.grad) < 1e-4


@pytest.mark.parametrize("M,D,rps", [(42, 176, 21), (834, 1408, 417), (2085, 1408, 417), (20, 3200, 10), (21, 4096, 7), (53376, 1408, 417)])
def test_rmsnorm_add_fwd_bwd_bf16_residual_stream(M, D, rps):
    """the same pair with the residual stream in bf16 (the reference's bf16 recipe: DropoutAddRMSNorm(prenorm=True), residual_in_fp32 False,
    P:283-286, 467): the sum is formed in fp32 and rounded ONCE for the stream, the norm output comes from the unrounded sum, the backward
    normalises the stored rows.  Against a torch fp32 reference of exactly that definition."""
    res_in = bf(randn(M, D, seed=1)); branch = bf(randn(M, D, seed=2)); gamma = 1 + 0.1 * randn(D, seed=3)
    rowscale = (torch.rand(M // rps, device=DEV) > 0.3).float() / 0.7
    w = 1 + 0.1 * randn(D, seed=4)
    res_out, y, rstd = ops.rmsnorm_add_fwd(res_in, branch, gamma, rowscale, rps, w, 1e-6)
    assert res_out.dtype == torch.bfloat16 and y.dtype == torch.bfloat16
    r_ref, y_ref = _rms_ref(res_in.float(), branch.float(), gamma, rowscale, rps, w, 1e-6)
    # one rounding of the fp32 sum: at most 1 bf16 ulp from the rounded reference (the kernel's fma order differs from torch's in fp32)
    assert rel(res_out.float(), r_ref) < 3e-3 and (res_out.float() - bf(r_ref).float()).abs().max() <= 2.0 ** -7 * r_ref.abs().max()
    assert rel(y.float(), y_ref) < 4e-3
    assert rel(rstd, torch.rsqrt((r_ref * r_ref).mean(-1) + 1e-6)) < 1e-6    # statistics of the unrounded sum
    # backward: x = the stored bf16 rows
    dy = bf(randn(M, D, seed=5)); dres = bf(randn(M, D, seed=6))
    xs = res_out.float().requires_grad_(True); ww = w.clone().requires_grad_(True)
    (O.rmsnorm(xs, ww, 1e-6) * dy.float()).sum().backward()
    want_dres = dres.float() + xs.grad
    # guard rows behind every output: the interior kernel addresses rows through a scalar offset the hardware does not range-check
    guard = torch.full((4 * D,), 7.0, device=DEV).bfloat16()
    dres_buf = torch.cat([dres.reshape(-1), guard]); dres_arg = dres_buf[:M * D].view(M, D)
    dres_in, dbranch, dw, dg, db = ops.rmsnorm_add_bwd(dy, dres_arg, res_out, torch.rsqrt((xs.detach() ** 2).mean(-1) + 1e-6), w, branch, gamma,
                                                       rowscale, rps, want_dbias=True)
    assert torch.equal(dres_buf[M * D:], guard), "rows past M were written"
    assert dres_in.dtype == torch.bfloat16 and rel(dres_in.float(), want_dres) < 4e-3
    assert torch.isfinite(dres_in.float()).all() and (dres_in.float() - want_dres).abs().max() < 0.08        # EVERY row (a norm hides a few bad ones)
    rs_rows = rowscale.repeat_interleave(rps)[:, None]
    assert rel(dbranch.float(), rs_rows * gamma * want_dres) < 6e-3
    assert torch.isfinite(dbranch.float()).all() and (dbranch.float() - rs_rows * gamma * want_dres).abs().max() < 0.12
    assert rel(dw, ww.grad) < 1e-4
    assert rel(dg, (rs_rows * branch.float() * want_dres).sum(0)) < 2e-3
    assert rel(db, dbranch.float().sum(0)) < 5e-3                              # fp32 sums of the values BEFORE their bf16 rounding
    # plain norm of a bf16 stream (first block) and the final add
    _, y0, _ = ops.rmsnorm_add_fwd(res_in, None, None, None, 1, w, 1e-6, want_res_out=False)
    assert rel(y0.float(), O.rmsnorm(res_in.float(), w, 1e-6)) < 4e-3
    r2, y2, _ = ops.rmsnorm_add_fwd(res_in, branch, gamma, None, 1, None, 1e-6)
    assert y2 is None and rel(r2.float(), res_in.float() + gamma * branch.float()) < 3e-3
    with pytest.raises(Exception):                                            # one stream type per call
        ops.rmsnorm_add_bwd(dy, dres.clone(), res_out.float(), rstd, w, branch, gamma, rowscale, rps)


@pytest.mark.parametrize("M,D,rps,gam", [(42, 176, 21, True), (2085, 1408, 417, True), (2085, 1408, 417, False), (20, 3200, 10, True), (53376, 1408, 417, True)])
def test_rmsnorm_add_bwd_bf16_stream_takes_a_tap_gradient_on_load(M, D, rps, gam):
    """dres_extra: the gradient of a feature tap (the decoders read the stream after chosen blocks, P:669-688) joins dres_out inside the
    kernel's loads instead of in a pass of its own.  Both kernels (bytes-in-flight interior kernel; generic kernel when there is no LayerScale)
    against the fp32 definition, every row, and against the two-pass form (the sum rounded to bf16 first): at most one rounding apart."""
    res_out = bf(randn(M, D, seed=1)); branch = bf(randn(M, D, seed=2)); gamma = (1 + 0.1 * randn(D, seed=3)) if gam else None
    rowscale = (torch.rand(M // rps, device=DEV) > 0.3).float() / 0.7
    w = 1 + 0.1 * randn(D, seed=4)
    dy = bf(randn(M, D, seed=5)); dres = bf(randn(M, D, seed=6)); tap = bf(randn(M, D, seed=7))
    xs = res_out.float().requires_grad_(True); ww = w.clone().requires_grad_(True)
    (O.rmsnorm(xs, ww, 1e-6) * dy.float()).sum().backward()
    rstd = torch.rsqrt((xs.detach() ** 2).mean(-1) + 1e-6)
    want = dres.float() + tap.float() + xs.grad
    guard = torch.full((4 * D,), 7.0, device=DEV).bfloat16()
    buf = torch.cat([dres.reshape(-1), guard]); arg = buf[:M * D].view(M, D)
    tap0 = tap.clone()
    dres_in, dbranch, dw, dg, db = ops.rmsnorm_add_bwd(dy, arg, res_out, rstd, w, branch, gamma, rowscale, rps, want_dbias=True, dres_extra=tap)
    assert torch.equal(buf[M * D:], guard) and torch.equal(tap, tap0)
    assert torch.isfinite(dres_in.float()).all() and rel(dres_in.float(), want) < 4e-3 and (dres_in.float() - want).abs().max() < 0.1
    rs_rows = rowscale.repeat_interleave(rps)[:, None]
    g_ = gamma if gam else 1.0
    assert rel(dbranch.float(), rs_rows * g_ * want) < 6e-3 and rel(dw, ww.grad) < 1e-4 and rel(db, dbranch.float().sum(0)) < 5e-3
    if gam:
        assert rel(dg, (rs_rows * branch.float() * want).sum(0)) < 2e-3
    two_pass = ops.rmsnorm_add_bwd(dy, (dres.float() + tap.float()).bfloat16(), res_out, rstd, w, branch, gamma, rowscale, rps, want_dbias=True)[0]
    assert (dres_in.float() - two_pass.float()).abs().max() <= 2.0 ** -6 * want.abs().max()
    with pytest.raises(Exception):                                            # fp32 streams add their taps with accum_rows
        ops.rmsnorm_add_bwd(dy, dres.float(), res_out.float(), rstd, w, branch, gamma, rowscale, rps, dres_extra=tap)
    with pytest.raises(Exception):
        ops.rmsnorm_add_bwd(dy, dres.clone(), res_out, rstd, w, branch, gamma, rowscale, rps, dres_extra=tap[:-1])


def test_rmsnorm_first_block_and_final_add():
    M, D = 34, 128
    x0 = randn(M, D, seed=1); w = 1 + 0.1 * randn(D, seed=2)
    r, y, rstd = ops.rmsnorm_add_fwd(x0, None, None, None, 1, w, 1e-6, want_res_out=False)
    assert r is None and rel(y.float(), O.rmsnorm(x0, w, 1e-6)) < 4e-3
    br = bf(randn(M, D, seed=3)); g = randn(D, seed=4)
    r2, y2, _ = ops.rmsnorm_add_fwd(x0, br, g, None, 1, None, 1e-6)
    assert y2 is None and rel(r2, x0 + g * br.float()) < 1e-6
    dres = randn(M, D, seed=5)
    dres_in, dbr, dw, dg = ops.rmsnorm_add_bwd(None, dres.clone(), None, None, None, br, g, None, 1)
    assert dw is None and rel(dres_in, dres) == 0 and rel(dbr.float(), g * dres) < 4e-3
    assert rel(dg, (br.float() * dres).sum(0)) < 1e-4


@pytest.mark.parametrize("M,D", [(34, 128), (42, 176), (417, 1408), (50, 3200), (1666, 3200), (19, 4096)])
def test_qk_rmsnorm_fwd_bwd(M, D):
    qkv = bf(randn(M, 3 * D, seed=1)); wq = 1 + 0.1 * randn(D, seed=2); wk = 1 + 0.1 * randn(D, seed=3)
    q0 = qkv.float().clone()
    work = qkv.clone()
    rq, rk = ops.qk_rmsnorm_fwd(work, wq, wk, 1e-6)
    qq = q0[:, :D].clone().requires_grad_(True); kk = q0[:, D:2 * D].clone().requires_grad_(True)
    wqq = wq.clone().requires_grad_(True); wkk = wk.clone().requires_grad_(True)
    qn, kn = O.rmsnorm(qq, wqq, 1e-6), O.rmsnorm(kk, wkk, 1e-6)
    assert rel(work[:, :D].float(), qn.detach()) < 4e-3 and rel(work[:, D:2 * D].float(), kn.detach()) < 4e-3
    assert torch.equal(work[:, 2 * D:], qkv[:, 2 * D:])
    d = bf(randn(M, 3 * D, seed=4))
    (qn * d[:, :D].float()).sum().backward(); (kn * d[:, D:2 * D].float()).sum().backward()
    dwork = d.clone()
    dwq, dwk = ops.qk_rmsnorm_bwd(work, dwork, wq, wk, rq, rk)
    assert rel(dwork[:, :D].float(), qq.grad) < 1e-2 and rel(dwork[:, D:2 * D].float(), kk.grad) < 1e-2
    assert torch.equal(dwork[:, 2 * D:], d[:, 2 * D:])
    assert rel(dwq, wqq.grad) < 1e-2 and rel(dwk, wkk.grad) < 1e-2


def test_qk_rmsnorm_bwd_at_the_bench_size_every_row_and_guard_rows():
    """the bytes-in-flight backward (qk_rmsnorm_bwd_b16_kernel: four waves share a token, scalar token offsets through buffer descriptors) at
    the bench's own size, M = 128 x 417 tokens of 3 x 1408: EVERY token row of dq and dk against fp32 torch, the v segment and guard rows
    behind the buffer untouched (the r3 attempt at this kernel produced wrong rows exactly at this size), a ragged token count as well."""
    for M in (53376, 53376 - 5):
        D = 1408
        g = torch.Generator(device=DEV).manual_seed(5)
        y0 = (torch.randn((M, 3 * D), device=DEV, generator=g)).to(torch.bfloat16)
        wq = (1 + 0.1 * torch.randn(D, device=DEV, generator=g)).float(); wk = (1 + 0.1 * torch.randn(D, device=DEV, generator=g)).float()
        work = y0.clone()
        rq, rk = ops.qk_rmsnorm_fwd(work, wq, wk, 1e-6)
        full = torch.randn((M + 8, 3 * D), device=DEV, generator=g).to(torch.bfloat16)
        d0 = full.clone()
        dwork = full[:M]
        dwq, dwk = ops.qk_rmsnorm_bwd(work, dwork, wq, wk, rq, rk)
        torch.cuda.synchronize()
        assert torch.equal(full[M:], d0[M:]) and torch.equal(dwork[:, 2 * D:], d0[:M, 2 * D:])
        for a, (w, r) in enumerate(((wq, rq), (wk, rk))):
            x = y0[:, a * D:(a + 1) * D].float()
            dy = d0[:M, a * D:(a + 1) * D].float()
            xh = x * r[:, None]
            wdy = dy * w
            want = r[:, None] * (wdy - xh * (wdy * xh).mean(-1, keepdim=True))
            got = dwork[:, a * D:(a + 1) * D].float()
            row_err = (got - want).norm(dim=-1) / want.norm(dim=-1).clamp_min(1e-20)
            assert row_err.max().item() < 1.2e-2, (M, a, row_err.max().item(), int(row_err.argmax()))
            assert rel((dwq, dwk)[a], (dy * xh).sum(0)) < 3e-3


# ----------------------------------------------------------------------------------------------------------------
def _attn_ref(qkv, B, L, H):
    D = qkv.shape[1] // 3
    hd = D // H
    x = qkv.float().reshape(B, L, 3, H, hd)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    att = ((q * hd ** -0.5) @ k.transpose(-2, -1))
    lse = torch.logsumexp(att, dim=-1)
    out = (att.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * L, D)
    return out, lse


ATTN_CFGS = [(2, 17, 2, 64), (2, 21, 2, 88), (1, 64, 1, 64), (1, 1, 2, 64), (2, 130, 3, 128), (1, 417, 16, 88), (1, 200, 2, 96),
             (3, 129, 2, 88), (2, 257, 2, 64), (1, 448, 1, 128), (2, 33, 1, 104),
             # round 5, the peeled first row (flash_attn32.hip a32_row_bfrags): L % 64 == 33 -> rows 1 .. L - 1 end with a HALF tile (a single half tile;
             # an even tile count, where dK / dV falls back; 417 at hd 64), L % 64 == 1 -> whole tiles only (129 / 257 above; 833 = the 6B length)
             (1, 33, 2, 88), (2, 97, 2, 96), (1, 417, 2, 64), (2, 833, 2, 128), (1, 161, 3, 88)]


@pytest.fixture(params=[1, 2], ids=["mfma16x16x32", "mfma32x32x16"])
def attn_kernel(request):
    """both attention kernel families (csrc/flash_attn.hip, csrc/flash_attn32.hip) through the same C entry points"""
    ops.set_attn_kernel(request.param)
    yield request.param
    ops.set_attn_kernel(0)


@pytest.mark.parametrize("B,L,H,hd", ATTN_CFGS)
def test_flash_attn_fwd_bwd(B, L, H, hd, attn_kernel):
    D = H * hd
    qkv = bf(randn(B * L, 3 * D, seed=L))
    out, lse = ops.flash_attn_fwd_packed(qkv, B, L, H)
    x = qkv.float().clone().requires_grad_(True)
    ref, lse_ref = _attn_ref(x, B, L, H)
    assert rel(out.float(), ref.detach()) < 6e-3, rel(out.float(), ref.detach())
    assert (lse - lse_ref.detach()).abs().max().item() < 2e-2
    dout = bf(randn(B * L, D, seed=L + 1))
    (ref * dout.float()).sum().backward()
    dqkv = ops.flash_attn_bwd_packed(qkv, out, dout, lse, B, L, H)
    gnorm = x.grad.double().norm().item()
    for i, name in enumerate("qkv"):
        got, want = dqkv[:, i * D:(i + 1) * D].double().cpu(), x.grad[:, i * D:(i + 1) * D].double().cpu()
        # relative to the part's own norm, floored at 1e-4 of the whole gradient: with a single key (L = 1) dq and dk are exactly zero
        # in exact arithmetic (p = 1, dP = delta) and only fp32 summation-order noise (~1e-8) remains
        e = (got - want).norm().item() / max(want.norm().item(), 1e-4 * gnorm)
        assert e < 1.5e-2, f"d{name}: {e}"


def test_flash_attn_rescale_branch_is_exercised(attn_kernel):
    """spike one key so that the running max jumps in a late tile (online-softmax rescale path)."""
    B, L, H, hd = 1, 200, 1, 64
    D = H * hd
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B * L, 3 * D, generator=g)
    x[150, D:2 * D] = 6.0 * x[3, :D] / x[3, :D].norm() * math.sqrt(hd)       # key 150 aligned with query 3
    qkv = bf(x.to(DEV))
    out, lse = ops.flash_attn_fwd_packed(qkv, B, L, H)
    ref, lse_ref = _attn_ref(qkv, B, L, H)
    assert rel(out.float(), ref) < 6e-3 and (lse - lse_ref).abs().max().item() < 2e-2


@pytest.mark.parametrize("B,Lq,Lk,H,hd", [(2, 1, 50, 2, 64), (2, 37, 150, 2, 88), (1, 130, 70, 3, 128), (3, 64, 64, 1, 96)])
def test_flash_attn_cross_lengths_and_kv_len(B, Lq, Lk, H, hd, attn_kernel):
    """Lq != Lk (the attention-pooling projector is the Lq = 1 case) and right-padded key batches (kv_len), unpacked q / k / v views"""
    q = bf(randn(B, Lq, H, hd, seed=1)); kv = bf(randn(2, B, Lk, H, hd, seed=2))
    k, v = kv[0], kv[1]
    kv_len = torch.tensor([max(1, Lk - 7 * (b + 1)) for b in range(B)], dtype=torch.int32, device=DEV)
    for lens in (None, kv_len):
        out, lse = ops.flash_attn_fwd(q, k, v, kv_len=lens)
        qq, kk, vv = (t.float().clone().requires_grad_(True) for t in (q, k, v))
        att = torch.einsum("bqhd,bkhd->bhqk", qq * hd ** -0.5, kk)
        if lens is not None:
            dead = torch.arange(Lk, device=DEV)[None, :] >= lens[:, None].long()
            att = att.masked_fill(dead[:, None, None, :], float("-inf"))
        ref = torch.einsum("bhqk,bkhd->bqhd", att.softmax(-1), vv)
        assert rel(out.float(), ref.detach()) < 6e-3
        assert (lse - torch.logsumexp(att, -1).detach()).abs().max().item() < 2e-2
        dout = bf(randn(B, Lq, H, hd, seed=3))
        (ref * dout.float()).sum().backward()
        dq, dkv = ops.flash_attn_bwd(q, k, v, out, dout, lse, kv_len=lens)
        gn = math.sqrt(qq.grad.double().norm().item() ** 2 + kk.grad.double().norm().item() ** 2 + vv.grad.double().norm().item() ** 2)
        for got, want, name in ((dq, qq.grad, "dq"), (dkv[0], kk.grad, "dk"), (dkv[1], vv.grad, "dv")):
            e = (got.double() - want.double()).norm().item() / max(want.double().norm().item(), 1e-4 * gn)
            assert e < 1.5e-2, f"{name}: {e} (kv_len {'on' if lens is not None else 'off'})"
        if lens is not None:                      # padded keys receive exactly zero gradient rows
            for b in range(B):
                assert dkv[:, b, int(lens[b]):].abs().max().item() == 0.0


def test_flash_attn_kernel_families_agree_at_the_1B_shape():
    """L = 417, 16 heads of 88 (InternVideo2-1B), B = 4: the two kernel families against each other, outputs and all three gradients"""
    B, L, H, hd = 4, 417, 16, 88
    D = H * hd
    qkv = bf(randn(B * L, 3 * D, seed=11)); dout = bf(randn(B * L, D, seed=12))
    res = {}
    for impl in (1, 2):
        ops.set_attn_kernel(impl)
        try:
            out, lse = ops.flash_attn_fwd_packed(qkv, B, L, H)
            dqkv = ops.flash_attn_bwd_packed(qkv, out, dout, lse, B, L, H)
        finally:
            ops.set_attn_kernel(0)
        res[impl] = (out.float(), lse, dqkv.float())
    assert rel(res[2][0], res[1][0]) < 4e-3
    assert (res[2][1] - res[1][1]).abs().max().item() < 1e-3
    assert rel(res[2][2], res[1][2]) < 8e-3


@pytest.mark.parametrize("B,L,H,hd", [(1, 1, 2, 88), (3, 33, 2, 88), (2, 64, 3, 64), (2, 97, 2, 96), (2, 257, 2, 64), (3, 417, 4, 88), (2, 833, 2, 128), (2, 130, 3, 128)])
def test_flash_attn32_unpacked_fp32_is_bitwise_the_packed_build(B, L, H, hd):
    """round 6: the 32x32 kernels run compiled WITHOUT the packed fp32 instructions by default (target attribute no-packed-fp32-ops: v_pk_fma / v_pk_mul /
    v_pk_add of one wave do not run beside another wave's MFMAs, profiles/r6_mfma_valu_mix_*.jsonl).  Same arithmetic in the same order: out, lse and all
    three gradients must be bit-identical to the packed instantiations (ivh_probe_attn32_unpacked)."""
    from internvideo_amd.lib import call
    D = H * hd
    qkv = bf(randn(B * L, 3 * D, seed=L + 5)); dout = bf(randn(B * L, D, seed=L + 6))
    ops.set_attn_kernel(2)
    res = {}
    try:
        for unp in (0, 1):
            call("ivh_probe_attn32_unpacked", unp)
            out, lse = ops.flash_attn_fwd_packed(qkv, B, L, H)
            res[unp] = (out, lse, ops.flash_attn_bwd_packed(qkv, out, dout, lse, B, L, H))
    finally:
        call("ivh_probe_attn32_unpacked", 1)
        ops.set_attn_kernel(0)
    for a, b_, name in zip(res[0], res[1], ("out", "lse", "dqkv")):
        assert torch.equal(a, b_), name


@pytest.mark.parametrize("mode", [1, 2, 3, 4, 5, 7])
def test_flash_attn32_two_wave_group_probe_is_bitwise_the_shipped_forward(mode):
    """round 6: attn32pp_fwd_kernel -- 8-wave workgroups whose two wave groups run one segment apart (MFMA segment of one beside
    the softmax segment of the other) -- is a measurement probe (slower than the shipped kernel: profiles/r6_attn_two_wave_groups_ab_v1.jsonl), kept
    correct: same arithmetic, same order, so out / lse equal the shipped forward bit for bit, ragged tails and device-side clip counts included."""
    from internvideo_amd.lib import call
    ops.set_attn_kernel(2)
    try:
        for B, L, H, hd in [(2, 1, 2, 88), (3, 33, 2, 88), (2, 97, 2, 88), (2, 256, 2, 88), (2, 257, 2, 88), (3, 417, 4, 88), (2, 833, 2, 88), (2, 161, 2, 64)]:
            if mode >= 3 and hd <= 64:
                continue                                      # modes 3-7 are instantiated for the 1B head dim only (7: the shipped kernel at two waves per SIMD)
            qkv = bf(randn(B * L, 3 * H * hd, seed=L + 9))
            call("ivh_probe_attn32_pingpong", 0)
            ref = ops.flash_attn_fwd_packed(qkv, B, L, H)
            call("ivh_probe_attn32_pingpong", mode)
            got = ops.flash_attn_fwd_packed(qkv, B, L, H)
            assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1]), (B, L, H, hd)
        # device-side clip count: the workgroups of absent clips leave, the rest is unchanged
        B, L, H, hd = 6, 417, 4, 88
        qkv = bf(randn(B * L, 3 * H * hd, seed=3))
        nb = torch.tensor([4], dtype=torch.int32, device=DEV)
        call("ivh_probe_attn32_pingpong", 0)
        ref = ops.flash_attn_fwd_packed(qkv, B, L, H, nb_dev=nb)
        call("ivh_probe_attn32_pingpong", mode)
        got = ops.flash_attn_fwd_packed(qkv, B, L, H, nb_dev=nb)
        assert torch.equal(ref[0][:4 * L], got[0][:4 * L]) and torch.equal(ref[1].reshape(B, -1)[:4], got[1].reshape(B, -1)[:4])
    finally:
        call("ivh_probe_attn32_pingpong", 0)
        ops.set_attn_kernel(0)


# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,C", [(34, 96), (40, 176), (417, 3200), (64, 768), (2500, 3200), (23, 2304), (1251, 1408)])
def test_ln_l2_fwd_bwd(M, C):
    y = bf(randn(M, C, seed=1)); w = 1 + 0.1 * randn(C, seed=2); b = 0.1 * randn(C, seed=3)
    t = randn(M, C, seed=4); t = t / t.norm(dim=-1, keepdim=True)
    out, stats, loss_rows = ops.ln_l2_fwd(y, w, b, 1e-5, target=t)
    yy = y.float().requires_grad_(True); ww = w.clone().requires_grad_(True); bb = b.clone().requires_grad_(True)
    ln = O.layernorm(yy, ww, bb, 1e-5)
    o_ref = ln / ln.norm(dim=-1, keepdim=True)
    assert rel(out.float(), o_ref.detach()) < 4e-3
    lr_ref = 2 - 2 * (o_ref * t).sum(-1)
    assert rel(loss_rows, lr_ref.detach()) < 1e-5
    loss = ops.sum_rows(loss_rows, 1.0 / M)
    assert abs(loss.item() - lr_ref.mean().item()) < 1e-5 * abs(lr_ref.mean().item()) + 1e-6
    lr_ref.mean().backward()
    dy, dw, db = ops.ln_l2_bwd(y, w, b, stats, None, t, -2.0 / M)
    assert rel(dy.float(), yy.grad) < 5e-3 and rel(dw, ww.grad) < 1e-4 and rel(db, bb.grad) < 1e-4
    # explicit upstream gradient (drop-in mode)
    yy.grad = None; ww.grad = None; bb.grad = None
    ln = O.layernorm(yy, ww, bb, 1e-5); o_ref = ln / ln.norm(dim=-1, keepdim=True)
    do = randn(M, C, seed=5)
    (o_ref * do).sum().backward()
    dy, dw, db = ops.ln_l2_bwd(y, w, b, stats, do, None, 0.0)
    assert rel(dy.float(), yy.grad) < 5e-3 and rel(dw, ww.grad) < 1e-4 and rel(db, bb.grad) < 1e-4
    # bf16 targets (teacher outputs under autocast)
    _, _, lr2 = ops.ln_l2_fwd(y, w, b, 1e-5, want_out=False, target=bf(t))
    assert rel(lr2, 2 - 2 * (o_ref.detach() * bf(t).float()).sum(-1)) < 1e-5
    # ... and their backward with the upstream scalar on the device: what the training step runs (the row-prefetching kernel up to C = 4096)
    yy.grad = None; ww.grad = None; bb.grad = None
    ln = O.layernorm(yy, ww, bb, 1e-5); o_ref = ln / ln.norm(dim=-1, keepdim=True)
    ((2 - 2 * (o_ref * bf(t).float()).sum(-1)).sum() * 0.37).backward()
    dy, dw, db = ops.ln_l2_bwd(y, w, b, stats, None, bf(t), -2.0, dscale_dev=torch.tensor([0.37], device=DEV))
    assert rel(dy.float(), yy.grad) < 5e-3 and rel(dw, ww.grad) < 1e-4 and rel(db, bb.grad) < 1e-4
    assert torch.isfinite(dy.float()).all() and (dy.float() - yy.grad).abs().max() < 0.02 * yy.grad.abs().max() + 1e-6      # every row


# ----------------------------------------------------------------------------------------------------------------
def test_token_edge_kernels_bit_exact():
    cfg = O.named_config("tiny88")
    B, n_vis = 3, 5
    video, mask, _ = O.synthetic_batch(cfg, B, n_vis, seed=3)
    L = 1 + cfg.grid[0] * n_vis
    idx_ref = O.visible_indices(mask)
    vis, inv, cnt = ops.mask_to_indices(torch.from_numpy(mask).to(DEV), L)
    assert np.array_equal(vis.cpu().numpy(), idx_ref)                                   # bit exact
    assert (cnt.cpu().numpy() == L).all()
    inv_ref = np.full(mask.shape, -1, dtype=np.int32)
    for b in range(B):
        inv_ref[b, idx_ref[b]] = np.arange(L)
    assert np.array_equal(inv.cpu().numpy(), inv_ref)
    # im2col == the reference's Conv3d unfolding of the bf16-cast input (bit exact)
    Kreal = 3 * cfg.tubelet_size * cfg.patch_size ** 2
    Kp = (Kreal + 63) // 64 * 64
    cols = ops.patch_im2col(video.to(DEV), vis, cfg.tubelet_size, cfg.patch_size, Kp)
    t, h, w = cfg.grid
    p = cfg.patch_size
    unf = video.reshape(B, 3, t, cfg.tubelet_size, h, p, w, p).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, t * h * w, Kreal)
    exp = torch.stack([unf[b][torch.from_numpy(idx_ref[b, 1:] - 1).long()] for b in range(B)]).reshape(B * (L - 1), Kreal)
    assert torch.equal(cols[:, :Kreal].cpu(), exp.to(torch.bfloat16))
    assert (cols[:, Kreal:] == 0).all()
    cols_b = ops.patch_im2col(video.to(DEV).to(torch.bfloat16), vis, cfg.tubelet_size, cfg.patch_size, Kp)
    assert torch.equal(cols_b, cols)
    # assemble / gather / scatter-free grads
    D = cfg.embed_dim
    tok = bf(randn(B * (L - 1), D, seed=1)); cls = randn(D, seed=2); pos = randn(mask.shape[1], D, seed=3)
    x0 = ops.assemble_tokens(tok, cls, pos, vis).reshape(B, L, D)
    ii = torch.from_numpy(idx_ref).long().to(DEV)
    exp0 = torch.cat([cls.expand(B, 1, D), tok.float().reshape(B, L - 1, D)], 1) + pos[ii]
    assert torch.equal(x0, exp0)
    for skip in (0, 1):
        posk = pos[skip:]
        y = ops.add_pos_gather(x0.reshape(B * L, D), posk, vis, skip).reshape(B, L - skip, D)
        assert torch.equal(y, (x0[:, skip:] + posk[ii[:, skip:] - skip]).to(torch.bfloat16))
        src = bf(randn(2, B, L - skip, D, seed=4 + skip))
        dpos = ops.pos_grad(src, 2, B, L - skip, inv, skip)
        expd = torch.zeros(mask.shape[1] - skip, D, device=DEV)
        for k in range(2):
            for b in range(B):
                expd.index_add_(0, ii[b, skip:] - skip, src[k, b].float())
        assert rel(dpos, expd) < 1e-6
        dst = randn(B * L, D, seed=9)
        d0 = dst.clone()
        ops.accum_rows(dst, src[0].reshape(-1, D), B, L, skip, True)
        e = d0.reshape(B, L, D).clone(); e[:, skip:] += src[0].float()
        assert torch.equal(dst.reshape(B, L, D), e)
        ops.accum_rows(dst, src[1].reshape(-1, D), B, L, skip, False)
        e = torch.zeros(B, L, D, device=DEV); e[:, skip:] = src[1].float()
        assert torch.equal(dst.reshape(B, L, D), e)
        r = ops.rows_to_bf16(x0.reshape(B * L, D), B, L, skip)
        assert torch.equal(r.reshape(B, L - skip, D), x0[:, skip:].to(torch.bfloat16))


# ----------------------------------------------------------------------------------------------------------------
def test_adamw_matches_torch():
    n = 4096 + 512
    p0 = randn(n, seed=1); g_list = [randn(n, seed=10 + i, scale=0.1) for i in range(3)]
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=1e-2, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.05)
    master = p0.clone(); m = torch.zeros_like(p0); v = torch.zeros_like(p0)
    shadow = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    for step, g in enumerate(g_list, 1):
        ref.grad = g.clone(); opt.step()
        ops.adamw_step(master, m, v, bf(g) if step == 2 else g, shadow, 1e-2, 0.9, 0.98, 1e-6, 0.05, step)
        if step == 2:     # bf16 gradient path: feed the same rounded gradient to torch for the comparison
            pass
    # step 2 used a bf16-rounded gradient on our side only -> compare loosely, then exactly on a clean run
    assert rel(master, ref.detach()) < 5e-3
    master = p0.clone(); m.zero_(); v.zero_()
    ref2 = p0.clone().requires_grad_(True)
    opt2 = torch.optim.AdamW([ref2], lr=1e-2, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.05)
    for step, g in enumerate(g_list, 1):
        ref2.grad = g.clone(); opt2.step()
        ops.adamw_step(master, m, v, g, shadow, 1e-2, 0.9, 0.98, 1e-6, 0.05, step)
    assert rel(master, ref2.detach()) < 2e-6
    assert torch.equal(shadow, master.to(torch.bfloat16))
    # global norm + clip coefficient
    out = torch.zeros(1, device=DEV)
    ops.sqnorm(g_list[0], out, False); ops.sqnorm(bf(g_list[1]), out, True)
    expn = (g_list[0].double() ** 2).sum() + (bf(g_list[1]).double() ** 2).sum()
    assert abs(out.item() - expn.item()) / expn.item() < 1e-5
    coef, nrm = ops.clip_coef(out, 3.0)
    assert abs(nrm.item() - math.sqrt(expn.item())) / nrm.item() < 1e-5
    assert abs(coef.item() - min(1.0, 3.0 / (math.sqrt(expn.item()) + 1e-6))) < 1e-6


def test_vtc_loss_matches_oracle_and_reference_golden():
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tables.npz"))
    rng = np.random.Generator(np.random.PCG64(5))
    v = torch.from_numpy(rng.standard_normal((24, 512)).astype(np.float32))
    t = torch.from_numpy(rng.standard_normal((24, 512)).astype(np.float32))
    idx = torch.from_numpy(g["vtc_idx"])
    loss, sim, dv, dt, dtemp = ops.vtc_loss_fwd_bwd(v.to(DEV), t.to(DEV), idx.to(DEV), 0.07)
    assert rel(sim, torch.from_numpy(g["vtc_sim_v2t"])) < 1e-5
    assert abs(loss.item() - g["vtc_loss"][0]) / g["vtc_loss"][0] < 1e-5
    assert rel(dv, torch.from_numpy(g["vtc_grad_v"])) < 1e-4 and rel(dt, torch.from_numpy(g["vtc_grad_t"])) < 1e-4
    loss2, *_ = ops.vtc_loss_fwd_bwd(v.to(DEV), t.to(DEV), None, 0.07, want_grad=False)
    assert abs(loss2.item() - g["vtc_loss"][1]) / g["vtc_loss"][1] < 1e-5
    tt = torch.tensor(0.07, requires_grad=True)
    O.vtc_loss(v, t, idx, tt).backward()
    assert abs(dtemp.item() - tt.grad.item()) / abs(tt.grad.item()) < 1e-4


def test_stage2_module_vtc_loss_autograd_matches_reference_golden():
    """internvideo_amd.stage2.VTC_VTM_Loss.vtc_loss (criterions.py:65-103 signature) with autograd, single rank."""
    from internvideo_amd import stage2
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "tables.npz"))
    rng = np.random.Generator(np.random.PCG64(5))
    v = torch.from_numpy(rng.standard_normal((24, 512)).astype(np.float32)).to(DEV).requires_grad_(True)
    t = torch.from_numpy(rng.standard_normal((24, 512)).astype(np.float32)).to(DEV).requires_grad_(True)
    idx = torch.from_numpy(g["vtc_idx"]).to(DEV)
    temp = torch.tensor(0.07, device=DEV, requires_grad=True)
    crit = stage2.VTC_VTM_Loss(False)
    loss = crit.vtc_loss(v, t, idx, temp, all_gather=True)          # world size 1: gather is the identity
    assert abs(loss.item() - g["vtc_loss"][0]) / g["vtc_loss"][0] < 1e-5
    (loss * 2.0).backward()
    assert rel(v.grad / 2, torch.from_numpy(g["vtc_grad_v"])) < 1e-4 and rel(t.grad / 2, torch.from_numpy(g["vtc_grad_t"])) < 1e-4
    assert temp.grad is not None and torch.isfinite(temp.grad).item()
    s1, s2 = stage2.get_sim(v.detach(), t.detach(), 0.07)
    assert rel(s1, torch.from_numpy(g["vtc_sim_v2t"])) < 1e-5 and torch.equal(s2, s1.T)
    m = crit.get_mask(s1, idx, normalize=True)
    assert torch.allclose(m.sum(1), torch.ones(24, device=DEV))


def test_gemm256_race_screen_and_agreement_with_128():
    """The 256^2 kernel reads LDS pieces that LDS-DMA requests fill asynchronously (counted vmcnt + barriers): repeated runs on
    a multi-round grid must be bitwise reproducible and agree with the independent 128^2 kernel (same fp32 accumulation
    order inside a K step is NOT guaranteed, so agreement is to bf16 rounding)."""
    M, N, K = 13344, 1408, 1408
    A = bf(randn(M, K, seed=21)); W = bf(randn(N, K, seed=22, scale=0.05)); dY = bf(randn(M, N, seed=23))
    try:
        ops.set_gemm_kernel(1)
        ref_f = ops.gemm(A, W); ref_d = ops.gemm(dY, W, a_kc=True, b_kc=False); ref_w = ops.gemm(dY, A, a_kc=False, b_kc=False)
        ops.set_gemm_kernel(2)
        outs = [(ops.gemm(A, W), ops.gemm(dY, W, a_kc=True, b_kc=False), ops.gemm(dY, A, a_kc=False, b_kc=False)) for _ in range(6)]
    finally:
        ops.set_gemm_kernel(0)
    for o in outs[1:]:
        assert all(torch.equal(x, y) for x, y in zip(o, outs[0]))
    for got, ref in zip(outs[0], (ref_f, ref_d, ref_w)):
        assert rel(got.float(), ref.float()) < 3e-3


def _no_split():
    from internvideo_amd import lib
    return lib.load().ivh_gemm256_debug_split


@pytest.mark.parametrize("M,N,K", [(13344, 1408, 6144), (13184, 1416, 4224), (2048, 2048, 4096), (1800, 1024, 6144), (13344, 1408, 5400)])
def test_gemm256_tail_split_matches_unsplit(M, N, K):
    """a last tile round that is at most half full is cut into K slices that meet through a workspace (gemm256.hip, SPLIT; long K only): same
    values as the unsplit launch up to the fp32 summation order (bf16 rounding), bitwise reproducible, every epilogue flavour, both operand
    layouts, ragged M / N / K, fewer tiles than CUs"""
    from internvideo_amd import lib
    import ctypes as C
    A = bf(randn(M, K, seed=31)); W = bf(randn(N, K, seed=32, scale=0.05)); bias = randn(N, seed=33)
    Wk = W.t().contiguous()                                                           # [K, N]: the rows-contiguous B of a dgrad
    dact = bf(randn(M, N, seed=36))
    d = lib.GemmDesc(); d.M, d.N, d.K, d.a_kc, d.b_kc, d.batch, d.lda, d.ldb, d.ldc = M, N, K, 1, 1, 1, K, K, N
    split = _no_split()
    try:
        ops.set_gemm_kernel(2)
        assert lib.load().ivh_gemm_split_workspace(C.byref(d)) > 0, "this shape is meant to take the split path"

        def run():
            f = ops.gemm(A, W, bias=bias)                                             # forward NT, bias
            g, gd = ops.gemm(A, W, bias=bias, act="gelu_erf_d", want_preact=True)     # GELU + derivative copy
            dg = ops.gemm(A, Wk, a_kc=True, b_kc=False)                               # dgrad layout, same product
            dd, part = ops.gemm(A, Wk, a_kc=True, b_kc=False, dact_in=dact, act="gelu_erf_d", want_colsum=True)
            return f, g, gd, dg, dd, (ops.colsum_finish(part) if part is not None else None)
        got = run()
        again = run()
        split(0)
        ref = run()
    finally:
        split(1)
        ops.set_gemm_kernel(0)
    for x, y in zip(got, again):
        assert (x is None and y is None) or torch.equal(x, y)
    for x, y in zip(got[:5], ref[:5]):
        assert rel(x.float(), y.float()) < 1.5e-3
        assert ((x.float() - y.float()).abs() <= 8e-3 * y.float().abs() + 1e-3).all()            # one bf16 ulp
    assert got[5] is not None and rel(got[5], ref[5]) < 1e-4
    prod = A.float() @ W.float().t()
    assert rel(got[0].float(), prod + bias) < 4e-3 and rel(got[3].float(), prod) < 4e-3
    assert rel(got[4].float(), prod * dact.float()) < 5e-3


def test_gemm_grouped_matches_individual_launches():
    """ivh_gemm_grouped_bf16: four wgrad-shaped problems (different M, N, leading dimensions, same K) in one persistent launch
    == the same problems launched one by one, bit for bit (each tile is computed by one workgroup either way)."""
    K = 1336                                                     # K tail: 20.875 K steps
    shapes = [(528, 264), (176, 176), (768, 176), (176, 768)]    # (M_out, N_out)
    probs, refs = [], []
    for i, (Mo, No) in enumerate(shapes):
        dy = bf(randn(K, Mo, seed=40 + i)); x = bf(randn(K, No, seed=50 + i))
        out = torch.full((Mo, No), float("nan"), dtype=torch.bfloat16, device=DEV)
        probs.append((dy, x, out))
        refs.append(dy.float().t() @ x.float())
    try:
        ops.set_gemm_kernel(2)
        singles = [ops.gemm(dy, x, a_kc=False, b_kc=False) for dy, x, _ in probs]
        ops.gemm_grouped(probs, a_kc=False, b_kc=False)
    finally:
        ops.set_gemm_kernel(0)
    for (dy, x, out), single, ref in zip(probs, singles, refs):
        assert torch.equal(out, single)
        assert rel(out.float(), ref) < 4e-3
    # 28 problems (seven ViT-B/14-sized blocks) in one launch, and 40 (two launches): the per-tile problem lookup walks all 32 slots
    for nprob in (28, 40):
        many, ref_out = [], []
        for i in range(nprob):
            Mo, No = shapes[i % 4]
            dy = bf(randn(K, Mo, seed=100 + i)); x = bf(randn(K, No, seed=200 + i))
            many.append((dy, x, torch.full((Mo, No), float("nan"), dtype=torch.bfloat16, device=DEV)))
        try:
            ops.set_gemm_kernel(2)
            ref_out = [ops.gemm(dy, x, a_kc=False, b_kc=False) for dy, x, _ in many]
            ops.gemm_grouped(many, a_kc=False, b_kc=False)
        finally:
            ops.set_gemm_kernel(0)
        for (dy, x, out), single in zip(many, ref_out):
            assert torch.equal(out, single)
    # not groupable (a K-contiguous layout): falls back to one launch per problem, same results
    A = bf(randn(300, 136, seed=60)); W = bf(randn(264, 136, seed=61))
    o1 = torch.empty((300, 264), dtype=torch.bfloat16, device=DEV); o2 = torch.empty((300, 264), dtype=torch.bfloat16, device=DEV)
    ops.gemm_grouped([(A, W, o1), (A, W, o2)], a_kc=True, b_kc=True)
    assert torch.equal(o1, ops.gemm(A, W)) and torch.equal(o1, o2)


def test_gemm256_dgrad_epilogue_column_sums():
    """the fc2 dgrad epilogue of the 256^2 kernel (C = (dY W2) * gelu', act = 3) also returns the column sums of C per 128-row
    block: the fc1 bias gradient.  Ragged M and N."""
    M, N, K = 13344 // 8 + 8, 1416, 264          # 1676 rows: 6.55 row tiles; 1416 cols: 5.53 col tiles
    dY = bf(randn(M, K, seed=71)); W2 = bf(randn(K, N, seed=72, scale=0.1)); d = bf(randn(M, N, seed=73))
    try:
        ops.set_gemm_kernel(2)
        got, part = ops.gemm(dY, W2, a_kc=True, b_kc=False, dact_in=d, act="gelu_erf_d", want_colsum=True)
    finally:
        ops.set_gemm_kernel(0)
    assert part is not None and part.shape == (2 * ((M + 255) // 256), N)
    ref = (dY.float() @ W2.float()) * d.float()
    assert rel(got.float(), ref) < 5e-3
    cs = ops.colsum_finish(part)
    assert rel(cs, ref.sum(0)) < 2e-3              # sums of the fp32 values before the bf16 rounding of C
    # not available on the 128^2 kernel / without dact_in: part is None and nothing changes
    _, none = ops.gemm(dY, W2, a_kc=True, b_kc=False, want_colsum=True)
    assert none is None


def test_gemm_operands_of_2_gib_and_more():
    """the 256^2 kernel addresses through 32-bit buffer offsets; ivh_gemm_bf16 splits a larger forward / dgrad problem into row blocks
    and sends a weight gradient whose token dimension exceeds 2 GiB to the 128^2 kernel.  Checked on row / column samples against
    torch.matmul in fp32 (the full reference product would need 10 GiB more)."""
    from internvideo_amd import ops
    M, K, N = 1_100_000, 1024, 512                             # A: 2.25 GB
    g = torch.Generator(device="cuda").manual_seed(0)
    a = (torch.rand((M, K), device=DEV, generator=g) - 0.5).to(torch.bfloat16)
    w = (torch.rand((N, K), device=DEV, generator=g) - 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV, generator=g)
    y = ops.gemm(a, w, bias=bias)
    rows = torch.tensor([0, 1, 255, 256, 1_048_575, 1_048_576, 1_048_831, 1_048_832, M - 2, M - 1], device=DEV)
    ref = a[rows].float() @ w.float().T + bias
    assert rel(y[rows], ref) < 5e-3
    dy = (torch.rand((M, N), device=DEV, generator=g) - 0.5).to(torch.bfloat16)
    dx = ops.gemm(dy, w, a_kc=True, b_kc=False)                 # [M, K] = dy W: output 2.25 GB
    assert rel(dx[rows], dy[rows].float() @ w.float()) < 5e-3
    del y, dx
    dw = ops.gemm(dy, a, a_kc=False, b_kc=False)                # [N, K] = dy^T a: both operands rows-contiguous, K = M tokens
    cols = torch.arange(0, K, 37, device=DEV)
    ref = dy.float().T @ a[:, cols].float()
    assert rel(dw[:, cols], ref) < 5e-3


@pytest.mark.parametrize("M,C", [(37, 128), (203, 1408), (90, 3200), (1300, 3200)])
@pytest.mark.parametrize("xdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("two_heads", [False, True])
def test_layernorm_fwd_bwd(M, C, xdtype, two_heads):
    """LayerNorm with one or two affine heads on shared statistics (norms.hip layernorm_*): rows of one wave (C <= 2048) and rows shared
    by the four waves of a workgroup (C = 3200), against torch.nn.functional.layer_norm in fp32"""
    from internvideo_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + C)
    x = torch.randn(M, C, generator=g).to(DEV).to(xdtype)
    w = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV); b = (0.1 * torch.randn(C, generator=g)).to(DEV)
    w2 = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV) if two_heads else None
    b2 = (0.1 * torch.randn(C, generator=g)).to(DEV) if two_heads else None
    dy = torch.randn(M, C, generator=g).to(DEV).bfloat16()
    dy2 = torch.randn(M, C, generator=g).to(DEV).bfloat16() if two_heads else None
    y, y2, stats = ops.layernorm_fwd(x, w, b, 1e-6, w2, b2)
    base = torch.randn(M, C, generator=g).to(DEV)
    dx_acc = base.clone()
    dx, dw, db, dw2, db2 = ops.layernorm_bwd(x, w, stats, dy, w2, dy2)
    ops.layernorm_bwd(x, w, stats, dy, w2, dy2, dx=dx_acc, accumulate=True)
    xr = x.float().requires_grad_(True)
    ps = [t.clone().requires_grad_(True) for t in (w, b)] + ([t.clone().requires_grad_(True) for t in (w2, b2)] if two_heads else [])
    yr = torch.nn.functional.layer_norm(xr, (C,), ps[0], ps[1], 1e-6)
    loss = (yr * dy.float()).sum()
    if two_heads:
        yr2 = torch.nn.functional.layer_norm(xr, (C,), ps[2], ps[3], 1e-6)
        loss = loss + (yr2 * dy2.float()).sum()
        assert rel(y2, yr2) < 4e-3
    loss.backward()
    assert rel(y, yr) < 4e-3
    assert rel(dx, xr.grad) < 1e-4 and rel(dx_acc - base, xr.grad) < 1e-3
    assert rel(dw, ps[0].grad) < 1e-4 and rel(db, ps[1].grad) < 1e-4
    if two_heads:
        assert rel(dw2, ps[2].grad) < 1e-4 and rel(db2, ps[3].grad) < 1e-4


def test_frame_level_contrastive_features_match_reference_golden():
    """criterions.py:31-50: 3-D features -- vision [B, L, C] against text [B, C] and text [B, L, C] against vision [B, C] -- with agg_method
    "mean" / "max" (refused by the round-5 mirror): similarities, vtc_loss with and without idx, and the gradients of both feature tensors
    against the reference's own criterions.py (tests/golden/vtc3d.npz, make_golden_vtc3d.py).  fp32 logits kernel: 1e-5 / 1e-4."""
    from internvideo_amd import stage2 as S2
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vtc3d.npz"))
    crit = S2.VTC_VTM_Loss(False)
    idx = torch.from_numpy(g["idx"]).to(DEV)
    temp = torch.tensor(0.07, device=DEV)
    for tag, vk, tk in (("vis", "v3", "t2"), ("txt", "v2", "t3")):
        for agg in ("mean", "max"):
            k = f"{tag}:{agg}:"
            v = torch.from_numpy(g[vk]).to(DEV).requires_grad_(True)
            t = torch.from_numpy(g[tk]).to(DEV).requires_grad_(True)
            s1, s2 = S2.get_sim(v, t, temp, agg_method=agg)
            assert rel(s1, torch.from_numpy(g[k + "sim_v2t"])) < 1e-5 and rel(s2, torch.from_numpy(g[k + "sim_t2v"])) < 1e-5, k
            loss = crit.vtc_loss(v, t, idx, temp, all_gather=False, agg_method=agg)
            assert abs(loss.item() - g[k + "loss"][0]) < 1e-5 * g[k + "loss"][0], k
            loss.backward()
            assert rel(v.grad, torch.from_numpy(g[k + "grad_v"])) < 1e-4 and rel(t.grad, torch.from_numpy(g[k + "grad_t"])) < 1e-4, k
            l2 = crit.vtc_loss(v.detach(), t.detach(), None, temp, all_gather=False, agg_method=agg)
            assert abs(l2.item() - g[k + "loss"][1]) < 1e-5 * g[k + "loss"][1], k
