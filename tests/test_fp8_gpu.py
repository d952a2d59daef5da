"""fp8 (OCP e4m3fn) path on a real MI355X -- BASELINE configs[4] ("InternVideo2-6B encoder ... fp8 MFMA"):
  * ivh_fp8_quantize bit-exact against torch's own float8_e4m3fn cast of x / scale (plain and transposed copies, zero pad columns);
  * ivh_gemm_fp8 (v_mfma_scale_f32_16x16x128_f8f6f4, unit block scales) against an fp32 matmul of the DEQUANTISED operands: the
    only difference is fp32 summation order (tolerance 2e-5 relative), including ragged M / N, K tails and every epilogue;
  * Fp8LinearFn forward / dgrad / wgrad at the 6B width (3200 -> 9600) against the fp32 Linear on the same bf16 inputs, tolerance
    6e-2 relative (e4m3 keeps 3 mantissa bits: ~3.6 % per product, which a random-sign sum does not average away)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from internvideo_amd import ops  # noqa: E402
from internvideo_amd import functional as Fn  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a = a.double().cpu(); b = b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def randn(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


@pytest.mark.parametrize("M,K", [(64, 64), (130, 176), (417, 3200), (33, 1408), (256, 16)])
def test_fp8_quantize_is_bit_exact(M, K):
    x = randn(M, K, seed=M + K, scale=3.0).to(torch.bfloat16)
    x[0, 0] = 100.0                                            # an outlier sets the scale
    q, qt, scale = ops.fp8_quantize(x, want_transposed=True)
    amax = x.float().abs().max()
    assert abs(scale.item() - (amax / 448.0).item()) <= 1e-6 * scale.item()
    inv = 448.0 / amax
    want = (x.float() * inv).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    assert torch.equal(q.view(torch.uint8), want.view(torch.uint8)), (q.float() - want.float()).abs().max().item()
    M16 = (M + 15) // 16 * 16
    assert tuple(qt.shape) == (K, M16)
    assert torch.equal(qt[:, :M].view(torch.uint8), want.t().contiguous().view(torch.uint8))
    assert qt[:, M:].view(torch.uint8).abs().max().item() == 0 if M16 > M else True


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (130, 264, 176), (417 * 2, 1408, 1408), (300, 9600, 3200), (64, 8, 16), (1000, 96, 6144)])
def test_gemm_fp8_matches_fp32_on_dequantised_operands(M, N, K):
    a = randn(M, K, seed=1).to(torch.bfloat16); b = randn(N, K, seed=2, scale=0.05).to(torch.bfloat16)
    aq, _, sa = ops.fp8_quantize(a); bq, _, sb = ops.fp8_quantize(b)
    ref = (aq.float() * sa) @ (bq.float() * sb).t()
    out = ops.gemm_fp8(aq, bq, sa, sb, out_fp32=True)
    assert rel(out, ref) < 2e-5, rel(out, ref)
    out16 = ops.gemm_fp8(aq, bq, sa, sb)
    assert rel(out16.float(), ref) < 4e-3


def test_gemm_fp8_epilogues_and_transposed_contraction():
    M, N, K = 200, 256, 320
    a = randn(M, K, seed=3).to(torch.bfloat16); b = randn(N, K, seed=4, scale=0.1).to(torch.bfloat16); bias = randn(N, seed=5)
    aq, aqt, sa = ops.fp8_quantize(a, want_transposed=True); bq, bqt, sb = ops.fp8_quantize(b, want_transposed=True)
    A, Bm = aq.float() * sa, bq.float() * sb
    pre = A @ Bm.t() + bias
    g, u = ops.gemm_fp8(aq, bq, sa, sb, bias=bias, act="gelu_erf", want_preact=True)
    assert rel(g.float(), torch.nn.functional.gelu(pre)) < 4e-3 and rel(u.float(), pre) < 4e-3
    d = randn(M, N, seed=6).to(torch.bfloat16)
    y = ops.gemm_fp8(aq, bq, sa, sb, dact_in=d, act="gelu_erf_d", out_fp32=True)
    assert rel(y, (A @ Bm.t()) * d.float()) < 1e-4
    # contraction over the row axis through the transposed copies: A^T A-like products (what wgrad does), pad columns are zeros
    c = ops.gemm_fp8(aqt, aqt, sa, sa, out_fp32=True)                        # [K, K] = A^T A
    assert rel(c, A.t() @ A) < 2e-5


def test_fp8_linear_forward_dgrad_wgrad_at_the_6B_width():
    M, K, N = 2 * 833, 3200, 9600                                              # 2 clips of the masked 16-frame sequence, qkv of the 6B block
    x = randn(M, K, seed=7).to(torch.bfloat16).requires_grad_(True)
    w = torch.nn.Parameter(randn(N, K, seed=8, scale=0.02)); b = torch.nn.Parameter(randn(N, seed=9, scale=0.02))
    y = Fn.Fp8LinearFn.apply(x, w, b)
    dy = randn(M, N, seed=10).to(torch.bfloat16)
    y.backward(dy)
    xr = x.detach().float().requires_grad_(True); wr = w.detach().clone().requires_grad_(True); br = b.detach().clone().requires_grad_(True)
    yr = xr @ wr.t() + br
    yr.backward(dy.float())
    e = dict(y=rel(y.float(), yr.detach()), dx=rel(x.grad.float(), xr.grad), dw=rel(w.grad, wr.grad), db=rel(b.grad, br.grad))
    assert e["y"] < 6e-2 and e["dx"] < 6e-2 and e["dw"] < 6e-2 and e["db"] < 5e-3, e
    assert e["y"] > 1e-3                                                       # it really is the fp8 path, not a bf16 detour
    # the weight cache follows the parameter (version counter / engine epoch)
    q0 = Fn.fp8_weight(w)[0]
    with torch.no_grad():
        w.mul_(2.0)
    assert not torch.equal(Fn.fp8_weight(w)[2], torch.zeros(1, device=DEV)) and Fn.fp8_weight(w)[0] is not q0
