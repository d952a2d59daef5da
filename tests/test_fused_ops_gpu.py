"""The fused-op seam (internvideo_amd.fused_ops) on a real MI355X: each drop-in module against a plain torch
fp32 reference of the same op (forward and backward), and a reference-style fused Block (the control flow of
models/internvideo2_pretrain.py:198-210,279-287 written in the test) assembled from the three modules against the CPU oracle's block."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from internvideo_amd.fused_ops import DropoutAddRMSNorm, FlashAttention, FusedMLP  # noqa: E402
from internvideo_amd.lib import InternVideoHipError  # noqa: E402
from oracle import internvideo2_oracle as O  # noqa: E402

DEV = "cuda"


def rel(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("B,S,H,hd,dtype", [(2, 65, 2, 64, torch.bfloat16), (1, 417, 4, 88, torch.bfloat16), (2, 50, 2, 128, torch.float16)])
def test_flash_attention_module(B, S, H, hd, dtype):
    gen = torch.Generator().manual_seed(0)
    qkv = (torch.randn(B, S, 3, H, hd, generator=gen) * 0.7).to(dtype)
    ref_in = qkv.float().requires_grad_(True)
    q, k, v = (ref_in[:, :, i].transpose(1, 2) for i in range(3))
    ref = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(-1) @ v                     # (B,H,S,hd)
    ref = ref.transpose(1, 2)
    w = torch.randn(B, S, H, hd, generator=gen)
    (ref * w).sum().backward()
    x = qkv.to(DEV).requires_grad_(True)
    attn = FlashAttention(attention_dropout=0.0)
    out, none = attn(x, key_padding_mask=None, need_weights=False, causal=False)
    assert none is None and out.dtype == dtype and tuple(out.shape) == (B, S, H, hd)
    assert rel(out.float(), ref.detach()) < 1e-2
    (out.float() * w.to(DEV)).sum().backward()
    assert x.grad.dtype == dtype and rel(x.grad.float(), ref_in.grad) < 2e-2
    with pytest.raises(InternVideoHipError):
        attn(x, causal=True)
    with pytest.raises(AssertionError):
        attn(x.float())
    full = attn(x.detach(), key_padding_mask=torch.ones(B, S, dtype=torch.bool, device=DEV))[0]      # an all-ones mask changes nothing
    assert torch.equal(full, out.detach())


def test_flash_attention_module_right_padded_batches():
    """the key_padding_mask branch of FlashAttention.forward (flash_attention_class.py:51-62: unpad -> varlen kernel -> pad) for
    right-padded batches: keys beyond each sequence's length excluded, padded rows zero, no gradient into padded tokens."""
    gen = torch.Generator().manual_seed(5)
    B, S, H, hd = 3, 70, 2, 64
    lens = [70, 33, 1]
    qkv = (torch.randn(B, S, 3, H, hd, generator=gen) * 0.7).bfloat16()
    keep = torch.arange(S).unsqueeze(0) < torch.tensor(lens).unsqueeze(1)
    ref_in = qkv.float().requires_grad_(True)
    w = torch.randn(B, S, H, hd, generator=gen)
    outs = []
    for b, n in enumerate(lens):                                   # per-sequence dense attention on the valid prefix
        q, k, v = (ref_in[b, :n, i].transpose(0, 1) for i in range(3))
        o = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(-1) @ v
        outs.append(torch.cat([o.transpose(0, 1), torch.zeros(S - n, H, hd)], 0))
    ref = torch.stack(outs)
    (ref * w).sum().backward()
    x = qkv.to(DEV).requires_grad_(True)
    attn = FlashAttention()
    out, _ = attn(x, key_padding_mask=keep.to(DEV))
    assert rel(out.float(), ref.detach()) < 1e-2 and not out[1, 33:].any() and not out[2, 1:].any()
    (out.float() * w.to(DEV)).sum().backward()
    assert rel(x.grad.float(), ref_in.grad) < 2e-2
    assert not x.grad[1, 33:].any() and not x.grad[2, 1:].any()
    holes = keep.clone(); holes[0, 5] = False
    with pytest.raises(InternVideoHipError):
        attn(x, key_padding_mask=holes.to(DEV))


@pytest.mark.parametrize("activation", ["gelu_approx", "gelu"])
def test_fused_mlp_module(activation):
    torch.manual_seed(0)
    m = FusedMLP(176, 768, activation=activation).to(DEV)
    x = (torch.randn(3, 37, 176) * 0.5)
    ref_p = {k: v.detach().cpu().float().requires_grad_(True) for k, v in m.named_parameters()}
    xr = x.clone().requires_grad_(True)
    h = F.gelu(xr @ ref_p["fc1.weight"].t() + ref_p["fc1.bias"], approximate="tanh" if activation == "gelu_approx" else "none")
    ref = h @ ref_p["fc2.weight"].t() + ref_p["fc2.bias"]
    w = torch.randn_like(ref)
    (ref * w).sum().backward()
    xg = x.to(DEV).bfloat16().requires_grad_(True)
    y = m(xg)
    assert y.dtype == torch.bfloat16 and rel(y.float(), ref.detach()) < 1e-2
    (y.float() * w.to(DEV)).sum().backward()
    assert rel(xg.grad.float(), xr.grad) < 2e-2
    for k, p in m.named_parameters():
        assert rel(p.grad, ref_p[k].grad) < 2e-2, k


def test_dropout_add_rmsnorm_module():
    torch.manual_seed(0)
    D = 176
    norm = DropoutAddRMSNorm(D, eps=1e-6, prenorm=True).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.1 * torch.randn(D))
    x, r = torch.randn(4, 21, D), torch.randn(4, 21, D)
    wt = norm.weight.detach().cpu().clone().requires_grad_(True)

    def ref_fn(x, r):
        s = x + r if r is not None else x
        return s * torch.rsqrt(s.pow(2).mean(-1, keepdim=True) + 1e-6) * wt, s

    xr, rr = x.clone().requires_grad_(True), r.clone().requires_grad_(True)
    yr, sr = ref_fn(xr, rr)
    w1, w2 = torch.randn_like(yr), torch.randn_like(sr)
    ((yr * w1).sum() + (sr * w2).sum()).backward()
    xg, rg = x.to(DEV).bfloat16().requires_grad_(True), r.to(DEV).bfloat16().requires_grad_(True)
    y, s = norm(xg, rg)
    assert y.dtype == torch.bfloat16 and s.dtype == torch.bfloat16
    assert rel(y.float(), yr.detach()) < 1e-2 and rel(s.float(), sr.detach()) < 1e-2
    ((y.float() * w1.to(DEV)).sum() + (s.float() * w2.to(DEV)).sum()).backward()
    assert rel(xg.grad.float(), xr.grad) < 2e-2 and rel(rg.grad.float(), rr.grad) < 2e-2
    assert rel(norm.weight.grad, wt.grad) < 2e-2
    # first block: residual None -> (norm(x), x);  prenorm=False -> tensor only;  residual_in_fp32 keeps the fp32 stream
    y0, s0 = norm(xg.detach(), None)
    assert rel(y0.float(), ref_fn(x.bfloat16().float(), None)[0].detach()) < 1e-2 and torch.equal(s0.float().cpu(), x.bfloat16().float())
    n2 = DropoutAddRMSNorm(D, eps=1e-6, prenorm=False).to(DEV)
    assert isinstance(n2(xg.detach(), rg.detach()), torch.Tensor)
    n3 = DropoutAddRMSNorm(D, eps=1e-6, prenorm=True, residual_in_fp32=True).to(DEV)
    assert n3(xg.detach(), rg.detach())[1].dtype == torch.float32
    with pytest.raises(InternVideoHipError):
        DropoutAddRMSNorm(D, p=0.1)


def test_reference_style_fused_block_from_the_three_modules_matches_oracle():
    """P:198-210 (`_flash_attn` with fused q/k norm) + P:279-287 (fused residual protocol), the Linear layers in torch bf16."""
    cfg = O.named_config("tiny88")
    p = O.synthetic_params(cfg, seed=1)
    D, H = cfg.embed_dim, cfg.num_heads
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 21, D, generator=gen)
    with torch.no_grad():
        want = O.block(O.block(x, p, 0, cfg), p, 1, cfg)                        # two blocks: exercises the residual hand-over

    def norm(key):
        n = DropoutAddRMSNorm(D, eps=1e-6, prenorm=True).to(DEV)
        n.weight.data.copy_(p[key])
        return n
    attn_core = FlashAttention()
    h, residual = x.to(DEV).bfloat16(), None
    for i in range(2):
        pre = f"blocks.{i}."
        w = {k: p[pre + k].to(DEV).bfloat16() for k in ("attn.qkv.weight", "attn.proj.weight", "attn.proj.bias")}
        mlp = FusedMLP(D, cfg.mlp_hidden, activation="gelu").to(DEV)
        mlp.load_state_dict({k: p[pre + "mlp." + k] for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")})
        h, residual = norm(pre + "norm1.weight")(h, residual)                                   # P:283
        qkv = F.linear(h, w["attn.qkv.weight"]).view(2, 21, 3, H, D // H)                        # P:195-196
        q, k, v = qkv.unbind(2)
        q = norm(pre + "attn.q_norm.weight")(q.flatten(-2, -1))[0].view(q.shape)               # P:200
        k = norm(pre + "attn.k_norm.weight")(k.flatten(-2, -1))[0].view(k.shape)               # P:201
        ctx, _ = attn_core(torch.stack([q, k, v], dim=2))                                       # P:208-210
        a = F.linear(ctx.reshape(2, 21, D), w["attn.proj.weight"], w["attn.proj.bias"])
        a = (a.float() * p[pre + "ls1.gamma"].to(DEV)).bfloat16()                               # P:284 (LayerScale fp32)
        h, residual = norm(pre + "norm2.weight")(a, residual)                                   # P:285
        h = (mlp(h).float() * p[pre + "ls2.gamma"].to(DEV)).bfloat16()                          # P:286
    out = h.float() + residual.float()                                                          # P:685-688
    assert rel(out, want) < 1.5e-2                    # bf16 residual stream (residual_in_fp32=False, like flash_attn's default)


def test_torch_ops_dispatch_to_the_hip_kernels():
    """torch.ops.internvideo_hip.* (internvideo_amd/torch_ops.py) returns exactly what the ctypes wrappers return"""
    import internvideo_amd.torch_ops  # noqa: F401
    from internvideo_amd import ops
    ns = torch.ops.internvideo_hip
    g = torch.Generator(device="cuda").manual_seed(3)
    a = (torch.rand((200, 128), device="cuda", generator=g) - 0.5).bfloat16()
    w = (torch.rand((72, 128), device="cuda", generator=g) - 0.5).bfloat16()
    bias = torch.randn(72, device="cuda", generator=g)
    assert torch.equal(ns.gemm(a, w, bias, "gelu_erf"), ops.gemm(a, w, bias=bias, act="gelu_erf"))
    B, L, H = 3, 40, 2
    qkv = (torch.rand((B * L, 3 * 128), device="cuda", generator=g) - 0.5).bfloat16()
    o, lse = ns.flash_attn_fwd(qkv, B, L, H)
    o2, lse2 = ops.flash_attn_fwd_packed(qkv, B, L, H)
    assert torch.equal(o, o2) and torch.equal(lse, lse2)
    do = (torch.rand((B * L, 128), device="cuda", generator=g) - 0.5).bfloat16()
    assert torch.equal(ns.flash_attn_bwd(qkv, o, do, lse, B, L, H), ops.flash_attn_bwd_packed(qkv, o2, do, lse2, B, L, H))
    res = torch.randn((B * L, 128), device="cuda", generator=g)
    wn = torch.rand(128, device="cuda", generator=g) + 0.5
    r1, y1, s1 = ns.rmsnorm_add_fwd(res, o, None, None, L, wn, 1e-6)
    r2, y2, s2 = ops.rmsnorm_add_fwd(res, o, None, None, L, wn, 1e-6)
    assert torch.equal(r1, r2) and torch.equal(y1, y2) and torch.equal(s1, s2)
    labels = torch.randint(0, 72, (200,), device="cuda", generator=g)
    loss = ns.cross_entropy_rows(ns.gemm(a, w), labels, 72)
    ref = torch.nn.functional.cross_entropy(ops.gemm(a, w).float(), labels)
    assert abs(loss.item() - ref.item()) < 1e-4 * abs(ref.item())
