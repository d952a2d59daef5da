"""Round-5 prototype of the q/k-norm fusion, forward half: what does applying the q/k RMSNorm INSIDE the attention forward
cost, against the standalone pass it would remove?  B = 128, L = 417, 16 heads of 88 (the 1B block).

    baseline : qk_rmsnorm_fwd (in place on packed qkv) -> flash_attn_fwd_packed           (what the step runs)
    fused    : ivh_probe_attn32_fwd_qkn on the RAW qkv + rstd_q / rstd_k + q_norm.weight * k_norm.weight (Q scaled at load, per-key factor on S)

Checks the fused output against the baseline's (bf16 rounding of k_hat is the only difference), then times: the norm pass alone, the shipped
forward alone, the fused forward alone.  One JSON line."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from internvideo_amd import lib, ops  # noqa: E402
from internvideo_amd.lib import ptr, stream_ptr  # noqa: E402

DEV = "cuda"


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / iters * 1e3)
    return round(statistics.median(ts), 1)


def main():
    B, L, H, hd = (32, 417, 16, 88) if "--b32" in sys.argv else (128, 417, 16, 88)
    D = H * hd
    M = B * L
    g = torch.Generator(device=DEV).manual_seed(0)
    qkv = (torch.randn((M, 3 * D), device=DEV, generator=g) * 0.7).to(torch.bfloat16)
    wq = (1 + 0.2 * torch.randn(D, device=DEV, generator=g)).float()
    wk = (1 + 0.2 * torch.randn(D, device=DEV, generator=g)).float()
    work = qkv.clone()
    rq, rk = ops.qk_rmsnorm_fwd(work, wq, wk, 1e-6)
    out0, lse0 = ops.flash_attn_fwd_packed(work, B, L, H)
    out1 = torch.empty((M, D), dtype=torch.bfloat16, device=DEV)
    lse1 = torch.empty((B, H, L), dtype=torch.float32, device=DEV)
    wqk = (wq * wk).contiguous()
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + D * 2, qkv.data_ptr() + 2 * D * 2
    scale = float(hd ** -0.5)

    def fused():
        lib.call("ivh_probe_attn32_fwd_qkn", q, L * 3 * D, 3 * D, hd, k, v, L * 3 * D, 3 * D, hd, ptr(out1), L * D, D, hd, ptr(lse1), B, H, L, L, hd, scale,
                 ptr(rq), ptr(rk), ptr(wqk), stream_ptr())
    fused()
    torch.cuda.synchronize()
    rel = ((out1.float() - out0.float()).norm() / out0.float().norm()).item()
    rel_lse = ((lse1 - lse0).abs().max()).item()
    # fp32 torch reference on two clips
    x = qkv[:2 * L].float().reshape(2, L, 3, H, hd)
    def rms(t, w):
        t2 = t.reshape(2, L, D)
        return (t2 * torch.rsqrt(t2.pow(2).mean(-1, keepdim=True) + 1e-6) * w).reshape(2, L, H, hd)
    qn, kn, vv = rms(x[:, :, 0], wq), rms(x[:, :, 1], wk), x[:, :, 2]
    att = torch.softmax(torch.einsum("blhd,bmhd->bhlm", qn, kn) * scale, -1)
    ref = torch.einsum("bhlm,bmhd->blhd", att, vv).reshape(2 * L, D)
    e0 = ((out0[:2 * L].float() - ref).norm() / ref.norm()).item()
    e1 = ((out1[:2 * L].float() - ref).norm() / ref.norm()).item()
    t_norm = timed(lambda: ops.qk_rmsnorm_fwd(work, wq, wk, 1e-6))
    t_base = timed(lambda: ops.flash_attn_fwd_packed(work, B, L, H))
    t_fused = timed(fused)
    print(json.dumps(dict(B=B, L=L, H=H, hd=hd, fused_vs_baseline_rel=rel, lse_max_abs_diff=rel_lse, baseline_vs_fp32=e0, fused_vs_fp32=e1, qk_rmsnorm_fwd_us=t_norm,
                          attn_fwd_shipped_us=t_base, attn_fwd_qkn_us=t_fused, attn_slowdown_us=round(t_fused - t_base, 1),
                          net_forward_us_per_layer_before_the_gemm_epilogue_partials=round(t_fused - t_base - t_norm, 1))))


if __name__ == "__main__":
    main()
