"""ln_l2 forward / backward (the clip decoders' LayerNorm -> l2 -> cosine-loss tail) at the bench shape: M = 53376 rows x 3200.  GPU box only.
IVH_BWD_PARTS selects the backward grid."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import ops  # noqa: E402
from tools.bench_kernels import rnd, timeit  # noqa: E402

M, C = 128 * 417, 3200
y = rnd(M, C); w = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
tg = torch.nn.functional.normalize(torch.randn(M, C, device="cuda"), dim=-1).bfloat16()
t = timeit(lambda: ops.ln_l2_fwd(y, w, b, 1e-5, want_out=False, target=tg))
print(json.dumps(dict(kernel="ln_l2_fwd", us=round(t * 1e6, 1), gbps=round(M * C * 4 / t / 1e9, 1))))
_, stats, rows = ops.ln_l2_fwd(y, w, b, 1e-5, want_out=False, target=tg)
one = torch.ones(1, device="cuda")
t = timeit(lambda: ops.ln_l2_bwd(y, w, b, stats, None, tg, -2.0, dscale_dev=one))
print(json.dumps(dict(kernel="ln_l2_bwd", us=round(t * 1e6, 1), gbps=round(M * C * 6 / t / 1e9, 1), bwd_parts=os.environ.get("IVH_BWD_PARTS", "512"))))
# every row against torch autograd (fp32) on the same definition
dy, dw, db = ops.ln_l2_bwd(y, w, b, stats, None, tg, -2.0, dscale_dev=one)
worst, bad = 0.0, 0
for r0 in range(0, M, 8192):
    yy = y[r0:r0 + 8192].float().requires_grad_(True)
    o = torch.nn.functional.normalize(torch.nn.functional.layer_norm(yy, (C,), w, b, 1e-5), dim=-1)
    (o * (-2.0 * tg[r0:r0 + 8192].float())).sum().backward()
    d = (dy[r0:r0 + 8192].float() - yy.grad).abs()
    worst = max(worst, d.max().item()); bad += int((~torch.isfinite(dy[r0:r0 + 8192].float())).sum())
    scale = yy.grad.abs().max().item()
print(json.dumps(dict(check="ln_l2_bwd vs autograd, all rows", max_abs_err=worst, grad_abs_max=scale, nonfinite=bad, pf=os.environ.get("IVH_LNL2_PF", "0"))))
