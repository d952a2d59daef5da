"""Round-6 pricing of the deferred rescale (IVH_ATTN_DEFER=1: O / l are rescaled only when some query's maximum rises by more than 8 in log2 units)
on the attention forward compiled without packed fp32.  Run twice (env IVH_ATTN_DEFER unset / 1); prints the forward time and the deviation
from the fp32 reference at the 1B shape.  GPU box only."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from internvideo_amd import ops  # noqa: E402
from tools.bench_kernels import rnd  # noqa: E402


def one(fn, iters=20):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


ops.set_attn_kernel(2)
res = dict(defer=os.environ.get("IVH_ATTN_DEFER", "0"))
for B, L, H, hd in [(128, 417, 16, 88), (112, 417, 16, 88), (8, 2049, 16, 88)]:
    qkv = rnd(B * L, 3 * H * hd)
    for _ in range(3):
        ops.flash_attn_fwd_packed(qkv, B, L, H)
    ts = [one(lambda: ops.flash_attn_fwd_packed(qkv, B, L, H)) for _ in range(7)]
    res[f"fwd_us_B{B}_L{L}"] = round(statistics.median(ts), 1)
B, L, H, hd = 4, 417, 16, 88
qkv = rnd(B * L, 3 * H * hd)
out, lse = ops.flash_attn_fwd_packed(qkv, B, L, H)
q, k, v = (qkv.float().view(B, L, 3, H, hd)[:, :, i].permute(0, 2, 1, 3) for i in range(3))
att = (q * hd ** -0.5) @ k.transpose(-1, -2)
ref = (att.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * L, H * hd)
res["rel_err_vs_fp32"] = float((out.float() - ref).norm() / ref.norm())
res["lse_max_err"] = float((lse - torch.logsumexp(att, -1)).abs().max())
print(json.dumps(res))
