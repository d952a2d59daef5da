"""qk_rmsnorm_bwd at the bench shape: run in two processes (IVH_QK_W4=0 / 1) and compare the dumps.  GPU box only."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import ops
M, D = int(sys.argv[2]), 1408
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.rand((M, 3 * D), device="cuda", generator=g) - 0.5).bfloat16()
wq = torch.rand(D, device="cuda", generator=g) + 0.5; wk = torch.rand(D, device="cuda", generator=g) + 0.5
rq, rk = ops.qk_rmsnorm_fwd(qkv, wq, wk, 1e-6)
d = (torch.rand((M, 3 * D), device="cuda", generator=g) - 0.5).bfloat16()
for it in range(3):
    dd = d.clone()
    dwq, dwk = ops.qk_rmsnorm_bwd(qkv, dd, wq, wk, rq, rk)
    torch.cuda.synchronize()
    print(it, "finite", bool(torch.isfinite(dd.float()).all()), bool(torch.isfinite(dwq).all()), "sum", dd.float().abs().sum().item(), dwq.sum().item(), dwk.sum().item())
torch.save(dict(dd=dd.cpu(), dwq=dwq.cpu(), dwk=dwk.cpu()), sys.argv[1])
