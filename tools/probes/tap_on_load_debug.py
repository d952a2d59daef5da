"""where do the dres_extra kernels differ from the definition?  (GPU box; debugging aid of round 5)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from internvideo_amd import ops
DEV = "cuda"
def randn(*s, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed); return torch.randn(*s, device=DEV, generator=g)
for (M, D, rps) in [(20, 3200, 10), (40, 3200, 10), (20, 3200, 20)]:
    bf = lambda x: x.bfloat16()
    res_out = bf(randn(M, D, seed=1)); branch = bf(randn(M, D, seed=2)); gamma = 1 + 0.1 * randn(D, seed=3)
    rowscale = torch.ones(M // rps, device=DEV); w = 1 + 0.1 * randn(D, seed=4)
    dy = bf(randn(M, D, seed=5)); dres = bf(randn(M, D, seed=6)); tap = bf(randn(M, D, seed=7))
    rstd = torch.rsqrt((res_out.float() ** 2).mean(-1) + 1e-6)
    keep = []
    for rep in range(3):
        d0 = dres.clone()
        dres_in, dbranch, dw, dg, db = ops.rmsnorm_add_bwd(dy, d0, res_out, rstd, w, branch, gamma, rowscale, rps, want_dbias=True, dres_extra=tap,
                                                           inplace_dres=False)
        torch.cuda.synchronize()
        keep.append((dres_in, dbranch))
        exp = gamma * dres_in.float()
        bad = ~torch.isfinite(dbranch.float()) | ((dbranch.float() - exp).abs() > 0.1)
        idx = bad.nonzero()
        rows = sorted(set(idx[:, 0].tolist()))
        print(M, D, rps, "rep", rep, "bad", int(bad.sum()), "rows", rows[:12], flush=True)
        if len(idx):
            r = rows[0]
            cols = idx[idx[:, 0] == r][:, 1].tolist()
            runs = []; s = cols[0]; p = cols[0]
            for c in cols[1:]:
                if c != p + 1: runs.append((s, p)); s = c
                p = c
            runs.append((s, p))
            print("  row", r, "bad col runs", runs[:12], "n", len(cols))
            c0 = cols[0]
            print("  got", dbranch[r, c0:c0 + 8].float().tolist()); print("  exp", exp[r, c0:c0 + 8].tolist())
            print("  gamma*(dres+xgrad only, no tap)?", (gamma * (dres_in.float() - tap.float()))[r, c0:c0 + 8].tolist())
            print("  dres_in", dres_in[r, c0:c0 + 8].float().tolist(), "tap", tap[r, c0:c0 + 8].float().tolist(), "dres", dres[r, c0:c0+8].float().tolist())
