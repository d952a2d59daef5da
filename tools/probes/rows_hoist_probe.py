"""Round-5 experiment: LayerScale gamma / norm weight columns kept in registers across a lane's rows (IVH_ROWS_HOIST) in the two forward row
kernels of the bf16 residual stream, on the 1B step's shapes.  GPU box only.
    IVH_ROWS_HOIST=<0 | 1 | grid> python tools/probes/rows_hoist_probe.py <dump.pt>
Prints one JSON line per kernel and saves the outputs so that two runs can be compared bit for bit (compare mode: two dump paths)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    if len(sys.argv) == 3:
        a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
        print(json.dumps({k: bool(torch.equal(a[k], b[k])) for k in a}))
        return
    from internvideo_amd import ops
    from tools.bench_kernels import timeit
    B, L, D = 128, 417, 1408
    M = B * L
    g = torch.Generator(device="cuda").manual_seed(3)
    rnd = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)   # noqa: E731
    res, br = rnd(M, D), rnd(M, D)
    gam, wv = torch.rand(D, device="cuda", generator=g), torch.rand(D, device="cuda", generator=g) + 0.5
    ro, y, rstd = ops.rmsnorm_add_fwd(res, br, gam, None, L, wv, 1e-6)
    t = timeit(lambda: ops.rmsnorm_add_fwd(res, br, gam, None, L, wv, 1e-6))
    print(json.dumps(dict(kernel="rmsnorm_add_fwd (bf16 stream)", hoist=os.environ.get("IVH_ROWS_HOIST", "0"), us=round(t * 1e6, 1), gbps=round(M * D * 8 / t / 1e9))), flush=True)
    qkv = rnd(M, 3 * D)
    q0 = qkv.clone()
    rq, rk = ops.qk_rmsnorm_fwd(q0, wv, gam + 0.5, 1e-6)
    t = timeit(lambda: ops.qk_rmsnorm_fwd(qkv, wv, gam + 0.5, 1e-6))
    print(json.dumps(dict(kernel="qk_rmsnorm_fwd", hoist=os.environ.get("IVH_ROWS_HOIST", "0"), us=round(t * 1e6, 1), gbps=round(M * D * 8 / t / 1e9))), flush=True)
    torch.save(dict(ro=ro.cpu(), y=y.cpu(), rstd=rstd.cpu(), q0=q0.cpu(), rq=rq.cpu(), rk=rk.cpu()), sys.argv[1])


if __name__ == "__main__":
    main()
