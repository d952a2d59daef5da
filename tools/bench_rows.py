"""Row-kernel / MFMA-shape experiments on the B = 128 shapes (M = 53376, D = 1408).  GPU box only.
  python tools/bench_rows.py probe      sustained rate of the two bf16 MFMA shapes, 1 / 2 waves per SIMD
  python tools/bench_rows.py rows       rmsnorm_add fwd / bwd, qk_rmsnorm fwd / bwd GB/s (IVH_BWD_PARTS selects the backward grid)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import ops  # noqa: E402
from tools.bench_kernels import rnd, timeit  # noqa: E402

DEV = "cuda"


def probe():
    for shape in (0, 1):
        for wps in (1, 2):
            iters = 400000 // wps
            for _ in range(2):
                ops.probe_mfma_rate2(shape, wps, iters)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                fl = ops.probe_mfma_rate2(shape, wps, iters)
                e.record()
                torch.cuda.synchronize()
                ts.append(s.elapsed_time(e) * 1e-3)
            t = sorted(ts)[len(ts) // 2]
            print(json.dumps(dict(probe="mfma_rate", shape="32x32x16" if shape == 0 else "16x16x32", waves_per_simd=wps, ms=round(t * 1e3, 2),
                                  tflops=round(fl / t / 1e12, 1), frac_of_2500=round(fl / t / 2.5e15, 4))), flush=True)


def rows():
    B, L, D = 128, 417, 1408
    M = B * L
    x = rnd(M, D)
    r = torch.randn(M, D, device=DEV); g = torch.ones(D, device=DEV); wv = torch.ones(D, device=DEV)
    out = []
    t = timeit(lambda: ops.rmsnorm_add_fwd(r, x, g, None, L, wv, 1e-6))
    out.append(dict(kernel="rmsnorm_add_fwd", us=t * 1e6, gbps=M * D * 12 / t / 1e9))
    ro, y, rstd = ops.rmsnorm_add_fwd(r, x, g, None, L, wv, 1e-6)
    dres = torch.randn(M, D, device=DEV)
    t = timeit(lambda: ops.rmsnorm_add_bwd(y, dres, ro, rstd, wv, x, g, None, L))
    out.append(dict(kernel="rmsnorm_add_bwd", us=t * 1e6, gbps=M * D * 18 / t / 1e9))
    qkv = rnd(M, 3 * D)
    t = timeit(lambda: ops.qk_rmsnorm_fwd(qkv, wv, wv, 1e-6))
    out.append(dict(kernel="qk_rmsnorm_fwd", us=t * 1e6, gbps=M * D * 8 / t / 1e9))
    rq, rk = ops.qk_rmsnorm_fwd(qkv, wv, wv, 1e-6)
    dqkv = rnd(M, 3 * D)
    t = timeit(lambda: ops.qk_rmsnorm_bwd(qkv, dqkv, wv, wv, rq, rk))
    out.append(dict(kernel="qk_rmsnorm_bwd", us=t * 1e6, gbps=M * D * 12 / t / 1e9))
    for o in out:
        o["bwd_parts"] = os.environ.get("IVH_BWD_PARTS", "512")
        print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in o.items()}), flush=True)


def rows16():
    """the residual kernels on a bf16 stream (model.residual_dtype = "bf16"); IVH_BWD_ROWS / IVH_BWD_PARTS select the backward's shape"""
    B, L, D = 128, 417, 1408
    M = B * L
    x = rnd(M, D)
    r = rnd(M, D); g = torch.ones(D, device=DEV); wv = torch.ones(D, device=DEV)
    out = []
    t = timeit(lambda: ops.rmsnorm_add_fwd(r, x, g, None, L, wv, 1e-6))
    out.append(dict(kernel="rmsnorm_add_fwd[bf16 stream]", us=t * 1e6, gbps=M * D * 8 / t / 1e9))
    ro, y, rstd = ops.rmsnorm_add_fwd(r, x, g, None, L, wv, 1e-6)
    dres = rnd(M, D)
    t = timeit(lambda: ops.rmsnorm_add_bwd(y, dres, ro, rstd, wv, x, g, None, L, want_dbias=True))
    out.append(dict(kernel="rmsnorm_add_bwd[bf16 stream]", us=t * 1e6, gbps=M * D * 12 / t / 1e9))
    for o in out:
        o["bwd_parts"] = os.environ.get("IVH_BWD_PARTS", "512"); o["bwd_rows"] = os.environ.get("IVH_BWD_ROWS", "1")
        print(json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in o.items()}), flush=True)


if __name__ == "__main__":
    {"probe": probe, "rows": rows, "rows16": rows16}[sys.argv[1]]()
