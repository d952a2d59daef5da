#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "rmsnorm or ln_l2 or layernorm" 2>&1 | tail -12 > gpurun_out/call17_tests.log
cat gpurun_out/call17_tests.log
timeout 300 python - <<'PY' 2>&1 | tail -12 | tee gpurun_out/call17_wide_rows.jsonl
import json, torch, sys
sys.path.insert(0, '.')
from internvideo_amd import ops
from tools.bench_kernels import rnd, timeit
DEV = "cuda"
for B, L, D in ((16, 833, 3200), (128, 417, 3200)):
    M = B * L
    x = rnd(M, D); r = torch.randn(M, D, device=DEV); g = torch.ones(D, device=DEV); wv = torch.ones(D, device=DEV)
    t = timeit(lambda: ops.rmsnorm_add_fwd(r, x, g, None, L, wv, 1e-6))
    print(json.dumps(dict(kernel="rmsnorm_add_fwd", M=M, D=D, us=round(t * 1e6, 1), gbps=round(M * D * 12 / t / 1e9, 1))))
    ro, y, rstd = ops.rmsnorm_add_fwd(r, x, g, None, L, wv, 1e-6)
    dres = torch.randn(M, D, device=DEV)
    t = timeit(lambda: ops.rmsnorm_add_bwd(y, dres, ro, rstd, wv, x, g, None, L))
    print(json.dumps(dict(kernel="rmsnorm_add_bwd", M=M, D=D, us=round(t * 1e6, 1), gbps=round(M * D * 18 / t / 1e9, 1))))
    if B == 16:
        qkv = rnd(M, 3 * D)
        t = timeit(lambda: ops.qk_rmsnorm_fwd(qkv, wv, wv, 1e-6))
        print(json.dumps(dict(kernel="qk_rmsnorm_fwd", M=M, D=D, us=round(t * 1e6, 1), gbps=round(M * D * 8 / t / 1e9, 1))))
        rq, rk = ops.qk_rmsnorm_fwd(qkv, wv, wv, 1e-6)
        dqkv = rnd(M, 3 * D)
        t = timeit(lambda: ops.qk_rmsnorm_bwd(qkv, dqkv, wv, wv, rq, rk))
        print(json.dumps(dict(kernel="qk_rmsnorm_bwd", M=M, D=D, us=round(t * 1e6, 1), gbps=round(M * D * 12 / t / 1e9, 1))))
    else:
        yy = rnd(M, D); tg = rnd(M, D); b = torch.zeros(D, device=DEV)
        t = timeit(lambda: ops.ln_l2_fwd(yy, wv, b, 1e-6, want_out=False, target=tg))
        print(json.dumps(dict(kernel="ln_l2_fwd(loss)", M=M, C=D, us=round(t * 1e6, 1), gbps=round(M * D * 4 / t / 1e9, 1))))
        _, stats, _ = ops.ln_l2_fwd(yy, wv, b, 1e-6, want_out=False, target=tg)
        t = timeit(lambda: ops.ln_l2_bwd(yy, wv, b, stats, None, tg, -2.0 / M))
        print(json.dumps(dict(kernel="ln_l2_bwd", M=M, C=D, us=round(t * 1e6, 1), gbps=round(M * D * 6 / t / 1e9, 1))))
PY
