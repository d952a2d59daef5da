import sys, os, torch
sys.path.insert(0, os.getcwd())
from internvideo_amd import ops
from tools.bench_kernels import rnd, timeit
M, D = 128 * 417, 1408
qkv = rnd(M, 3 * D); dqkv = rnd(M, 3 * D)
w = torch.ones(D, device="cuda") * 1.1
rq, rk = ops.qk_rmsnorm_fwd(qkv, w, w, 1e-6)
t = timeit(lambda: ops.qk_rmsnorm_bwd(qkv, dqkv, w, w, rq, rk))
print("qk_rmsnorm_bwd B=128 us", round(t * 1e6, 1), "GB/s", round(M * D * 12 / t / 1e9))
t = timeit(lambda: ops.qk_rmsnorm_fwd(qkv, w, w, 1e-6))
print("qk_rmsnorm_fwd B=128 us", round(t * 1e6, 1), "GB/s", round(M * D * 8 / t / 1e9))
