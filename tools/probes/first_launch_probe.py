import sys, os, torch
sys.path.insert(0, os.getcwd())
from internvideo_amd import ops
from tools.bench_kernels import rnd
for (m, n, k) in ((53248, 1408, 592), (53376, 1408, 1408), (53376, 4224, 1408)):
    a, w = rnd(m, k), rnd(n, k)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    ts = []
    for i in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm(a, w, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(round(e0.elapsed_time(e1) * 1e3, 1))
    print(m, n, k, os.environ.get("IVH_NO_HALF"), ts, flush=True)
