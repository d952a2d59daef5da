"""AdamW / gradient-norm kernels at the 1B model's size (1.07e9 parameters, bf16 gradients, bf16 shadow weights): us per launch and TB/s for
the variants of optim.hip (IVH_ADAMW_VARIANT bit 0 = two groups per thread and trip, bit 1 = non-temporal accesses; IVH_ADAMW_BLOCKS = grid
cap).  One subprocess per setting (the switches are read once per process).  Usage: python tools/bench_adamw.py [--n N]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(n):
    import torch
    sys.path.insert(0, ROOT)
    from internvideo_amd import ops
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    master = torch.randn(n, device=dev, generator=g) * 0.02
    m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    grad = (torch.randn(n, device=dev, generator=g) * 1e-3).bfloat16()
    shadow = torch.empty(n, device=dev, dtype=torch.bfloat16)
    out = torch.zeros(1, device=dev)
    res = {}
    for name, fn, nbytes in (("adamw", lambda s: ops.adamw_step(master, m, v, grad, shadow, 1e-4, 0.9, 0.98, 1e-6, 0.05, s), n * 28),
                             ("sqnorm", lambda s: ops.sqnorm(grad, out, False), n * 2)):
        for s in range(1, 4):
            fn(s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in range(4, 14):
            fn(s)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0
        res[name] = dict(us=round(us, 1), tbps=round(nbytes / us / 1e6, 3))
    res["checksum"] = float(master.double().sum().item())          # identical for every variant: the arithmetic does not change
    print(json.dumps(res))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_070_000_000 // 4 * 4)
    ap.add_argument("--worker", action="store_true")
    a = ap.parse_args()
    if a.worker:
        return worker(a.n)
    for variant in (0, 1, 2, 3):
        for blocks in (4096, 8192, 16384, 32768):
            env = dict(os.environ, IVH_ADAMW_VARIANT=str(variant), IVH_ADAMW_BLOCKS=str(blocks))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", "--n", str(a.n)], env=env, capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            print(json.dumps(dict(variant=variant, blocks=blocks, **(json.loads(line[-1]) if line else {"error": r.stderr[-300:]}))), flush=True)


if __name__ == "__main__":
    main()
