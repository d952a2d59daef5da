"""Is the 256x256 GEMM power / clock limited?  Runs the fwd_fc1 GEMM of the B = 128 step back to back for a few seconds per ablation
mode (tools/bench_gemm_bound.py) while sampling rocm-smi (sclk, socket power) from a side thread.  GPU box only."""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import lib, ops  # noqa: E402
from tools.bench_kernels import rnd  # noqa: E402


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10)
            d = json.loads(r.stdout)
            c = d.get("card0", {})
            out.append({k: v for k, v in c.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower()})
        except Exception as e:  # noqa: BLE001
            out.append({"error": str(e)[:80]})
        time.sleep(0.3)


def main():
    L = lib.load()
    M, D, Hm = 128 * 417, 1408, 6144
    a, b = rnd(M, D), rnd(Hm, D)
    out = torch.empty((M, Hm), dtype=torch.bfloat16, device="cuda")
    ops.set_gemm_kernel(2)
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    for mode, name in ((0, "full"), (1, "no_mfma"), (2, "no_dma"), (3, "no_ds_read"), (0, "full_again")):
        L.ivh_gemm256_debug_ablate(mode)
        stop, samples = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, samples))
        th.start()
        t0 = time.time()
        n = 0
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        while time.time() - t0 < secs:
            for _ in range(50):
                ops.gemm(a, b, out=out)
            n += 50
            torch.cuda.synchronize()
        e.record()
        torch.cuda.synchronize()
        stop.set(); th.join()
        us = s.elapsed_time(e) * 1e3 / n
        print(json.dumps(dict(mode=name, us_per_gemm=round(us, 1), launches=n, samples=samples[1:-1][:8])), flush=True)
    L.ivh_gemm256_debug_ablate(0)
    # the known-rate MFMA streams, sustained
    for shape in (0, 1):
        stop, samples = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, samples))
        th.start()
        t0 = time.time()
        fl = 0.0
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        while time.time() - t0 < secs:
            for _ in range(10):
                fl += ops.probe_mfma_rate2(shape, 2, 100000)
            torch.cuda.synchronize()
        e.record()
        torch.cuda.synchronize()
        stop.set(); th.join()
        print(json.dumps(dict(mode="mfma_stream_" + ("32x32x16" if shape == 0 else "16x16x32"), tflops=round(fl / (s.elapsed_time(e) * 1e-3) / 1e12, 1),
                              samples=samples[1:-1][:8])), flush=True)


if __name__ == "__main__":
    main()
