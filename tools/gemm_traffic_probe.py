"""What do the operand traffic beyond the L2, the C stores and the number of busy CUs cost the power-limited 256x256 GEMM?
Runs the fwd_fc1 GEMM of the B = 128 step (53376 x 6144 x 1408) back to back for a few seconds per variant while sampling rocm-smi:

  full            the shipped kernel, 256 workgroups
  alias           DBG 6: every tile loads the same 4 A + 2 B panels (L2 hits), same instruction stream and C stores
  alias_nostore   the same without the C stores
  nostore         the shipped loads, no C stores
  wg224 .. wg128  the shipped kernel on fewer workgroups (28 / 24 / 16 per XCD): throughput per CU when fewer CUs draw power

GPU box only.  python tools/gemm_traffic_probe.py [seconds per variant]"""
import json
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import lib, ops  # noqa: E402
from tools.bench_kernels import rnd  # noqa: E402
from tools.gemm_power_probe import sample  # noqa: E402


def main():
    L = lib.load()
    M, D, Hm = 128 * 417, 1408, 6144
    a, b = rnd(M, D), rnd(Hm, D)
    out = torch.empty((M, Hm), dtype=torch.bfloat16, device="cuda")
    ops.set_gemm_kernel(2)
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    flop = 2.0 * M * D * Hm
    variants = (("full", 0, 0, 0), ("alias", 6, 0, 0), ("alias_nostore", 6, 0, 1), ("nostore", 0, 0, 1), ("wg224", 0, 224, 0), ("wg192", 0, 192, 0),
                ("wg128", 0, 128, 0), ("full_again", 0, 0, 0))
    for name, dbg, wg, skip in variants:
        L.ivh_gemm256_debug_ablate(dbg)
        L.ivh_gemm256_debug_max_wg(wg)
        L.ivh_gemm256_debug(-1, skip)
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
        print(json.dumps(dict(mode=name, us_per_gemm=round(us, 1), tflops=round(flop / us / 1e6, 1), workgroups=wg or 256, launches=n,
                              samples=samples[1:-1][:6])), flush=True)
    L.ivh_gemm256_debug_ablate(0); L.ivh_gemm256_debug_max_wg(0); L.ivh_gemm256_debug(-1, 0)

    # ---- can an HBM-bound row kernel hide behind the power-limited GEMM when the GEMM leaves some CUs free? -------------------------------
    # stream A: 40 fc1 GEMMs (on `wg` workgroups); stream B: 110 rmsnorm_add_fwd launches (~ the same time alone).  `serial` = both alone.
    res = torch.randn((M, D), device="cuda")
    br = rnd(M, D)
    gamma = torch.ones(D, device="cuda"); w = torch.ones(D, device="cuda")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def run(wg, gemms, rows):
        L.ivh_gemm256_debug_max_wg(wg)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(sa):
            for _ in range(gemms):
                ops.gemm(a, b, out=out)
        with torch.cuda.stream(sb):
            for _ in range(rows):
                ops.rmsnorm_add_fwd(res, br, gamma, None, 417, w, 1e-6)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    for _ in range(2):
        run(0, 10, 30)
    for rep in range(2):
        g_alone = run(0, 40, 0)
        r_alone = run(0, 0, 110)
        line = dict(mode="overlap", rep=rep, gemm_alone_ms=round(g_alone, 2), rows_alone_ms=round(r_alone, 2))
        for wg in (256, 240, 224, 192):
            line[f"both_wg{wg}_ms"] = round(run(wg if wg != 256 else 0, 40, 110), 2)
        print(json.dumps(line), flush=True)
    L.ivh_gemm256_debug_max_wg(0)


if __name__ == "__main__":
    main()
