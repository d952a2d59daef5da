"""The two epilogue-heavy GEMMs of a 1B block beside their plain twins (B = 128: 53376 rows; `--b32`: 13344), interleaved rounds, random
operands.  GPU box only.  fc1 forward = gelu(x W1^T + b) with the gelu' copy (gemm256 EPI 2); fc2 dgrad = (dy W2) * gelu' with the bias
column sums (EPI 3).  One JSON line per flavour: median / min us per launch and TFLOP/s."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from internvideo_amd import ops  # noqa: E402

DEV = "cuda"


def rnd(*shape, scale=1.0):
    return ((torch.rand(*shape, device=DEV) * 2 - 1) * scale).to(torch.bfloat16)


def timed(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    M = 13344 if "--b32" in sys.argv else 53376
    D, Hm = 1408, 6144
    x, w1, b1 = rnd(M, D), rnd(Hm, D, scale=0.05), torch.rand(Hm, device=DEV) - 0.5
    dy, w2 = rnd(M, D), rnd(D, Hm, scale=0.05)
    g, u = ops.gemm(x, w1, bias=b1, act="gelu_erf_d", want_preact=True)
    flavours = {
        "fc1_fwd_gelu_with_dgelu_copy(EPI2)": lambda: ops.gemm(x, w1, bias=b1, act="gelu_erf_d", want_preact=True),
        "fc1_fwd_plain_bias": lambda: ops.gemm(x, w1, bias=b1),
        "fc2_dgrad_times_dgelu_colsum(EPI3)": lambda: ops.gemm(dy, w2, a_kc=True, b_kc=False, dact_in=u, act="gelu_erf_d", want_colsum=True),
        "fc2_dgrad_plain": lambda: ops.gemm(dy, w2, a_kc=True, b_kc=False),
        # round 5: what the two heavy epilogues are made of
        "fc1_fwd_gelu_only_no_copy(EPI2,MODE2)": lambda: ops.gemm(x, w1, bias=b1, act="gelu_erf"),
        "fc2_dgrad_times_dgelu_no_colsum(EPI3)": lambda: ops.gemm(dy, w2, a_kc=True, b_kc=False, dact_in=u, act="gelu_erf_d"),
    }
    for f in flavours.values():
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    times = {k: [] for k in flavours}
    for _ in range(7):
        for k, f in flavours.items():
            times[k].append(timed(f, 10))
    fl = 2.0 * M * D * Hm
    for k, ts in times.items():
        t = statistics.median(ts)
        print(json.dumps(dict(flavour=k, M=M, us=round(t, 1), min_us=round(min(ts), 1), tflops=round(fl / t / 1e6, 1))), flush=True)


if __name__ == "__main__":
    main()
