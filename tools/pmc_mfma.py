"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE; csv output), calibrated.
    python tools/pmc_mfma.py <dir>/pmc_counter_collection.csv gpurun_out/mfma_probe.json
ROCm 7.2 ships no gfx950 derived-counter section (MI355X_MICROARCH.md), so the gfx94x "MfmaUtil" formula's normalisation cannot be
trusted.  Instead the same pass contains a known-rate MFMA stream (tools/mfma_calib_run.py -> ivh_probe_mfma_rate) whose achieved
fraction of the 2.5 PFLOP/s dense bf16 peak was measured with HIP events in the profiled process:
    scale = probe_fraction_of_peak / (probe BUSY / probe GUI_ACTIVE)          MfmaUtil(kernel) = scale * BUSY / GUI_ACTIVE
i.e. MfmaUtil is "matrix-pipe busy time relative to a kernel that keeps every matrix pipe busy all the time", on the 2.5 PFLOP/s scale."""
import csv
import json
import re
import sys
from collections import defaultdict


def _digest():
    """digest of the kernel sources these counters were measured on (same as bench.py's _source_digest)"""
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        from internvideo_amd.csrc import build as b
        root = os.path.dirname(os.path.dirname(b.HERE))
        return b._digest(b.sources() + [os.path.join(b.HERE, "common.h")] + b.headers())[:16]
    except Exception:
        return None


def short(name):
    m = re.search(r"ivh::(\w+)<([^>]*)>", name)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    m = re.search(r"ivh::(\w+)", name)
    return m.group(1) if m else name[:50]


def main():
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    with open(sys.argv[1], newline="") as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                n[k] += 1
    probe = json.load(open(sys.argv[2])) if len(sys.argv) > 2 else None
    raw = {k: c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["GRBM_GUI_ACTIVE"] for k, c in acc.items()
           if c.get("GRBM_GUI_ACTIVE", 0) > 0 and c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0}
    scale, note = None, "uncalibrated (no probe rows)"
    pk = next((k for k in raw if "probe_mfma_rate" in k), None)
    if probe and pk:
        scale = probe["frac_of_2500"] / raw[pk]
        c = acc[pk]
        per_launch = c["SQ_VALU_MFMA_BUSY_CYCLES"] / n[pk]
        note = (f"probe: {probe['tflops']:.0f} TFLOP/s by HIP events = {100 * probe['frac_of_2500']:.1f} % of 2.5 PFLOP/s; raw BUSY / GUI_ACTIVE = {raw[pk]:.2f}; "
                f"BUSY per launch {per_launch:.3e} vs 1024 SIMDs x {probe['expected_busy_cycles_per_simd']} expected cycles = "
                f"{1024.0 * probe['expected_busy_cycles_per_simd']:.3e} (ratio {per_launch / (1024.0 * probe['expected_busy_cycles_per_simd']):.3f}); scale = {scale:.5f}")
    print(f"calibration: {note}\n")
    print("| kernel | launches | raw BUSY / GUI_ACTIVE | MfmaUtil (calibrated, of 2.5 PFLOP/s) |\n|---|---:|---:|---:|")
    for k in sorted(raw, key=lambda k: -acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"])[:16]:
        u = f"{100 * scale * raw[k]:.1f} %" if scale else "n/a"
        print(f"| `{k}` | {n[k]} | {raw[k]:.3f} | {u} |")
    out = {k: dict(raw=raw[k], mfma_util=(scale * raw[k] if scale else None), launches=n[k]) for k in raw}
    json.dump(dict(source_digest=_digest(), calibration=note, scale=scale, kernels=out), open("profiles/pmc_mfma_util.json", "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
