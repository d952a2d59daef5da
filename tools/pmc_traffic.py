"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, csv output).
    python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/pmc_counter_collection.csv gpurun_out/pmc_WRITE_SIZE/pmc_counter_collection.csv
Writes profiles/pmc_traffic.json {bench kernel key: bytes per launch} and prints a table.
Corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B as reported by this
rocprofv3; on gfx950 FETCH_SIZE counts a wide coalesced read at half its bytes (TCC_EA0_RDREQ x 64 B for 128 B requests), so it
is doubled; WRITE_SIZE is taken as reported (uncalibrated on gfx950)."""
import csv
import json
import os
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
        args = [a.strip() for a in m.group(2).split(",")]
        if m.group(1) in ("gemm256_kernel", "gemm_bf16_kernel"):
            return f"{m.group(1)}<{int(args[0] == 'true')},{int(args[1] == 'true')}>"
        return f"{m.group(1)}<{','.join(args)}>"
    m = re.search(r"ivh::(\w+)", name)
    return m.group(1) if m else name[:60]


def load(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            k = short(row["Kernel_Name"])
            acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
    return acc


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    rows = []
    for k in sorted(fetch, key=lambda k: -fetch[k][0]):
        n = fetch[k][1]
        rd = 2.0 * fetch[k][0] / n * 1024.0
        wr = write[k][0] / max(write[k][1], 1) * 1024.0 if k in write else 0.0
        out[k] = dict(bytes_per_launch=round(rd + wr), read_bytes=round(rd), write_bytes=round(wr), launches=n)
        rows.append((k, n, rd / 1e6, wr / 1e6))
    os.makedirs("profiles", exist_ok=True)
    json.dump(dict(source_digest=_digest(), kernels=out), open("profiles/pmc_traffic.json", "w"), indent=1, sort_keys=True)
    print("| kernel | launches | read MB/launch (2 x FETCH_SIZE) | write MB/launch (WRITE_SIZE) |\n|---|---:|---:|---:|")
    for k, n, rd, wr in rows[:24]:
        print(f"| `{k}` | {n} | {rd:.1f} | {wr:.1f} |")


if __name__ == "__main__":
    main()
