"""Ordered kernel sequence of ONE training step out of a rocprofv3 --kernel-trace CSV (an EAGER step: `bench.py --no-graph`).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-b32 --no-kernel-events
    python tools/step_sequence.py /tmp/kt/.../kt_kernel_trace.csv > gpurun_out/step_sequence.md

The step is the span between the last two `adamw_kernel<true...>` (matrix-region AdamW) dispatches.  Output: the per-step totals by kernel
family, then every dispatch that is NOT one of the block-interior kernels (GEMM, attention, norms) with its position, duration and the gap
to its predecessor -- i.e. where the at::native / copy / fill launches sit -- and the run-length-compressed sequence of the last block's
forward and backward (what a block costs in launches)."""
from __future__ import annotations

import collections
import csv
import re
import sys


def short(name: str) -> str:
    m = re.search(r"ivh::(\w+)(<[^>(]*>)?", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    name = re.sub(r"at::native::", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:110]


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1], newline="")):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "adamw_kernel<true" in r[2]]
    if len(marks) < 2:
        print("fewer than two matrix-region AdamW launches in the trace"); return
    lo, hi = marks[-2] + 1, marks[-1] + 1
    # the vector-region AdamW follows the matrix one: push both boundaries past it
    while lo < len(rows) and "adamw_kernel" in rows[lo][2]:
        lo += 1
    while hi < len(rows) and "adamw_kernel" in rows[hi][2]:
        hi += 1
    step = rows[lo:hi]
    t0, t1 = step[0][0], step[-1][1]
    busy = sum(e - s for s, e, _ in step)
    print(f"# one eager step: {len(step)} dispatches, span {(t1 - t0) / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms\n")
    fam = collections.defaultdict(lambda: [0, 0])
    for s, e, n in step:
        k = short(n)
        fam[k][0] += 1; fam[k][1] += e - s
    print("| kernel | launches | ms | avg us |\n|---|---:|---:|---:|")
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {c} | {t / 1e6:.3f} | {t / c / 1e3:.1f} |")
    interior = ("gemm256_kernel", "gemm_bf16_kernel", "attn32_", "attn_", "rmsnorm_add", "qk_rmsnorm", "colsum_finish")
    print("\n## dispatches outside the block-interior families (position in the step, kernel, us, gap to the previous dispatch in us)\n")
    prev_end = t0
    for i, (s, e, n) in enumerate(step):
        k = short(n)
        if not any(x in k for x in interior):
            print(f"{i:5d}  {(s - t0) / 1e6:8.3f} ms  {k:100s} {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:6.1f}")
        prev_end = e
    gaps = sorted(((step[i][0] - step[i - 1][1]) / 1e3, i) for i in range(1, len(step)))
    print("\n## ten largest gaps (us, position):", [(round(g, 1), i) for g, i in gaps[-10:]])


if __name__ == "__main__":
    main()
