"""Per-kernel SQ counter summary of rocprofv3 --pmc passes (csv): where the wave-cycles of a kernel go.
    python tools/pmc_sq.py <dir1>/..._counter_collection.csv [<dir2>/...csv ...] [--match attn]
Every counter is summed over the launches of a kernel and shown per launch and relative to SQ_WAVE_CYCLES of the same pass when present
(guide, MI355X_MICROARCH.md: WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"ivh::(\w+)<([^>]*)>", name)
    if m:
        return f"{m.group(1)}<{m.group(2)}>"
    m = re.search(r"ivh::(\w+)", name)
    return m.group(1) if m else name[:40]


def main():
    args = sys.argv[1:]
    match = ""
    if "--match" in args:
        i = args.index("--match")
        match = args[i + 1]
        del args[i:i + 2]
    files = [a for a in args if not a.startswith("--")]
    for path in files:
        acc = defaultdict(lambda: defaultdict(float))
        n = defaultdict(lambda: defaultdict(int))
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if match and match not in k:
                    continue
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                n[k][row["Counter_Name"]] += 1
        print(f"## {path}\n")
        for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
            c = acc[k]
            wc = c.get("SQ_WAVE_CYCLES", 0.0)
            launches = max(n[k].values())
            print(f"`{k}` ({launches} launches)")
            for name in sorted(c):
                per = c[name] / max(n[k][name], 1)
                frac = f"  ({100 * c[name] / wc:.1f} % of SQ_WAVE_CYCLES)" if wc and name != "SQ_WAVE_CYCLES" else ""
                print(f"    {name:32s} {per:14.4g} per launch{frac}")
            print()


if __name__ == "__main__":
    main()
