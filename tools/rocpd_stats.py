"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel calls / total / average / share, like --stats CSV.
    python tools/rocpd_stats.py gpurun_out/prof/bench_results.db [--skip-first-frac 0.4] > profiles/<name>.md"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [x for x in cols if "name" in x][0]
    rows = c.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
    tot = sum(a[1] for a in agg.values())
    print(f"# rocprofv3 --kernel-trace summary of {db}\n")
    print(f"total kernel time {tot/1e6:.2f} ms over {len(rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{n}` | {a[0]} | {a[1]/1e6:.3f} | {a[1]/a[0]/1e3:.1f} | {a[2]/1e3:.1f} | {a[3]/1e3:.1f} | {100*a[1]/tot:.2f} |")


if __name__ == "__main__":
    main()
