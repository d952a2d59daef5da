"""Per-kernel MFMA utilisation from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE; csv output).
    python tools/pmc_mfma.py gpurun_out/pmc_mfma/pmc_counter_collection.csv
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs) (gfx94x formula; MI355X_MICROARCH.md notes that ROCm 7.2 has no
gfx950 derived-counter section): the fraction of SIMD-cycles the matrix pipes were busy while the kernel ran."""
import csv
import re
import sys
from collections import defaultdict


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
    print("| kernel | launches | MFMA busy / (GUI_ACTIVE x 1024 SIMDs) |\n|---|---:|---:|")
    rows = []
    for k, c in acc.items():
        if c.get("GRBM_GUI_ACTIVE", 0) > 0 and c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
            rows.append((c["SQ_VALU_MFMA_BUSY_CYCLES"], k, n[k], c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 1024.0)))
    for _, k, nn, u in sorted(rows, reverse=True)[:14]:
        print(f"| `{k}` | {nn} | {100 * u:.1f} % |")


if __name__ == "__main__":
    main()
