"""gfx950 store-data hazard audit (round 5).

Found with `rmsnorm_add_bwd_b16_kernel<2, 1, true>`: a 128-bit MUBUF store whose soffset is an SGPR,

    buffer_store_dwordx4 v[82:85], v195, s[48:51], s63 offen
    v_pk_mul_f32 v[82:83], s[18:19], v[2:3]            <- next instruction, rewrites dwords 0 AND 1 of the store data

stored garbage in dword 1 of lanes 12-15 of every row of 16 (deterministic; an `s_nop` between the two removes it:
profiles/r5_store_data_hazard_gfx950.txt).  LLVM's hazard recognizer (GCNHazardRecognizer::createsVALUHazard) treats a MUBUF store
of more than 64 bits as hazardous only when soffset is NOT a register, so it inserts nothing here.  Single-dword writes in program order
(v82 at +0, v83 at +1 ...) never showed the problem: the store seems to read data dword k about k issue slots after it issues.

This script scans device assembly (hipcc -S --cuda-device-only) for the pattern: a buffer_store_dwordx3 / x4 with an SGPR soffset followed,
within `--window` instructions, by a VALU instruction whose destination overlaps data dword k at instruction distance d <= k
(d = 0 is the next instruction).  Such sites need a wait state (`__builtin_amdgcn_s_nop`) in the source.

    python tools/isa_store_hazard_scan.py [--window 3] file.s ...       exit code 1 when a site is found
"""
import re
import sys

STORE = re.compile(r"^\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+?)(\s|$)")
DEST = re.compile(r"^\s*(v_\w+)\s+(v\[(\d+):(\d+)\]|v(\d+))(?=[,\s]|$)")


def dest_regs(line):
    m = DEST.match(line)
    if not m:
        return None
    op = m.group(1)
    if op.startswith(("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane", "v_nop")):
        return None
    if m.group(5) is not None:
        lo = hi = int(m.group(5))
    else:
        lo, hi = int(m.group(3)), int(m.group(4))
    return op, lo, hi


def scan(path, window):
    hits = []
    func = "?"
    lines = open(path).read().splitlines()
    code = []                                      # (line number, text, function) of real instructions
    for n, ln in enumerate(lines, 1):
        t = ln.strip()
        if t.endswith(":") and not t.startswith(".") and not t.startswith(";"):
            func = t[:-1].split(":")[0]
        if re.match(r"^[A-Za-z_][\w.$]*:\s*(;.*)?$", t) and not t.startswith(".L"):
            func = t.split(":")[0]
        if not t or t.startswith((";", ".", "//")) or t.endswith(":") or re.match(r"^[\w.$]+:", t):
            continue
        code.append((n, t, func))
    for i, (n, t, f) in enumerate(code):
        m = STORE.match(t)
        if not m:
            continue
        soff = m.group(5).rstrip(",")
        if not re.match(r"^s\d+$", soff):
            continue
        lo, hi = int(m.group(2)), int(m.group(3))
        for d in range(0, window):
            if i + 1 + d >= len(code):
                break
            n2, t2, f2 = code[i + 1 + d]
            if f2 != f or t2.startswith(("s_endpgm", "s_branch", "s_cbranch", "s_setpc")):
                break
            dr = dest_regs(t2)
            if dr is None:
                continue
            op, a, b = dr
            for r in range(max(a, lo), min(b, hi) + 1):
                k = r - lo
                if d <= k - 1 or (d == 0 and k >= 1):         # dword k rewritten earlier than ~k slots after the store issued
                    hits.append((path, f, n, t, n2, t2, d, k))
                    break
    return hits


def main():
    args = sys.argv[1:]
    window = 3
    if args and args[0] == "--window":
        window = int(args[1]); args = args[2:]
    total = 0
    for p in args:
        for (path, f, n, t, n2, t2, d, k) in scan(p, window):
            total += 1
            print(f"{path}:{n}: {f[:90]}\n    {t}\n    +{d}: {t2}    (rewrites data dword {k})")
    print(f"{total} site(s)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
