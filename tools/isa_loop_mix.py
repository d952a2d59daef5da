"""Instruction mix of the largest loop of a kernel in hipcc's -S output (authoring-container aid).
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -S --cuda-device-only -o x.s file.hip
    python tools/isa_loop_mix.py x.s <mangled-name-substring>"""
import re
import sys
from collections import Counter


def main(path, key):
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(key) + r'\S*:', l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    body = [l.strip() for l in lines[start + 1:end] if l.strip() and not l.strip().startswith(';')]
    labels = {re.match(r'(\.LBB\d+_\d+):', l).group(1): i for i, l in enumerate(body) if re.match(r'\.LBB\d+_\d+:', l)}
    best = None
    for i, l in enumerate(body):
        m = re.match(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = (labels[m.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    loop = [l for l in body[best[0]:best[1]] if not l.startswith('.')]
    c = Counter()
    for l in loop:
        op = l.split()[0]
        if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith('v_exp'): c['v_exp'] += 1
        elif op.startswith('v_'): c['valu'] += 1
        elif op.startswith('s_waitcnt'): c['s_waitcnt'] += 1
        elif op.startswith('s_'): c[op if op in ('s_barrier', 's_nop') else 'salu'] += 1
        else: c[op] += 1
    print(f"{key}: loop of {len(loop)} instructions")
    print(dict(c))
    print(Counter(l.split()[0] for l in loop if l.startswith('v_') and not l.startswith('v_mfma')).most_common(16))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
