"""ISA-level audit of the row kernels (CPU: hipcc cross-compiles gfx950 assembly without a GPU).

Round 5 found a gfx950 hazard LLVM does not cover: a 128-bit MUBUF store with an SGPR soffset followed directly by a VALU write of its data
registers stores garbage (tools/isa_store_hazard_scan.py, profiles/r5_store_data_hazard_gfx950.txt).  norms.hip holds every such store of
the library (scalar row offsets of the bytes-in-flight kernels); its assembly must contain no site of the pattern, whatever the register
allocator did this time."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_store_hazard_scan as scan  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

BAD = """
_Zkernel:
	buffer_store_dwordx4 v[82:85], v195, s[48:51], s63 offen
	; sched_barrier mask(0x00000000)
	;;#ASMSTART
	;;#ASMEND
	v_pk_mul_f32 v[82:83], s[18:19], v[2:3]
	s_endpgm
"""
GOOD = """
_Zkernel:
	buffer_store_dwordx4 v[82:85], v195, s[48:51], s63 offen
	s_nop 3
	v_pk_mul_f32 v[82:83], s[18:19], v[2:3]
	buffer_store_dwordx4 v[82:85], v195, s[48:51], 0 offen
	v_pk_mul_f32 v[82:83], s[18:19], v[2:3]
	buffer_store_dwordx4 v[10:13], v195, s[48:51], s63 offen
	v_cvt_pk_bf16_f32 v10, v88, v89
	v_cvt_pk_bf16_f32 v11, v88, v89
	s_endpgm
"""


def test_scanner_flags_the_pattern_that_corrupted_stores_and_nothing_else(tmp_path):
    bad, good = tmp_path / "bad.s", tmp_path / "good.s"
    bad.write_text(BAD); good.write_text(GOOD)
    hits = scan.scan(str(bad), 4)
    assert len(hits) == 1 and hits[0][6] == 0 and hits[0][7] == 1          # next instruction, data dword 1
    # a wait state in between; an immediate soffset (LLVM's own hazard handling applies); single dwords rewritten in program order
    assert scan.scan(str(good), 4) == []


@pytest.mark.skipif(not (os.path.isfile(HIPCC) or shutil.which("hipcc")), reason="no hipcc")
def test_row_kernels_have_no_store_data_hazard_site(tmp_path):
    src = os.path.join(ROOT, "internvideo_amd", "csrc", "norms.hip")
    out = tmp_path / "norms.s"
    from internvideo_amd.csrc import build as B
    cmd = [HIPCC if os.path.isfile(HIPCC) else "hipcc"] + B.FLAGS + ["-S", "--cuda-device-only", src, "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = out.read_text()
    n_scalar = sum(1 for ln in text.splitlines() if scan.STORE.match(ln) and scan.STORE.match(ln).group(5).rstrip(",").startswith("s"))
    assert n_scalar >= 10, "the audit lost its subject: no 128-bit stores with an SGPR soffset in norms.hip"
    hits = scan.scan(str(out), 4)
    assert hits == [], "\n".join(f"{h[1][:80]}: {h[3]}  ->  +{h[6]}: {h[5]}" for h in hits)


def test_scalar_offset_wide_stores_live_only_where_the_audit_looks():
    """every 128-bit buffer store outside norms.hip passes a literal 0 as soffset (LLVM's own hazard handling covers those); a new
    scalar-offset store in another file would have to join the audit above"""
    import glob
    import re
    csrc = os.path.join(ROOT, "internvideo_amd", "csrc")
    seen = 0
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        text = open(path).read()
        for m in re.finditer(r"__builtin_amdgcn_raw_buffer_store_b(96|128)\s*\(", text):
            depth, i, args, cur = 1, m.end(), [], ""
            while depth:
                ch = text[i]
                if ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                    if depth == 0:
                        break
                if ch == "," and depth == 1:
                    args.append(cur.strip()); cur = ""
                else:
                    cur += ch
                i += 1
            args.append(cur.strip())
            assert len(args) == 5, (path, args)
            seen += 1
            if os.path.basename(path) != "norms.hip":
                assert args[3] == "0", f"{os.path.basename(path)}: 128-bit buffer store with soffset `{args[3]}` outside the audited file"
        for m in re.finditer(r"\basm\s+volatile\s*\(", text):                       # inline-asm stores would escape both checks
            depth, i = 1, m.end()
            while depth:
                depth += {"(": 1, ")": -1}.get(text[i], 0)
                i += 1
            assert not re.search(r"buffer_store_dwordx[34]", text[m.end():i]), f"{path}: wide buffer store in inline asm"
    assert seen >= 6


@pytest.mark.skipif(not (os.path.isfile(HIPCC) or shutil.which("hipcc")), reason="no hipcc")
def test_attention_np_kernels_contain_no_packed_fp32_instruction(tmp_path):
    """round 6: the shipped 32x32 attention kernels (`attn32_*_np_kernel`, flash_attn32.hip) are compiled under target("no-packed-fp32-ops") --
    v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 of one wave never run beside another wave's MFMAs on a gfx950 SIMD and hold it twice as long per
    issue (tools/probes/mfma_valu_mix.hip, profiles/r6_mfma_valu_mix_*.jsonl).  The property lives in an attribute the compiler could stop
    honouring: the assembly of every np kernel must hold MFMAs and no packed fp32 arithmetic, and the packed twins must still hold some (else
    the A/B switch IVH_ATTN_NOPK compares a kernel with itself)."""
    import re
    from collections import Counter
    src = os.path.join(ROOT, "internvideo_amd", "csrc", "flash_attn32.hip")
    out = tmp_path / "flash_attn32.s"
    from internvideo_amd.csrc import build as B
    cmd = [HIPCC if os.path.isfile(HIPCC) else "hipcc"] + B.FLAGS + ["-S", "--cuda-device-only", src, "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    cur, cnt = None, {}
    for line in out.read_text().splitlines():
        m = re.match(r"^(_ZN3ivh\w+):", line)
        if m:
            cur = m.group(1); cnt[cur] = Counter(); continue
        if line.startswith(".Lfunc_end"):
            cur = None
        if cur:
            m = re.match(r"\s+(v_pk_(?:fma|mul|add)_f32|v_mfma_\w+)", line)
            if m:
                cnt[cur]["pk" if m.group(1).startswith("v_pk") else "mfma"] += 1
    np_k = {k: v for k, v in cnt.items() if re.search(r"attn32(pp)?_(fwd|bwd_dq|bwd_dkdv)_np_kernel", k)}
    pk_k = {k: v for k, v in cnt.items() if re.search(r"attn32_(fwd|bwd_dq|bwd_dkdv)_kernel", k)}
    assert len(np_k) >= 8 and len(pk_k) >= 8, (sorted(np_k), sorted(pk_k))      # fwd x 3 head dims, dq x 3, dkdv x 2 (+ the two-group probe)
    for k, v in np_k.items():
        assert v["mfma"] >= 24 and v["pk"] == 0, (k, dict(v))
    assert all(v["pk"] > 0 for v in pk_k.values()), {k: dict(v) for k, v in pk_k.items()}
