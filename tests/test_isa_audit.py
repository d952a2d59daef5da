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
