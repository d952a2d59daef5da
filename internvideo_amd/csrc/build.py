"""Build libinternvideo_hip.so (gfx950) in-tree with hipcc.  `python -m internvideo_amd.csrc.build [--force]`."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["host.hip", "gemm.hip", "gemm256.hip", "gemm_fp8.hip", "norms.hip", "flash_attn.hip", "flash_attn32.hip", "embed.hip", "optim.hip", "contrastive.hip", "teacher.hip", "videomae.hip", "bert.hip"]
LIB = os.path.join(HERE, "libinternvideo_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
         "-ffp-contract=fast"]


def _digest(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def sources():
    return [os.path.join(HERE, s) for s in SOURCES if os.path.isfile(os.path.join(HERE, s))]


def headers():
    """the product header (the drop-in C ABI) and the debug header (measurement hooks / hardware probes), both compiled into the library"""
    return [os.path.join(ROOT, "include", "internvideo_hip.h"), os.path.join(ROOT, "include", "internvideo_hip_debug.h")]


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = sources()
    deps = srcs + [os.path.join(HERE, "common.h")] + headers()
    stamp = os.path.join(HERE, ".build_stamp")
    dig = _digest(deps)
    if not force and os.path.isfile(LIB) and os.path.isfile(stamp) and open(stamp).read().strip() == dig:
        return LIB
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    def cc(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(cc, srcs))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
