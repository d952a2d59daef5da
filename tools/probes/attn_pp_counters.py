"""Launches for the SQ-counter comparison of the one-group attention forward (mode 0) and the two-wave-group variant (mode 1): run under
rocprofv3 --pmc (tools/probes/attn_pp_counters.sh).  The two kernels have different names, so the per-kernel summary separates them."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from internvideo_amd import ops  # noqa: E402
from internvideo_amd.lib import call  # noqa: E402
from tools.bench_kernels import rnd  # noqa: E402

ops.set_attn_kernel(2)
B, L, H, hd = (8, 2049, 16, 88) if "--long" in sys.argv else (128, 417, 16, 88)
qkv = rnd(B * L, 3 * H * hd)
for md in (0, 1):
    call("ivh_probe_attn32_pingpong", md)
    for _ in range(6):
        ops.flash_attn_fwd_packed(qkv, B, L, H)
torch.cuda.synchronize()
