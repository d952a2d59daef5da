"""Where a forward attention workgroup's time goes (32x32x16 kernel, the 1B training shape): wave 0 of every workgroup stamps the shader
clock at entry, at the start and end of the key-tile loop and at exit (ivh_attn32_debug_stamps).  GPU box only.
    python tools/attn_timeline.py [B] -> one JSON line + a small markdown table on stdout"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import lib as L_, ops  # noqa: E402
from tools.bench_kernels import rnd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L, H, hd = 417, 16, 88
D = H * hd
qkv = rnd(B * L, 3 * D)
ops.set_attn_kernel(2)
for _ in range(3):
    ops.flash_attn_fwd_packed(qkv, B, L, H)
npass = (L + 127) // 128
nwg = npass * H * B
stamps = torch.zeros((nwg, 4), dtype=torch.int64, device="cuda")
lib = L_.load()
lib.ivh_attn32_debug_stamps(stamps.data_ptr(), nwg)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.flash_attn_fwd_packed(qkv, B, L, H)
e1.record()
torch.cuda.synchronize()
lib.ivh_attn32_debug_stamps(None, 0)
us = e0.elapsed_time(e1) * 1e3
t = stamps.cpu().double()
# The stamps are shader-clock cycles (s_memtime); the counters of the 8 XCDs are not synchronised with each other, so only differences inside
# a workgroup are used.  Cycles -> microseconds through the launch itself: the sum of all workgroup lifetimes is (average concurrency) x
# (launch time), and the launch time of an UNstamped launch of the same problem is measured with HIP events right here.
xcd = torch.arange(nwg) % 8
spans = torch.stack([t[xcd == x, 3].max() - t[xcd == x, 0].min() for x in range(8)])
span = spans.median().item()
e0.record()
for _ in range(5):
    ops.flash_attn_fwd_packed(qkv, B, L, H)
e1.record()
torch.cuda.synchronize()
us_plain = e0.elapsed_time(e1) * 1e3 / 5
mhz = 2100.0                                                      # nominal: the shares below do not depend on it
pro, loop, epi = (t[:, 1] - t[:, 0]), (t[:, 2] - t[:, 1]), (t[:, 3] - t[:, 2])
life = t[:, 3] - t[:, 0]
# the kernel maps workgroup id -> (b, h, pass) after an XCD remap; the pass with 33 of 128 queries is the short one: split by loop length
short = loop < loop.median() * 0.6
q = lambda x: [round(float(v) / mhz, 2) for v in (x.mean(), x.median(), x.quantile(0.9))]     # noqa: E731
out = dict(kernel="attn32_fwd_kernel<96>", B=B, L=L, H=H, hd=hd, launch_us_stamped=round(us, 1), launch_us=round(us_plain, 1), workgroups=nwg,
           assumed_cycles_per_us=mhz, avg_concurrent_workgroups=round(float((t[:, 3] - t[:, 0]).sum()) / (us_plain * mhz), 1),
           lifetime_us=q(life), prologue_us=q(pro), loop_us=q(loop), epilogue_us=q(epi),
           prologue_share=round(float(pro.sum() / life.sum()), 4), loop_share=round(float(loop.sum() / life.sum()), 4),
           epilogue_share=round(float(epi.sum() / life.sum()), 4), per_tile_us_full_workgroups=round(float(loop[~short].mean()) / mhz / 7, 3),
           short_workgroups=int(short.sum()), short_loop_us=q(loop[short]) if short.any() else None,
           slots=256 * 3, raw_mean_cycles=dict(lifetime=float(life.mean()), prologue=float(pro.mean()),
           loop=float(loop.mean()), epilogue=float(epi.mean())), xcd_span_ticks=[float(v) for v in spans])
print(json.dumps(out))
print("| phase | mean us | median us | p90 us | share of workgroup lifetime |\n|---|---:|---:|---:|---:|")
for name, x, sh in (("prologue (Q load, first K / V tile DMA, barrier)", pro, out["prologue_share"]), ("7 key tiles", loop, out["loop_share"]),
                    ("epilogue (normalise, store)", epi, out["epilogue_share"]), ("lifetime", life, 1.0)):
    a = q(x)
    print(f"| {name} | {a[0]} | {a[1]} | {a[2]} | {sh:.3f} |")
