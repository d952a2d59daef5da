"""What does the 1B training step draw, and what do its kernel families draw on their own?  (Round 5: a 20 us saving per launch in an
HBM-bound kernel and in a GEMM epilogue -- 1.5 ms per step by the kernels' own clocks -- left the step time unchanged on the same box, with
every GEMM launch ~1 % slower: profiles/r5_ab_dpp_colsum_qk768_v1.jsonl.  If the board's power limit acts on an average over more than a
kernel's length, the step is bound by ENERGY, and time saved in a low-power kernel comes back as lower clocks in the GEMMs.)

    python tools/step_power_probe.py [seconds per leg]          (GPU box; rocm-smi sampled from a side thread)

Legs: the graph-replayed step back to back; then, each back to back on the step's own shapes, the fc1 forward GEMM, the grouped-wgrad-shaped
GEMM, attention forward + backward, the residual RMSNorm backward, the q/k-norm backward, AdamW.  One JSON line per leg: mean socket
power, mean sclk, and the leg's own rate."""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from internvideo_amd import internvideo2_pretrain as M, ops  # noqa: E402
from internvideo_amd.engine import IVTrainEngine  # noqa: E402

DEV = "cuda"


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10)
            c = json.loads(r.stdout).get("card0", {})
            rec = {}
            for k, v in c.items():
                kl = k.lower()
                if "power" in kl and "w" in kl:
                    try:
                        rec["power_w"] = float(str(v).split()[0])
                    except ValueError:
                        pass
                if "sclk" in kl:
                    try:
                        rec["sclk_mhz"] = float(str(v).strip("()").lower().replace("mhz", "").split()[-1])
                    except ValueError:
                        pass
            out.append(rec)
        except Exception as e:  # noqa: BLE001
            out.append({"error": str(e)[:80]})
        time.sleep(0.2)


def leg(name, fn, secs, unit_per_call=None):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, samples))
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < secs:
        for _ in range(10):
            fn()
        n += 10
        torch.cuda.synchronize()
    dt = time.time() - t0
    stop.set(); th.join()
    s = [x for x in samples[2:-1] if "power_w" in x]
    pw = sum(x["power_w"] for x in s) / max(len(s), 1)
    ck = [x["sclk_mhz"] for x in s if "sclk_mhz" in x]
    print(json.dumps(dict(leg=name, calls=n, ms_per_call=round(dt / n * 1e3, 3), mean_power_w=round(pw, 1), mean_sclk_mhz=(round(sum(ck) / len(ck), 0) if ck else None),
                          samples=len(s), energy_j_per_call=round(pw * dt / n, 2), **(unit_per_call or {}))), flush=True)


def rnd(*shape):
    return ((torch.rand(*shape, device=DEV) * 2 - 1)).to(torch.bfloat16)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    B, T, n_vis, Dm, H, Hm = 128, 8, 52, 1408, 16, 6144
    L = 1 + T * n_vis
    Mrows = B * L
    torch.manual_seed(0)
    with torch.device(DEV):
        model = M.pretrain_internvideo2_1B_patch14_224(drop_path_rate=0.25, num_frames=T, clip_return_layer=6, mae_return_layer=4)
    model.residual_dtype = "bf16"
    model.train()
    eng = IVTrainEngine(model, lr=1.5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.05, max_grad_norm=3.0)
    g = torch.Generator(device=DEV).manual_seed(1)
    video = torch.rand((B, 3, T, 224, 224), device=DEV, generator=g).to(torch.bfloat16)
    perm = torch.rand((B, T, 256), device=DEV, generator=g).argsort(-1)
    mask = torch.ones((B, T, 256), dtype=torch.bool, device=DEV); mask.scatter_(2, perm[:, :, :n_vis], False)
    mask = torch.cat([torch.zeros((B, 1), dtype=torch.bool, device=DEV), mask.reshape(B, -1)], 1).to(torch.uint8)
    unit = lambda *s: torch.nn.functional.normalize(torch.randn(s, device=DEV, generator=g), dim=-1).to(torch.bfloat16)   # noqa: E731
    targets = (unit(6, B, L, 3200), unit(B, 768), unit(4, B, L - 1, 1408))
    eng.capture_step(video, mask, targets, L=L)
    leg("whole step (graph replay + AdamW)", lambda: eng.train_step_graphed(), secs, dict(clips=B))
    # --- families on the step's own shapes
    x, w1, b1 = rnd(Mrows, Dm), rnd(Hm, Dm) * 0.05, torch.rand(Hm, device=DEV) - 0.5
    leg("fc1 forward GEMM, plain bias (gemm256 NT)", lambda: ops.gemm(x, w1, bias=b1), secs, dict(gflop=2.0 * Mrows * Dm * Hm / 1e9))
    leg("fc1 forward GEMM + GELU + gelu' copy (EPI 2)", lambda: ops.gemm(x, w1, bias=b1, act="gelu_erf_d", want_preact=True), secs, dict(gflop=2.0 * Mrows * Dm * Hm / 1e9))
    dy = rnd(Mrows, Hm)
    leg("wgrad GEMM dW[6144,1408] = dy^T x (gemm256 TN, K = 53376)", lambda: ops.gemm(dy, x, a_kc=False, b_kc=False), secs, dict(gflop=2.0 * Mrows * Dm * Hm / 1e9))
    qkv = rnd(Mrows, 3 * Dm)
    att, lse = ops.flash_attn_fwd_packed(qkv, B, L, H)
    datt = rnd(Mrows, Dm)
    leg("attention forward (packed qkv, hd 88, L 417)", lambda: ops.flash_attn_fwd_packed(qkv, B, L, H), secs)
    leg("attention backward (dq + dkdv)", lambda: ops.flash_attn_bwd_packed(qkv, att, datt, lse, B, L, H), secs)
    wn = torch.ones(Dm, device=DEV)
    rq, rk = ops.qk_rmsnorm_fwd(qkv.clone(), wn, wn, 1e-6)
    dqkv = rnd(Mrows, 3 * Dm)
    leg("qk_rmsnorm_bwd (902 MB)", lambda: ops.qk_rmsnorm_bwd(qkv, dqkv, wn, wn, rq, rk), secs, dict(mbytes=Mrows * Dm * 12 / 1e6))
    res = rnd(Mrows, Dm); br = rnd(Mrows, Dm); gam = torch.ones(Dm, device=DEV)
    res2, n2, rstd = ops.rmsnorm_add_fwd(res, br, gam, None, L, wn, 1e-6)
    leg("rmsnorm_add_fwd (bf16 stream, 601 MB)", lambda: ops.rmsnorm_add_fwd(res, br, gam, None, L, wn, 1e-6), secs, dict(mbytes=Mrows * Dm * 8 / 1e6))
    dn, dres = rnd(Mrows, Dm), rnd(Mrows, Dm)
    leg("rmsnorm_add_bwd (bf16 stream, 902 MB)", lambda: ops.rmsnorm_add_bwd(dn, dres, res2, rstd, wn, br, gam, None, L, want_dbias=True), secs, dict(mbytes=Mrows * Dm * 12 / 1e6))
    leg("AdamW + clip (30 GB)", lambda: eng.optimizer_step(), secs)


if __name__ == "__main__":
    main()
