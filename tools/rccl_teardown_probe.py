"""How often does `dist.destroy_process_group()` abort after an RCCL run, and does the teardown ORDER matter?  

    python tools/rccl_teardown_probe.py [--trials 6]          (one GPU: 1-rank RCCL groups, each trial in its own process)

Each trial: init a 1-rank "nccl" group, build the S/14 student under IVTrainEngine(force_comm=True), run the step in one of the multi-rank
modes (segments = chain of graphs with eager collectives between them -- the bench default; overlap = collectives captured INSIDE the graph;
eager), then tear down either in order (engine.close(): graphs, then streams; barrier; destroy_process_group(); normal interpreter exit) or
naively (destroy_process_group() with the graphs alive).  Prints one JSON line: per (mode, teardown) the return codes of the trials.
A non-zero code after "STEP_OK" was printed is a teardown abort."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
mode, teardown = sys.argv[1], sys.argv[2]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from internvideo_amd import internvideo2_pretrain as M
from internvideo_amd.engine import IVTrainEngine
torch.manual_seed(0)
with torch.device("cuda"):
    m = M.PretrainInternVideo2(img_size=112, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4.0, num_frames=4, attn_pool_num_heads=6, clip_embed_dim=384,
                               clip_teacher_embed_dim=768, clip_teacher_final_dim=384, clip_return_layer=2, mae_teacher_embed_dim=384, mae_return_layer=2)
m.train()
eng = IVTrainEngine(m, force_comm=True, bucket_bytes=8 << 20)
B, T, n_vis = 4, 4, 13
g = torch.Generator(device="cuda").manual_seed(1)
video = torch.rand((B, 3, T, 112, 112), device="cuda", generator=g).to(torch.bfloat16)
perm = torch.rand((B, T, 64), device="cuda", generator=g).argsort(-1)
mask = torch.ones((B, T, 64), dtype=torch.bool, device="cuda"); mask.scatter_(2, perm[:, :, :n_vis], False)
mask = torch.cat([torch.zeros((B, 1), dtype=torch.bool, device="cuda"), mask.reshape(B, -1)], 1).to(torch.uint8)
L = 1 + T * n_vis
unit = lambda *s: torch.nn.functional.normalize(torch.randn(s, device="cuda", generator=g), dim=-1).to(torch.bfloat16)
targets = (unit(2, B, L, 768), unit(B, 384), unit(2, B, L - 1, 384))
if mode == "segments":
    eng.capture_step(video, mask, targets, L=L, segmented=True)
elif mode == "overlap":
    eng.capture_step(video, mask, targets, L=L, capture_comm=True)
for _ in range(3):
    if mode == "eager":
        vi = M.build_gather_indices(mask, torch.device("cuda"), L=L, check=False)
        loss, _ = eng.train_step(video, mask, targets, vis_inv=vi)
    else:
        loss, _ = eng.train_step_graphed()
torch.cuda.synchronize()
assert torch.isfinite(loss).item()
print("STEP_OK", flush=True)
if teardown == "ordered":
    eng.close()
    dist.barrier()
    torch.cuda.synchronize()
dist.destroy_process_group()
print("TEARDOWN_OK", flush=True)
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=6)
    a = ap.parse_args()
    script = os.path.join("/tmp", "ivh_teardown_worker.py")
    open(script, "w").write(WORKER.format(root=ROOT))
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = {}
    for mode in ("segments", "overlap", "eager"):
        for td in ("ordered", "naive"):
            codes = []
            for _ in range(a.trials):
                r = subprocess.run([sys.executable, script, mode, td], env=env, capture_output=True, text=True, timeout=300)
                codes.append(dict(rc=r.returncode, step_ok="STEP_OK" in r.stdout, teardown_ok="TEARDOWN_OK" in r.stdout,
                                  err=(r.stderr.strip().splitlines()[-1][:160] if r.returncode and r.stderr.strip() else None)))
            res[f"{mode}/{td}"] = dict(aborts_after_step=sum(1 for c in codes if c["step_ok"] and not c["teardown_ok"]), trials=len(codes),
                                       rcs=[c["rc"] for c in codes], errs=sorted({c["err"] for c in codes if c["err"]}))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
