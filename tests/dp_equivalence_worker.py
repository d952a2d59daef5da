"""Is the W-rank data-parallel step the single-rank step on the concatenated batch?  

    python tests/dp_equivalence_worker.py --wire fp32                         one process: the whole batch on one rank (the reference run)
    python -m torch.distributed.run --nproc-per-node 2 ... tests/dp_equivalence_worker.py --wire fp32 --share-gpu
                                                                       W processes: rank r takes clips [r B / W, (r + 1) B / W)

Runs `--steps` real engine steps (forward + fused distillation loss + backward + bucketed gradient reduction + clip + fused AdamW) of the
S/14 student (BASELINE configs[0] geometry, drop_path 0: DropPath's per-sample draws are the only batch-layout-dependent randomness) on a
fixed synthetic batch and prints ONE JSON line from rank 0: the loss of every step averaged over the ranks (the local losses are means
over equal-sized local batches, so their mean IS the loss of the concatenated batch) and the global gradient norm the engine clipped with.
tests/test_multiproc_gpu.py compares the W = 2 lines of the three reductions with the W = 1 line: step 1 (same weights) to fp32 rounding,
the later steps (weights moved by the reduced gradients) within the 1e-3 relative loss bar of north_star.

--share-gpu: every rank on cuda:0, collectives over gloo staged through host memory (bench._host_staged_collectives) -- the only way to run
W > 1 real processes on a one-GPU box; with one GPU per rank and --backend nccl the same script runs over RCCL.
Reference: single_modality/run_pretraining.py:377-379 (DDP), engines/engine_for_pretraining.py:151-158 (loss averaged over ranks)."""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--wire", default="fp32", choices=["fp32", "bf16", "zero1"])
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="GLOBAL batch (split over the ranks)")
    ap.add_argument("--config", default="S14")
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--lr", type=float, default=1.5e-4)
    a = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = 0 if a.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        if a.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
            spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
            bench = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(bench)
            bench._host_staged_collectives()
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=180))
    from internvideo_amd import internvideo2_pretrain as M
    from internvideo_amd.engine import IVTrainEngine
    from oracle import internvideo2_oracle as O          # test infrastructure: the synthetic batch / parameter generators only

    cfg = O.named_config(a.config)
    torch.manual_seed(0)
    with torch.device(dev):
        model = M.PretrainInternVideo2(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
                                       num_frames=cfg.num_frames, attn_pool_num_heads=cfg.attn_pool_num_heads, clip_embed_dim=cfg.clip_embed_dim,
                                       clip_teacher_embed_dim=cfg.clip_teacher_embed_dim, clip_teacher_final_dim=cfg.clip_teacher_final_dim,
                                       clip_return_layer=cfg.clip_return_layer, mae_teacher_embed_dim=cfg.mae_teacher_embed_dim,
                                       mae_return_layer=cfg.mae_return_layer, drop_path_rate=0.0)
    model.load_state_dict({k: v.to(dev) for k, v in O.synthetic_params(cfg, seed=3).items()}, strict=True)
    model.train()
    kw = dict(reduce_mode="zero1") if a.wire == "zero1" else dict(reduce_dtype=a.wire)
    eng = IVTrainEngine(model, lr=a.lr, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.05, max_grad_norm=3.0, bucket_bytes=4 << 20, **kw)
    T, h, w = cfg.grid
    n_vis = max(1, (h * w) // 5)
    video, mask, targets = O.synthetic_batch(cfg, a.batch, n_vis, seed=7)
    assert a.batch % world == 0
    b0, b1 = rank * a.batch // world, (rank + 1) * a.batch // world
    video = video[b0:b1].to(dev).to(torch.bfloat16)
    mask_t = torch.from_numpy(mask[b0:b1]).to(dev).to(torch.uint8)
    tg = (targets[0][:, b0:b1].to(dev).to(torch.bfloat16).contiguous(), targets[1][b0:b1].to(dev).to(torch.bfloat16).contiguous(),
          targets[2][:, b0:b1].to(dev).to(torch.bfloat16).contiguous())
    L = 1 + T * n_vis
    losses, norms = [], []
    for _ in range(a.steps):
        vis_inv = M.build_gather_indices(mask_t, dev, L=L, check=False)
        loss, _ = eng.train_step(video, mask_t, tg, vis_inv=vis_inv)
        lv = loss.detach().reshape(1).double()
        if world > 1:
            dist.all_reduce(lv)
            lv /= world
        losses.append(float(lv.item()))
        norms.append(float(eng.grad_norm.item()))
    torch.cuda.synchronize()
    if rank == 0:
        print(json.dumps({"world": world, "wire": a.wire, "config": a.config, "global_batch": a.batch, "losses": losses, "grad_norms": norms,
                          "buckets": len(eng.buckets), "reduce": f"{eng.reduce_mode}/{eng.reduce_dtype}"}), flush=True)
    if world > 1:
        dist.barrier()
        if dist.get_backend() != "nccl":
            dist.destroy_process_group()
        else:
            sys.stdout.flush()
            os._exit(0)


if __name__ == "__main__":
    main()
