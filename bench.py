"""bench.py -- clips/s of one InternVideo2-1B stage-1 student training step on N MI355X GPUs (one process per GPU).

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher around it: starts its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Step = forward + fused distillation loss + backward + gradient all-reduce (RCCL, overlapped) + fused AdamW of
pretrain_internvideo2_1B_patch14_224 (clip_return_layer 6, mae_return_layer 4, drop_path 0.25: the recipe of
InternVideo2/single_modality/scripts/pretraining/1B_pt.sh) on synthetic random-pixel clips 8 x 224^2, 52 visible tokens per
frame (mask ratio 0.8 -> L = 417), bf16 MFMA compute, per-GPU batch 128 by default (weak scaling; --batch 32 = the reference recipe's).  Inputs and the synthetic teacher
targets are resident in HBM before the timed region; the mask -> gather-index compaction runs inside it.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline     : the dominant kernel (the bf16 MFMA GEMM family), achieved TFLOP/s from per-launch HIP events recorded on
                 the launch stream during the timed steps, against the 2.5 PFLOP/s dense bf16 MFMA peak of gfx950;
  cpu_baseline : the CPU oracle (oracle/, a port of the reference's unfused fp32 path) timed on this box's host cores on a
                 bounded sample (rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 MFMA peak
FLOP_PER_CLIP_FWD_BWD = 2.770e12   # BASELINE.md section 2 (3 x 923.3 GFLOP)
FLOP_PER_CLIP_ENCODER = 2.653e12   # the ENCODER alone (3 x (40 blocks 880.9 + patch embed 3.39 GFLOP)) -- what north_star's
                                   # ">= 40 % of the MFMA peak on the ViT-1B encoder fwd+bwd" is quoted on (heads, loss and AdamW excluded)
PORT_OVER_REFERENCE = 1.11         # profiles/r4_cpu_baseline_reference_vs_port.json: the oracle port is 1.11x the reference module's CPU speed

MODELS = {
    "1B": dict(factory="pretrain_internvideo2_1B_patch14_224", frames=8, img=224, n_vis=52,
               kw=dict(clip_return_layer=6, mae_return_layer=4), flop=FLOP_PER_CLIP_FWD_BWD),
    "B14": dict(factory=None, frames=8, img=224, n_vis=51,
                kw=dict(embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0, clip_teacher_embed_dim=1408, clip_return_layer=6,
                        mae_return_layer=4), flop=0.253e12),
    # BASELINE configs[4]: the 6B encoder, 16 x 224^2 (mask 0.8 -> 52 visible patches per frame, L = 833).  FLOPs analytic: 48 blocks x
    # (24 D^2 + 4 L D) per token + decoders / patch embed = 10.4 TFLOP forward per clip, x 3 for forward + backward.
    "6B": dict(factory="pretrain_internvideo2_6B_patch14_224", frames=16, img=224, n_vis=52,
               kw=dict(clip_return_layer=6, mae_return_layer=4), flop=31.2e12),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None,
                    help="clips per GPU; default 128 (1B, B14), 16 for the 6B model (16-frame clips, L = 833: 128 of them do not fit 288 GB without "
                         "recomputation).  The reference recipe uses 32 (scripts/pretraining/1B_pt.sh, sized for 80 GB GPUs with activation "
                         "checkpointing); 128 uses ~155 of the 288 GB of an MI355X with no recomputation and quantises better onto 256 CUs "
                         "(measured: 32 -> 228, 48 -> 264, 64 -> 255, 96 -> 278, 128 -> 280 clips/s)")
    ap.add_argument("--model", default="1B", choices=sorted(MODELS) + ["stage2-1B"],
                    help="'stage2-1B' = BASELINE configs[3] (1B vision tower at 4x224^2 + BERT-large text / fusion tower, all four losses, forward + "
                         "backward, B = 64 unless --batch is given): runs tools/bench_stage2.py --graph --batch-text --group-wgrad and prints its line")
    ap.add_argument("--drop-path", type=float, default=0.25)
    ap.add_argument("--no-droppath-skip", action="store_true",
                    help="compute every (sample, branch) pair and multiply the dropped ones by 0, as the reference does (A/B of DropPath skipping)")
    ap.add_argument("--fp8", action="store_true", help="block GEMMs (forward, dgrad, wgrad) on per-tensor-scaled e4m3 operands (BASELINE configs[4])")
    ap.add_argument("--fp8-scaling", default="current", choices=["current", "delayed"],
                    help="--fp8: 'current' = scale from the tensor's own max|x| (two passes), 'delayed' = from the amax history of the call site (one pass)")
    ap.add_argument("--checkpoint-num", type=int, default=0, help="recompute the first N blocks in backward (use_checkpoint / checkpoint_num)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-b32", action="store_true", help="skip the secondary block measured at the reference recipe's per-GPU batch (32)")
    ap.add_argument("--no-contention", action="store_true", help="skip the measured contention term of scaling_model_predictions (a side-stream copy "
                    "of the N = 8 wire bytes through 16 / 32 CUs beside the replayed step)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the `secondary` block: the B/14 student step and the stage-2 training step (the other single-GPU BASELINE configs), each "
                         "measured by a short run of this file in a child process after the headline")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-launch HIP events (roofline -> null)")
    ap.add_argument("--wgrad-stream", action="store_true", help="run the (grouped) weight-gradient GEMMs on a second stream (A/B)")
    ap.add_argument("--no-graph", action="store_true", help="issue every kernel of the timed steps from Python instead of replaying a HIP graph")
    ap.add_argument("--force-dist", action="store_true", help="single GPU: create a 1-rank RCCL group and run the multi-GPU step (eager launches, "
                    "bucketed all-reduce on the side stream) -- exercises the N > 1 code path on a 1-GPU box")
    ap.add_argument("--with-teachers", action="store_true",
                    help="time the whole reference recipe step (engines/engine_for_pretraining.py:63-148): 16-frame clips -> frozen InternVL-6B CLIP "
                         "teacher (8 frames) + VideoMAE-g teacher (16 frames, tubelet 2), random weights -> attention-guided mask -> visible "
                         "targets -> student step.  A different workload from the default (student step on resident targets): reported under its own metric name")
    ap.add_argument("--dist-mode", default="auto", choices=["auto", "eager", "graph", "graph-overlap", "graph-segments"],
                    help="N > 1 (or --force-dist): 'graph-segments' = the step replayed from a CHAIN of HIP graphs cut where a gradient bucket becomes "
                         "final, with ordinary (eager) RCCL all-reduces on the side stream between them: overlapped with backward, ~40 host calls per "
                         "step; 'eager' = per-kernel launches with the same overlap (~2200 launches per step from Python); 'graph-overlap' = one HIP "
                         "graph INCLUDING the collectives (needs an RCCL that captures them; only ever run on a 1-rank group); 'graph' = one graph, "
                         "buckets reduced after it (no overlap); 'auto' = graph-segments, checked against an eager step on every rank, falling back "
                         "to eager on all ranks if any rank fails to capture or disagrees")
    ap.add_argument("--residual", default="bf16", choices=["bf16", "fp32"],
                    help="type of the residual stream between the blocks: 'bf16' = what the reference's bf16 recipe carries (DropoutAddRMSNorm(prenorm=True), "
                         "residual_in_fp32 False: internvideo2_pretrain.py:283-286, 467), 'fp32' = the parity setting of the tests (12 instead of 8 bytes "
                         "per element through the residual kernels)")
    ap.add_argument("--fp8-weight-scales", default="tensor", choices=["tensor", "channel"],
                    help="--fp8 only: 'channel' = one scale per output feature of each weight for the forward GEMM and one per input feature for the "
                         "transposed copy of the dgrad GEMM (ivh_fp8_quantize_weight / ivh_gemm_fp8_cs); activations and gradients stay per-tensor")
    ap.add_argument("--teacher-fp8", action="store_true",
                    help="--with-teachers only: the frozen InternVL-6B CLIP teacher's block GEMMs on the e4m3 MFMA path (weights quantised once with "
                         "per-channel scales, activations per tensor).  Opt-in: the reference runs its teachers in bf16; the line says so in `teachers`")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (nccl = RCCL; gloo only with --dry-run or "
                    "--share-gpu)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of the N > 1 step on a 1-GPU box: all ranks use cuda:0 and the collectives run over gloo (RCCL refuses two ranks on "
                         "one device).  Exercises the real multi-process step (segmented graphs, bucket collectives between them, the all-ranks "
                         "checks and the timing protocol); the line says \"shared_gpu\": true and its value is NOT a throughput claim")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / timing-protocol check without a GPU: every rank runs a trivial host-side step (one all-reduce), rank 0 "
                         "prints the JSON line with \"dry_run\": true and no throughput claim.  Used by the CPU test of the N > 1 launch path")
    ap.add_argument("--hard-exit", action="store_true", help="N > 1 / --force-dist on RCCL: leave with os._exit(0) after the line is printed instead of "
                    "engine.close() + destroy_process_group() (the default for every mode but graph-overlap, see the comment at the end of main())")
    ap.add_argument("--dist-timeout", type=int, default=600, help="seconds before a stuck collective raises instead of hanging the job")
    ap.add_argument("--check-finite", action="store_true", help="the reference's per-step NaN / Inf loss guard (engine_for_pretraining.py:151-161; "
                    "one host sync per step)")
    ap.add_argument("--gemm-kernel", type=int, default=0, help="0 cost model (default), 1 force 128^2, 2 force 256^2 (A/B)")
    ap.add_argument("--attn-kernel", type=int, default=0, help="0 automatic (32x32x16-MFMA attention kernels), 1 force the 16x16x32 kernels, 2 force 32x32x16 (A/B)")
    ap.add_argument("--reduce-mode", default="allreduce", choices=["allreduce", "zero1"],
                    help="N > 1: 'allreduce' = bucketed all-reduce of the gradients (DDP role); 'zero1' = all-to-all of bf16 shards + fp32 accumulation + "
                         "sharded AdamW + all-gather of the bf16 weights (the ZeRO-1 role of scripts/pretraining/1B_pt.sh:65)")
    ap.add_argument("--reduce-dtype", default="fp32", choices=["bf16", "fp32"],
                    help="wire / accumulation type of --reduce-mode allreduce.  fp32 (default): every bucket is widened to an fp32 communication buffer and "
                         "summed exactly -- what the reference's DeepSpeed bf16 engine does, and what keeps the W-rank step within the 1e-3 loss bar of "
                         "the single-rank step on the concatenated batch (tests/test_multiproc_gpu.py); bf16 halves the bytes on the wire and rounds the "
                         "W-way sum to bf16 inside RCCL")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 16 if args.model == "6B" else 128
    return args


def _cpu_baseline_reference(spec, iters, cores):
    """the REFERENCE's own PretrainInternVideo2 (unfused path) on the host cores -- only where the reference tree is
    mounted (the authoring container; IV_REFERENCE_ROOT).  The GPU box has no such tree and times the oracle port instead."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import ref_loader as R
    from oracle import internvideo2_oracle as O
    cfg = O.named_config("1B" if spec["factory"] else "B14")
    torch.manual_seed(0)
    m = R.build_reference_student(cfg).train()
    video, mask, targets = O.synthetic_batch(cfg, 1, spec["n_vis"], seed=0)
    mask_t = torch.from_numpy(mask)
    times = []
    for it in range(iters + 1):
        t0 = time.perf_counter()
        oc, of, om = m(video, mask_t)
        loss = ((2 - 2 * (oc * targets[0]).sum(-1)).mean() + (2 - 2 * (of * targets[1]).sum(-1)).mean() + (2 - 2 * (om * targets[2]).sum(-1)).mean())
        loss.backward()
        m.zero_grad(set_to_none=True)
        times.append(time.perf_counter() - t0)
    t = float(np.mean(times[1:]))
    return dict(value=round(1.0 / t, 4), unit="clips/s", cores=cores, kind="reference",
                sample=f"the reference's own PretrainInternVideo2 (fp32, unfused path, imported from IV_REFERENCE_ROOT) fwd+bwd, 1 clip 8x224^2 L=417, "
                       f"{iters} timed iterations after 1 warm-up, {t:.2f} s/clip")


def _cpu_baseline_6b(spec, cores):
    """BASELINE configs[4] on the host cores, bounded: the oracle at the 6B width (3200, 25 heads of 128, mlp_ratio 4, 16 x 224^2, L = 833) is timed
    at depth 1 and depth 2, forward + backward on one clip; the 48 blocks are identical, so step(48) = t(1) + 47 (t(2) - t(1)).  (The whole
    5.9 G-parameter model is 24 GB of fp32 weights and minutes of host time per pass: outside the sample budget of a bench run.)"""
    from oracle import internvideo2_oracle as O
    ts = {}
    for depth in (1, 2):
        cfg = O.StudentConfig(embed_dim=3200, depth=depth, num_heads=25, mlp_ratio=4.0, num_frames=16, attn_pool_num_heads=16, clip_embed_dim=768,
                              clip_teacher_embed_dim=3200, clip_teacher_final_dim=768, clip_return_layer=1, mae_teacher_embed_dim=1408, mae_return_layer=1)
        params = {k: v.requires_grad_(True) for k, v in O.synthetic_params(cfg, seed=0).items()}
        batch = O.synthetic_batch(cfg, 1, spec["n_vis"], seed=0)
        best = None
        for it in range(3):
            t0 = time.perf_counter()
            out = O.student_forward(params, batch[0], batch[1], cfg)
            loss, _ = O.distill_losses(out, batch[2])
            loss.backward()
            for p_ in params.values():
                p_.grad = None
            dt = time.perf_counter() - t0
            if it > 0:
                best = dt if best is None else min(best, dt)
        ts[depth] = best
    per_block = max(ts[2] - ts[1], 1e-6)
    total = ts[1] + 47 * per_block
    return dict(value=round(1.0 / total, 5), unit="clips/s", cores=cores, kind="port", extrapolated=True, scope="depth 1 and 2 timed, 48 blocks extrapolated",
                sample=f"CPU oracle (fp32, unfused reference path) at the 6B width, 1 clip 16x224^2 L=833, fwd+bwd timed at depth 1 ({ts[1]:.2f} s) and depth 2 "
                       f"({ts[2]:.2f} s): 48 identical blocks -> {total:.1f} s/clip extrapolated (one block {per_block:.2f} s)")


def cpu_baseline(spec, iters):
    """CPU baseline on this box's host cores: the reference module itself when its tree is present, else the oracle (a port of the
    reference's unfused fp32 path), fwd+bwd on 1 clip."""
    from oracle import internvideo2_oracle as O
    from internvideo_amd.hostinfo import usable_cores
    cores = usable_cores()                # affinity / cgroup-quota aware (os.cpu_count() over-reports in containers)
    torch.set_num_threads(cores)
    if spec.get("frames") == 16 and spec.get("factory", "").endswith("6B_patch14_224"):
        return _cpu_baseline_6b(spec, cores)
    ref_root = os.environ.get("IV_REFERENCE_ROOT", "/root/reference")
    if os.path.isfile(os.path.join(ref_root, "InternVideo2", "single_modality", "models", "internvideo2_pretrain.py")):
        try:
            return _cpu_baseline_reference(spec, iters, cores)
        except Exception as e:            # fall back to the port, say why
            note = f" (reference import failed: {e!r})"
    else:
        note = ""
    cfg = O.named_config("1B" if spec["factory"] else "B14")

    def make(dtype):
        g = torch.Generator().manual_seed(0)
        params = {}
        for k, shp in O.param_shapes(cfg).items():
            t = torch.randn(shp, generator=g) * 0.02
            if k.endswith("weight") and "norm" in k.split(".")[-2] or k.endswith("gamma"):
                t = torch.ones(shp)
            params[k] = t.to(dtype).requires_grad_(True)
        return params

    def run(params, batch, backward):
        video, mask, targets = batch
        t0 = time.perf_counter()
        if backward:
            out = O.student_forward(params, video, mask, cfg)
            loss, _ = O.distill_losses(out, targets)
            loss.backward()
            for p_ in params.values():
                p_.grad = None
        else:
            with torch.no_grad():
                O.student_forward(params, video, mask, cfg)
        return time.perf_counter() - t0

    def flavour(dtype, B, backward, n_timed):
        """-> (clips/s, seconds per pass, timed passes); a flavour whose warm-up pass alone takes > 12 s is reported from that pass"""
        params = make(dtype)
        batch = O.synthetic_batch(cfg, B, spec["n_vis"], seed=0, dtype=dtype)
        w = run(params, batch, backward)
        if w > 12.0:
            return round(B / w, 4), round(w, 2), 0
        ts = [run(params, batch, backward) for _ in range(n_timed)]
        t_ = float(np.mean(ts))
        return round(B / t_, 4), round(t_, 2), n_timed

    v, t, n = flavour(torch.float32, 1, True, iters)
    out = dict(value=v, unit="clips/s", cores=cores, kind="port",
               sample=f"CPU oracle (fp32, unfused reference path) fwd+bwd, 1 clip {spec['frames']}x224^2 L={1 + spec['frames'] * spec['n_vis']}, "
                      f"{n} timed iterations after 1 warm-up, {t:.2f} s/clip" + note)
    # the other flavours of the same module on the same cores (bounded: 1 timed pass each after a warm-up pass)
    fl = {}
    for name, dt, B_, bw in (("fp32_fwd_b1", torch.float32, 1, False), ("fp32_fwd_bwd_b2", torch.float32, 2, True),
                             ("bf16_fwd_b1", torch.bfloat16, 1, False), ("bf16_fwd_bwd_b1", torch.bfloat16, 1, True)):
        try:
            v_, t_, n_ = flavour(dt, B_, bw, 1)
            fl[name] = dict(clips_per_s=v_, s_per_pass=t_, timed_passes=n_)
        except Exception as e:       # noqa: BLE001
            fl[name] = {"error": repr(e)[:200]}
    out["flavours"] = fl
    # the port is FASTER than the module it restates (profiles/r4_cpu_baseline_reference_vs_port.json: 0.267 vs 0.240 clips/s on the same 8 cores,
    # alternating legs): what the reference itself would do on these cores is value / 1.11
    out["extrapolated"] = False
    out["port_over_reference"] = PORT_OVER_REFERENCE
    out["reference_equivalent_value"] = round(v / PORT_OVER_REFERENCE, 4)
    out["note"] = ("kind 'port': the GPU box has no reference tree; the oracle is pinned to the reference's own CPU outputs (tests/golden/*.npz, "
                   "tests/test_oracle_golden.py).  Where the tree is mounted (IV_REFERENCE_ROOT) the reference module itself is timed: kind 'reference'")
    return out


def scaling_model(engine, world, step_s, link_gbs=153.0, links=7, reduce_mode=None, reduce_dtype=None):
    """What the step's gradient exchange costs on paper, so that a measured 1 -> 8 GPU curve can be read against a prediction (UNMEASURED
    until the driver runs N > 1: this builder has one GPU).  xGMI is a point-to-point mesh, `links` x ~`link_gbs` GB/s per direction and GPU
    (MI355X_MICROARCH / task brief): a bandwidth-optimal all-reduce moves 2 (W-1)/W of the buffer out of every GPU; with every peer link busy
    (RCCL's multi-ring / direct algorithms on a full mesh) the wire rate is (W-1) links, with ONE ring it is one link -- both are given.
    Only what is left after backward ends is exposed: the last matrix bucket (block 0 + patch embed) and the fp32 vector region.
    Bytes on the wire follow the reduction: all-reduce of bf16 buckets 2 B / element, of fp32 communication buffers (the N > 1 default: exact
    accumulation, what DeepSpeed's bf16 engine does) 4 B; zero1 = an all-to-all of bf16 shards ((W-1)/W of the buffer out, once) plus, after
    AdamW, an all-gather of the bf16 weights (the same amount again, exposed in front of the next forward except for the vector AdamW)."""
    W = max(int(world), 1)
    mode = reduce_mode or engine.reduce_mode
    dtype = reduce_dtype or engine.reduce_dtype
    eb = 4 if (mode == "allreduce" and dtype == "fp32") else 2
    mat_b = [eb * (hi - lo) for lo, hi in engine.buckets]
    vec_b = 4 * engine.n_vec
    f = (2.0 if mode == "allreduce" else 1.0) * (W - 1) / W if W > 1 else 0.0
    fv = 2.0 * (W - 1) / W if W > 1 else 0.0                  # the vector region is always an fp32 all-reduce

    def t(nbytes, nlinks, fac):
        return fac * nbytes / (nlinks * link_gbs * 1e9) if W > 1 else 0.0
    mesh, ring = min(W - 1, links) if W > 1 else 1, 1
    total = sum(mat_b) + vec_b
    gather_b = 2 * sum(hi - lo for lo, hi in engine.buckets) if mode == "zero1" else 0     # bf16 weights back to every rank

    def comm(nl):
        return t(sum(mat_b), nl, f) + t(vec_b, nl, fv)

    def tail(nl):
        return t(mat_b[-1] if mat_b else 0, nl, f) + t(vec_b, nl, fv) + t(gather_b, nl, (W - 1) / W if W > 1 else 0.0)
    return {"world": W, "reduce": f"{mode}/{dtype}", "grad_bytes_per_step": total, "matrix_buckets": len(mat_b),
            "bucket_bytes_min_max": [min(mat_b), max(mat_b)] if mat_b else None, "vector_region_bytes": vec_b,
            "wire_bytes_out_per_gpu": int(f * sum(mat_b) + fv * vec_b + ((W - 1) / W * gather_b if W > 1 else 0)), "link_GBps": link_gbs, "links_per_gpu": links,
            "comm_ms_all_links": round(comm(mesh) * 1e3, 2), "comm_ms_one_ring": round(comm(ring) * 1e3, 2),
            "exposed_tail_ms_all_links": round(tail(mesh) * 1e3, 2), "exposed_tail_ms_one_ring": round(tail(ring) * 1e3, 2),
            "step_ms_measured_here": round(step_s * 1e3, 2),
            "predicted_scaling_efficiency_all_links": round(step_s / (step_s + tail(mesh)), 4) if W > 1 else 1.0,
            "predicted_scaling_efficiency_one_ring": (round(step_s / max(step_s + tail(ring), comm(ring)), 4) if W > 1 else 1.0),
            "predicted_speedup_all_links": round(W * step_s / (step_s + tail(mesh)), 3) if W > 1 else 1.0,
            "status": "model only -- no N > 1 run has been measured by the builder"}


def droppath_straggler(model, B, L, step_ms, dp):
    """DropPath skipping makes a rank's step length follow its own draw: the all-reduce waits for the rank that kept the most.  Each (block, branch)
    keeps Binomial(B, keep) samples, so the executed block work of a step has a relative standard deviation sqrt(sum_i w_i^2 B p_i (1 - p_i)) /
    (B sum_i w_i keep_i) (w = the branch's FLOPs); the expected excess of the slowest of N ranks over the mean is that sigma times the expected
    maximum of N standard normals (0.56, 1.03, 1.42 for N = 2, 4, 8).  -> predicted extra ms per step and the efficiency factor, per N."""
    if not dp or not dp.get("enabled"):
        return None
    rates = np.asarray(model.drop_path_rates, dtype=np.float64)
    br = np.asarray(dp["branch_flop_per_clip_attn_mlp"], dtype=np.float64)
    w = np.repeat(br[None, :], len(rates), 0)
    keep = np.repeat((1.0 - rates)[:, None], 2, 1)
    var = (w ** 2 * B * keep * (1.0 - keep)).sum()
    mean = (w * B * keep).sum()
    block_share = dp["executed_flop_per_clip"] and (mean / B) / dp["executed_flop_per_clip"]
    sigma_ms = float(np.sqrt(var) / mean * block_share * step_ms)
    emax = {2: 0.5642, 4: 1.0294, 8: 1.4236}
    return {"sigma_ms_per_rank": round(sigma_ms, 3), "expected_wait_for_slowest_rank_ms": {str(n): round(sigma_ms * e, 3) for n, e in emax.items()},
            "efficiency_factor": {str(n): round(step_ms / (step_ms + sigma_ms * e), 4) for n, e in emax.items()},
            "how": "binomial kept counts per (block, branch) -> sigma of a rank's executed block work -> expected maximum over N ranks (model, not measured)"}


def droppath_account(model, steps, B, L, nominal_flop_per_clip):
    """executed FLOPs per clip of the steps since model.dp_count_acc was zeroed.  DropPath skipping does not compute the (block, branch, sample)
    triples the draw drops; a branch's forward FLOPs per clip are 2 L D (3D + D) + 4 L^2 D (attention) / 4 L D F (MLP), x 3 with the backward
    (the same arithmetic as the nominal 2.770 TFLOP).  Without skipping (or with drop_path 0) executed = nominal."""
    acc = model.dp_count_acc.cpu().numpy().astype(np.float64)          # [depth, 2, (samples, rows)] summed over the steps
    D = model.blocks[0].attn.qkv.weight.shape[1]
    F = model.blocks[0].mlp.fc1.weight.shape[0]
    br = np.array([3.0 * (2.0 * L * D * 4 * D + 4.0 * L * L * D), 3.0 * (4.0 * L * D * F)])     # fwd + bwd FLOPs per clip of one block's branches
    nominal_blocks = model.depth * br.sum()
    if acc.sum() == 0:                                                  # the skipping path never ran: everything was computed
        return dict(enabled=False, kept_fraction=1.0, executed_flop_per_clip=nominal_flop_per_clip, nominal_flop_per_clip=nominal_flop_per_clip,
                    skipped_flop_share=0.0, branch_flop_per_clip_attn_mlp=[float(br[0]), float(br[1])])
    kept = acc[:, :, 0] / (steps * B)                                   # [depth, 2]
    executed_blocks = float((kept * br[None, :]).sum())
    executed = nominal_flop_per_clip - nominal_blocks + executed_blocks
    return dict(enabled=True, kept_fraction=round(float(kept.mean()), 4), kept_fraction_last_block=[round(float(kept[-1, 0]), 4), round(float(kept[-1, 1]), 4)],
                executed_flop_per_clip=round(executed, 1), nominal_flop_per_clip=nominal_flop_per_clip,
                skipped_flop_share=round(1.0 - executed / nominal_flop_per_clip, 4), branch_flop_per_clip_attn_mlp=[float(br[0]), float(br[1])],
                how="kept (block, branch, sample) triples summed on the device inside every timed step (ops.droppath_plan counts)")


def secondary_lines(timeout_s: float = 150.0):
    """the other single-GPU workloads of BASELINE.json under the same clock: the B/14 distillation-size student step
    (`--model B14`, per-GPU batch 256) and the stage-2 TRAINING step (`--model stage2-1B`: 1B vision tower + BERT-large, all four losses, clip +
    AdamW).  Each is a short run of this file in a child process (its own HIP graph, its own memory) started after the headline was measured;
    the child's whole line is kept under profiles/ by the measurement scripts, here only the figures."""
    import subprocess
    out = {}
    runs = {"b14": ["--model", "B14", "--batch", "256", "--steps", "10", "--warmup", "3"],
            "stage2": ["--model", "stage2-1B", "--steps", "6", "--warmup", "2"]}
    for tag, extra in runs.items():
        t0 = time.perf_counter()
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra + ["--no-cpu-baseline", "--no-b32", "--no-secondary", "--no-kernel-events"],
                               capture_output=True, text=True, timeout=timeout_s)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                out[tag] = {"error": (r.stderr or r.stdout)[-300:], "returncode": r.returncode}
                continue
            d = json.loads(line[-1])
            out[tag] = {k: d.get(k) for k in ("metric", "value", "unit", "ms_per_step", "mfma_frac_of_step", "mfma_frac_nominal_equivalent", "launch_mode")}
            out[tag]["per_gpu_batch"] = (d.get("config") or {}).get("per_gpu_batch", d.get("batch"))
            if d.get("droppath_skip"):
                out[tag]["droppath_kept_fraction"] = d["droppath_skip"].get("kept_fraction")
            out[tag]["wall_s"] = round(time.perf_counter() - t0, 1)
        except Exception as e:       # noqa: BLE001   (never lose the headline to a secondary run)
            out[tag] = {"error": repr(e)[:300]}
    return out


def comm_contention(engine, step, step_ms, dev, steps=4):
    """What the gradient exchange of an 8-GPU job would cost THIS GPU's step beyond the wire time -- the part of the scaling
    prediction one GPU can measure.  A collective's kernels take CUs, HBM bandwidth and watts from a step that already runs at the board's
    power limit; the ideal-links model ignores that.  Stand-in: while the (graph-replayed) step runs, a side stream of high priority moves the
    wire bytes of N = 8 through k workgroups (ivh_probe_cu_hog: a copy src -> dst, i.e. read + write of the bytes an all-reduce sends and
    receives) -- fp32 wire (the default: 2 (W-1)/W x 4 B per gradient element) and bf16 wire (half), k = 16 and 32 CUs (RCCL's channel counts
    on a full mesh).  The slow-down of the step under each is reported next to the ideal-links prediction; no multi-GPU run is involved."""
    from internvideo_amd import lib as L_
    n_mat, n_vec = int(engine.n_mat), int(engine.n_vec)
    W = 8
    out = {"world_modelled": W, "how": "step replayed with ivh_probe_cu_hog on a high-priority side stream moving the N = 8 wire bytes through k workgroups; "
                                      "median of %d steps per cell against %d plain steps measured the same way" % (steps, steps)}
    side = torch.cuda.Stream(device=dev, priority=-1)

    def timed(fn):
        ts = []
        for _ in range(steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[len(ts) // 2]
    base = timed(step)
    out["plain_ms"] = round(base, 2)
    cells = {}
    for wire, eb in (("fp32", 4), ("bf16", 2)):
        nbytes = int(2.0 * (W - 1) / W * (eb * n_mat + 4 * n_vec)) // 16 * 16
        src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        for k in (16, 32):
            def both():
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    L_.call("ivh_probe_cu_hog", src.data_ptr(), dst.data_ptr(), nbytes, k, side.cuda_stream)
                step()
                torch.cuda.current_stream().wait_stream(side)
            ms = timed(both)
            with torch.cuda.stream(side):                  # the stand-in alone: how long k workgroups need for the bytes
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side)
                L_.call("ivh_probe_cu_hog", src.data_ptr(), dst.data_ptr(), nbytes, k, side.cuda_stream)
                e1.record(side)
            torch.cuda.synchronize()
            cells[f"{wire}_wire_{k}_cus"] = {"wire_bytes": nbytes, "step_ms": round(ms, 2), "slowdown": round(ms / base - 1.0, 4),
                                            "stand_in_alone_ms": round(e0.elapsed_time(e1), 2)}
        del src, dst
    out["cells"] = cells
    out["efficiency_with_contention"] = {k: round(1.0 / (1.0 + max(v["slowdown"], 0.0)), 4) for k, v in cells.items()}
    return out


def _dyn_scale(dyn):
    """executed / nominal work of a profiled launch that followed a device-side count (ops.GEMM_PROFILE / KERNEL_PROFILE entries)"""
    if dyn is None:
        return 1.0
    if isinstance(dyn, list):                                           # grouped launch: [(nominal FLOPs, (count, nominal K) | None)]
        tot = sum(f for f, _ in dyn)
        return sum(f * (float(r[0].item()) / r[1] if r is not None else 1.0) for f, r in dyn) / max(tot, 1.0)
    return float(dyn[0].item()) / float(dyn[1])


def _source_digest():
    """digest of the kernel sources + build flags (internvideo_amd/csrc/build.py): stamps PMC summaries under profiles/ to the code they measured"""
    try:
        from internvideo_amd.csrc import build as b
        deps = b.sources() + [os.path.join(b.HERE, "common.h")] + b.headers()
        return b._digest(deps)[:16]
    except Exception:
        return None


def _stamped(path, key):
    """entry `key` of a PMC summary under profiles/ (written by tools/pmc_*.py from separate rocprofv3 --pmc passes) + whether its
    source digest is the current one"""
    if not os.path.isfile(path):
        return None
    try:
        d = json.load(open(path))
    except Exception:
        return None
    table = d.get("kernels", d)
    ent = table.get(key)
    if ent is None:                                       # template argument lists grow: match "name<first arguments" as a prefix
        stem = key[:-1] if key.endswith(">") else key
        hits = [k for k in table if k.startswith(stem + ",") or k.startswith(stem + ">")]
        # several instantiations share the leading arguments (tail split, device-side row count ...): the one with the most launches speaks
        ent = table.get(max(hits, key=lambda k: (table[k].get("launches", 0), -len(k)))) if hits else None
    if ent is None:
        return None
    ent = dict(ent)
    ent["source_digest"] = d.get("source_digest")
    ent["matches_current_sources"] = (d.get("source_digest") is not None and d.get("source_digest") == _source_digest())
    return ent


def _reserve_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version banner through C stdio when a
    communicator is created; it is flushed at exit, i.e. AFTER our line).  Keep a private duplicate of the real stdout for the JSON line and
    point file descriptor 1 -- Python's sys.stdout and every C library's stdout -- at stderr for the rest of the process."""
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    return keep


def _self_launch(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it (no WORLD_SIZE in the environment): start N ranks of this file, one
    per GPU, the way the reference's scripts do (InternVideo2/multi_modality/torchrun.sh:13; single_modality/utils.py:332-373 reads
    RANK / WORLD_SIZE / LOCAL_RANK from the environment exactly as main() below does).  Rank 0's single JSON line passes through."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL's peer mappings need it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without a launcher: starting {args.gpus} ranks on 127.0.0.1:{port}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def _host_staged_collectives():
    """--share-gpu only: gloo's device path faults on this build (memory access fault in the first bucket's all-reduce), so device tensors
    are staged through host memory for the rehearsal's collectives.  Synchronous on the calling stream, which keeps the engine's ordering."""
    real_ar, real_ag, real_agt = dist.all_reduce, dist.all_gather, dist.all_gather_into_tensor

    def _h(t):
        return t.detach().to("cpu", torch.float32 if t.dtype == torch.bfloat16 else t.dtype)

    def all_reduce(t, op=dist.ReduceOp.SUM, group=None, async_op=False):
        if not t.is_cuda:
            return real_ar(t, op=op, group=group, async_op=async_op)
        h = _h(t)
        real_ar(h, op=op, group=group)
        t.copy_(h)

    def all_gather(parts, t, group=None, async_op=False):
        if not t.is_cuda:
            return real_ag(parts, t, group=group, async_op=async_op)
        hp = [_h(p) for p in parts]
        real_ag(hp, _h(t), group=group)
        for p, h in zip(parts, hp):
            p.copy_(h)

    def all_gather_into_tensor(out, t, group=None, async_op=False):
        if not t.is_cuda:
            return real_agt(out, t, group=group, async_op=async_op)
        ho = _h(out)
        real_agt(ho, _h(t), group=group)
        out.copy_(ho)

    dist.all_reduce, dist.all_gather, dist.all_gather_into_tensor = all_reduce, all_gather, all_gather_into_tensor


def _dry_run(args, json_fd):
    """the N-rank protocol of the contract (rendezvous, barrier + max-over-ranks timing, ONE line from rank 0) on host tensors"""
    import datetime
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend if args.backend == "gloo" or torch.cuda.is_available() else "gloo", rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=args.dist_timeout))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    buf = torch.ones(1 << 16)

    def step():
        if world > 1:
            dist.all_reduce(buf)
            buf.mul_(1.0 / world)

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.all(buf == 1.0)
    if rank == 0:
        out = {"metric": "dry run of the N-rank launch / timing protocol (no throughput claim)", "value": 0.0, "unit": "clips/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "n/a", "data": "synthetic", "dry_run": True,
               "config": {"workload": "dry run: one host all-reduce per step", "parallelism": f"dp{world}"},
               "rccl_ranks": (dist.get_world_size() if world > 1 else 1), "backend": (dist.get_backend() if world > 1 else "none")}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args))
    if args.dry_run:
        return _dry_run(args, _reserve_stdout())
    if args.model == "stage2-1B":                         # a different workload with its own metric name: the stage-2 bench tool's line
        sys.path.insert(0, ROOT)
        from tools import bench_stage2
        sys.argv = ["bench_stage2.py", "--batch", str(64 if args.batch == 128 else args.batch), "--steps", str(args.steps), "--warmup", str(args.warmup),
                    "--batch-text", "--group-wgrad", "--engine", "--gpus", str(args.gpus), "--reduce-mode", args.reduce_mode] + ([] if args.no_graph else ["--graph"]) + \
                   (["--no-cpu-baseline"] if args.no_cpu_baseline else []) + (["--share-gpu"] if args.share_gpu else [])
        return bench_stage2.main()
    json_fd = _reserve_stdout()
    spec = MODELS[args.model]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.share_gpu:
        local = 0
        args.backend = "gloo"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.share_gpu and world > 1:
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.dist_timeout))
        _host_staged_collectives()
    elif world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import datetime
        kw = {}
        try:                                                     # RCCL's kernels on a high-priority stream: a bucket's all-reduce starts as soon as a
            opts = dist.ProcessGroupNCCL.Options()               # CU frees up instead of queueing behind the next persistent GEMM launch
            opts.is_high_priority_stream = True
            kw["pg_options"] = opts
        except Exception:       # noqa: BLE001
            pass
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=args.dist_timeout), **kw)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (start the ranks with torch.distributed.run, or run without a launcher)"

    from internvideo_amd import internvideo2_pretrain as M, ops
    from internvideo_amd.engine import IVTrainEngine

    torch.manual_seed(0)
    with torch.device(dev):
        if spec["factory"]:
            model = getattr(M, spec["factory"])(drop_path_rate=args.drop_path, num_frames=spec["frames"], use_checkpoint=args.checkpoint_num > 0,
                                                checkpoint_num=args.checkpoint_num, **spec["kw"])
        else:
            model = M.PretrainInternVideo2(drop_path_rate=args.drop_path, num_frames=spec["frames"], use_checkpoint=args.checkpoint_num > 0,
                                           checkpoint_num=args.checkpoint_num, **spec["kw"])
    model.fp8_gemm = bool(args.fp8)
    model.fp8_scaling = args.fp8_scaling
    model.fp8_weight_scales = args.fp8_weight_scales
    model.residual_dtype = args.residual
    model.drop_path_skip = False if args.no_droppath_skip else "auto"
    model.train()
    # DropPath skipping (functional.BlockStackFn): how many (block, branch, sample) triples each step really computed -- summed on the device by
    # one 160-element add inside the (captured) step, read after the timed region: the MFMA fractions below are priced on EXECUTED FLOPs
    model.dp_count_acc = torch.zeros((model.depth, 2, 2), dtype=torch.int64, device=dev)
    n_params = sum(p.numel() for p in model.parameters())
    engine = IVTrainEngine(model, lr=1.5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.05, max_grad_norm=3.0,
                           wgrad_stream=args.wgrad_stream, force_comm=args.force_dist, reduce_mode=args.reduce_mode, reduce_dtype=args.reduce_dtype,
                           check_finite=args.check_finite)
    ops.set_gemm_kernel(args.gemm_kernel)
    ops.set_attn_kernel(args.attn_kernel)

    B, T, n_vis = args.batch, spec["frames"], spec["n_vis"]
    gh = spec["img"] // 14
    N = T * gh * gh
    L = 1 + T * n_vis
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)           # per-rank seed (run_pretraining.py:269)
    video = torch.rand((B, 3, T, spec["img"], spec["img"]), device=dev, generator=gen).to(torch.bfloat16)   # engine:127 videos.bfloat16()
    perm = torch.rand((B, T, gh * gh), device=dev, generator=gen).argsort(-1)
    mask = torch.ones((B, T, gh * gh), dtype=torch.bool, device=dev)
    mask.scatter_(2, perm[:, :, :n_vis], False)
    mask = torch.cat([torch.zeros((B, 1), dtype=torch.bool, device=dev), mask.reshape(B, -1)], 1).to(torch.uint8)

    def unit(*shape):
        t = torch.randn(shape, device=dev, generator=gen)
        return (t / t.norm(dim=-1, keepdim=True)).to(torch.bfloat16)

    cfgm = model
    targets = (unit(len(cfgm.clip_decoder), B, L, cfgm.clip_decoder[0].head.out_features),
               unit(B, cfgm.final_clip_decoder.head.out_features),
               unit(len(cfgm.mae_decoder), B, L - 1, cfgm.mae_decoder[0].head[2].out_features))

    def eager_step():
        vis_inv = M.build_gather_indices(mask, dev, L=L, check=False)       # HIP compaction kernel, no host sync
        return engine.train_step(video, mask, targets, vis_inv=vis_inv)

    distiller = None
    if args.with_teachers:
        from functools import partial
        from internvideo_amd.internvl_clip_vision import InternVL_CLIP
        from internvideo_amd.stage1 import Stage1Distiller
        from internvideo_amd.videomae_teacher import VisionTransformer
        assert args.model == "1B", "--with-teachers is the 1B recipe (scripts/pretraining/1B_pt.sh)"
        with torch.device(dev):
            clip_t = InternVL_CLIP(img_size=224, layerscale_no_force_fp32=False, clip_return_layer=6, return_attn=True).bfloat16().eval()
            mae_t = VisionTransformer(patch_size=14, embed_dim=1408, depth=40, num_heads=16, mlp_ratio=48 / 11, qkv_bias=True,
                                      norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), all_frames=16, tubelet_size=2,
                                      mae_return_layer=4).bfloat16().eval()
        for t_ in (clip_t, mae_t):                               # frozen teachers with unit-scale random weights (no checkpoint offline)
            for n_, p_ in t_.named_parameters():
                if p_.dim() >= 2:
                    torch.nn.init.normal_(p_, std=0.02)
        clip_t.fp8_gemm = bool(args.teacher_fp8)
        distiller = Stage1Distiller(engine, clip_t, mae_t, mask_type="attention", mask_ratio=0.8, td_ratio=2)
        video16 = torch.rand((B, 3, 2 * T, spec["img"], spec["img"]), device=dev, generator=gen).to(torch.bfloat16)

        def eager_step():                                        # noqa: F811  (the recipe step replaces the student-only step)
            return distiller.step(video16)

    # N = 1: the step (mask -> indices, forward, loss, backward on both streams) is captured once into a HIP graph and replayed;
    # AdamW runs after each replay.  N > 1: eager launches with the RCCL bucket overlap (collectives are not captured).
    graphed = (world == 1) and not args.no_graph and not args.force_dist and not args.with_teachers
    if graphed:
        engine.capture_step(video, mask, targets, L=L)
    step = engine.train_step_graphed if graphed else eager_step

    dist_mode, dist_note = "n/a", None
    if (world > 1 or args.force_dist) and not args.with_teachers:
        dist_mode = args.dist_mode

        def all_ranks_ok(flag: bool) -> bool:
            t_ = torch.tensor([1.0 if flag else 0.0], device=dev)
            dist.all_reduce(t_, op=dist.ReduceOp.MIN)
            return t_.item() > 0

        def back_to_eager():
            engine._segments, engine._graph, engine._defer_reduce = None, None, False
            model.grad_ready_hook = engine._on_block_done if engine.overlap else None

        if dist_mode == "auto":
            # Preferred: the step as a chain of HIP graphs cut where a gradient bucket becomes final, the bucketed collectives issued
            # eagerly between them on the side stream -- the overlap of eager mode at ~40 host calls per step instead of ~2200 launches
            # (8 Python ranks on one host cannot all enqueue 290 ms of launches per 400 ms step), and nothing asked of RCCL beyond plain
            # all-reduce calls.  Checked per run: every rank must capture AND produce a finite loss from a replayed step, otherwise all
            # ranks fall back to eager launches together.
            ok = True
            try:
                engine.capture_step(video, mask, targets, L=L, segmented=True)
            except Exception as e:           # noqa: BLE001
                print(f"[bench] rank {rank}: segmented capture failed ({e!r})", file=sys.stderr, flush=True)
                ok = False
            if all_ranks_ok(ok):
                try:
                    l_, _ = engine.train_step_graphed()
                    ok = bool(torch.isfinite(l_).item())
                except Exception as e:       # noqa: BLE001
                    print(f"[bench] rank {rank}: replaying the segmented step failed ({e!r})", file=sys.stderr, flush=True)
                    ok = False
                if all_ranks_ok(ok):
                    dist_mode, step = "graph-segments", engine.train_step_graphed
                else:
                    dist_note = "graph-segments replay failed on some rank -> eager"
            else:
                dist_note = "graph-segments capture failed on some rank -> eager"
            if dist_mode == "auto":
                back_to_eager()
                dist_mode = "eager"
        elif dist_mode == "graph-segments":
            engine.capture_step(video, mask, targets, L=L, segmented=True)
            step = engine.train_step_graphed
        elif dist_mode == "graph":
            engine.capture_step(video, mask, targets, L=L, defer_reduce=True)
            step = engine.train_step_graphed
        elif dist_mode == "graph-overlap":
            engine.capture_step(video, mask, targets, L=L, capture_comm=True)
            step = engine.train_step_graphed
    for _ in range(args.warmup):
        loss, _ = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    graphed_any = graphed or dist_mode in ("graph", "graph-overlap", "graph-segments")
    prof = None if (args.no_kernel_events or graphed_any) else []
    kprof = None if prof is None else []
    eager_ms = None
    ops.GEMM_PROFILE, ops.KERNEL_PROFILE = prof, kprof
    model.dp_count_acc.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host_step = []
    for _ in range(args.steps):
        th = time.perf_counter()
        loss, _ = step()
        host_step.append(time.perf_counter() - th)
    t_enqueued = time.perf_counter() - t0                      # host time to enqueue all steps (no sync inside a step)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.GEMM_PROFILE = ops.KERNEL_PROFILE = None
    # HBM in use by the timed workload (weights + fp32 optimizer state + flat gradient buffers + the step's activations and graph pools)
    hbm = None
    if dev.type == "cuda":
        hbm = {"peak_allocated_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1), "reserved_gb": round(torch.cuda.memory_reserved(dev) / 2 ** 30, 1),
               "capacity_gb": round(torch.cuda.get_device_properties(dev).total_memory / 2 ** 30, 1), "at": "end of the timed region (before the B = 32 / secondary blocks)"}
    dp = droppath_account(model, args.steps, B, L, spec["flop"])     # executed vs nominal FLOPs of the timed steps
    events_from, event_steps = "timed steps", args.steps
    marks = []                                                   # (label, event) at the encoder's boundaries of the eager event pass, if it runs
    if graphed_any and not args.no_kernel_events:
        # HIP events cannot be recorded inside a graph replay: the per-launch GEMM events come from eager steps of the same
        # workload, run right after the timed region (they include the host-side launch gaps the graph removes)
        eager_step()                                             # untimed: the allocator refills its pools after graph mode
        prof, kprof = [], []
        ops.GEMM_PROFILE, ops.KERNEL_PROFILE = prof, kprof
        from internvideo_amd import functional as Fn
        Fn.STEP_MARKS = marks = []                               # encoder boundaries of these eager steps (encoder_fwd_bwd_frac below)
        torch.cuda.synchronize()
        t_e0 = time.perf_counter()
        for _ in range(2):
            eager_step()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t_e0) / 2 * 1e3
        ops.GEMM_PROFILE = ops.KERNEL_PROFILE = None
        Fn.STEP_MARKS = None
        events_from, event_steps = "2 eager steps of the same workload (after 1 untimed eager step) following the timed (graph-replayed) region", 2
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss.item())

    # north_star's literal target: the ENCODER's forward + backward (patch embed + 40 blocks; decoders / attention pool / loss / AdamW
    # excluded) against the MFMA peak.  Its share of the step comes from HIP events at the encoder's boundaries in the eager pass above
    # (functional.STEP_MARKS); the share is applied to the timed (graph-replayed) step.
    encoder = None
    if graphed_any and not args.no_kernel_events and args.model == "1B" and not args.with_teachers:
        try:
            lab = {}
            for name, ev in marks:
                lab.setdefault(name, []).append(ev)
            n_e = min(len(v) for v in lab.values())
            enc_ms = sum(lab["enc_fwd_begin"][i].elapsed_time(lab["enc_fwd_end"][i]) + lab["enc_bwd_begin"][i].elapsed_time(lab["enc_bwd_end"][i])
                         for i in range(n_e)) / n_e
            share = enc_ms / eager_ms
            step_ms = elapsed / args.steps * 1e3
            enc_step_ms = share * step_ms
            enc_exec = FLOP_PER_CLIP_ENCODER - (dp["nominal_flop_per_clip"] - dp["executed_flop_per_clip"])     # the skipped FLOPs are all the encoder's
            encoder = dict(encoder_fwd_bwd_frac=round(B * enc_exec / (enc_step_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                           encoder_fwd_bwd_frac_nominal_equivalent=round(B * FLOP_PER_CLIP_ENCODER / (enc_step_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                           encoder_ms_per_step=round(enc_step_ms, 2), encoder_share_of_step=round(share, 4),
                           flop_per_clip=FLOP_PER_CLIP_ENCODER, executed_flop_per_clip=round(enc_exec, 1),
                           heads_loss_optimizer_ms_per_step=round(step_ms - enc_step_ms, 2),
                           estimate=True,
                           how="HIP events at the encoder's boundaries (patch embed .. last block; block-stack backward .. patch-embed backward) in "
                               "the 2 eager steps after the timed region; that share of the eager step applied to the timed graph-replayed step")
        except Exception as e:       # noqa: BLE001
            encoder = {"error": repr(e)[:200]}

    roofline = None
    if prof:
        kinds = {}
        for ent in prof:
            kern, a_kc, b_kc, fl, e0, e1 = ent[:6]
            fl = fl * _dyn_scale(ent[6] if len(ent) > 6 else None)       # launches under a device-side row count: the FLOPs they executed
            k = kinds.setdefault((kern, a_kc, b_kc), [0.0, 0.0, 0])
            k[0] += fl; k[1] += e0.elapsed_time(e1) * 1e-3; k[2] += 1
        role = {(1, 1): "forward NT", (1, 0): "dgrad", (0, 0): "wgrad", (0, 1): "TN"}
        kname = {2: "gemm256_kernel", 8: "gemm_fp8_kernel"}
        names = {k: f"{kname.get(k[0], 'gemm_bf16_kernel')}<{k[1]},{k[2]}> ({'e4m3: forward / dgrad / wgrad' if k[0] == 8 else role[k[1:]]})" for k in kinds}
        tot_fl = sum(v[0] for v in kinds.values()); tot_t = sum(v[1] for v in kinds.values())
        dom = max(kinds, key=lambda k: kinds[k][1])
        fl, tt, n = kinds[dom]
        kkey = names[dom].split(" ")[0]
        traffic = _stamped(os.path.join(ROOT, "profiles", "pmc_traffic.json"), kkey)
        mfma_util = _stamped(os.path.join(ROOT, "profiles", "pmc_mfma_util.json"), {"gemm256_kernel<1,1>": "gemm256_kernel<true, true, 0, false, 0, 0, false, false>",
                             "gemm256_kernel<1,0>": "gemm256_kernel<true, false, 0, false, 0, 0, false, false>",
                             "gemm256_kernel<0,0>": "gemm256_kernel<false, false, 0, true, 0, 0, false, false>"}.get(kkey, kkey))
        peak = 5000.0 if dom[0] == 8 else PEAK_BF16_TFLOPS     # dense MX-fp8 MFMA peak when the dominant GEMM is the e4m3 kernel
        roofline = dict(bound="mfma", kernel=names[dom], events_from=events_from, achieved=round(fl / tt / 1e12, 1), peak=peak, unit="TFLOP/s",
                        frac=round(fl / tt / 1e12 / peak, 4), traffic=traffic, mfma_util_pmc=mfma_util,
                        eager_ms_per_step_during_events=(round(eager_ms, 2) if eager_ms else None),
                        launches=n, avg_launch_us=round(tt / n * 1e6, 1), flop_per_launch=round(fl / n / 1e9, 2),
                        gemm_family=dict(achieved=round(tot_fl / tot_t / 1e12, 1), frac=round(tot_fl / tot_t / 1e12 / PEAK_BF16_TFLOPS, 4),
                                         time_share_of_step=round((tot_t / event_steps) / (elapsed / args.steps), 3),
                                         by_kernel={names[k]: dict(tflops=round(v[0] / v[1] / 1e12, 1), launches=v[2],
                                                                   avg_launch_us=round(v[1] / v[2] * 1e6, 1)) for k, v in kinds.items()}))

    # the other kernels of the step, live per-launch HIP events of the same eager pass: HBM-bound row kernels against the 8 TB/s spec,
    # attention against the bf16 MFMA peak (its HBM floor is noted in DESIGN.md)
    other = None
    if kprof:
        agg = {}
        for ent in kprof:
            name, work, unit, e0, e1 = ent[:5]
            work = work * _dyn_scale(ent[5] if len(ent) > 5 else None)
            a = agg.setdefault(name, [0.0, 0.0, 0, unit])
            a[0] += work; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
        other = {}
        for name, (work, tt, n, unit) in sorted(agg.items()):
            if unit == "B":
                other[name] = dict(bound="hbm", achieved=round(work / tt / 1e9, 1), peak=8000.0, unit="GB/s", frac=round(work / tt / 8e12, 4),
                                   launches=n, avg_launch_us=round(tt / n * 1e6, 1), ms_per_step=round(tt / event_steps * 1e3, 2))
            else:
                other[name] = dict(bound="mfma", achieved=round(work / tt / 1e12, 1), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                                   frac=round(work / tt / 1e12 / PEAK_BF16_TFLOPS, 4), launches=n, avg_launch_us=round(tt / n * 1e6, 1),
                                   ms_per_step=round(tt / event_steps * 1e3, 2))

    # measured contention term of the scaling prediction (before the engine's graph is re-captured for the B = 32 block below)
    contention = None
    if world == 1 and graphed and args.model == "1B" and not args.no_contention:
        try:
            contention = comm_contention(engine, step, elapsed / args.steps * 1e3, dev)
        except Exception as e:       # noqa: BLE001
            contention = {"error": repr(e)[:200]}

    # secondary block: the reference recipe's per-GPU batch (scripts/pretraining/1B_pt.sh:50), same engine, re-captured on B = 32 inputs
    b32 = None
    if world == 1 and graphed and args.batch != 32 and args.model == "1B" and not args.no_b32:
        try:
            Bs = 32
            v32, m32 = video[:Bs].clone(), mask[:Bs].clone()
            t32 = tuple(t[:, :Bs].clone() if t.dim() == 4 else t[:Bs].clone() for t in targets)
            engine.capture_step(v32, m32, t32, L=L)
            for _ in range(3):
                engine.train_step_graphed()
            torch.cuda.synchronize()
            model.dp_count_acc.zero_()
            t1 = time.perf_counter()
            n32 = max(10, args.steps)
            for _ in range(n32):
                engine.train_step_graphed()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / n32
            dp32 = droppath_account(model, n32, Bs, L, spec["flop"])
            b32 = dict(per_gpu_batch=Bs, steps=n32, ms_per_step=round(dt * 1e3, 2), clips_per_s=round(Bs / dt, 2),
                       mfma_frac_of_step=round(Bs / dt * dp32["executed_flop_per_clip"] / 1e12 / PEAK_BF16_TFLOPS, 4),
                       mfma_frac_nominal_equivalent=round(Bs / dt * spec["flop"] / 1e12 / PEAK_BF16_TFLOPS, 4),
                       droppath_skip=dp32)
        except Exception as e:
            b32 = {"error": repr(e)}

    if rank == 0:
        clips = args.steps * B * world
        value = clips / elapsed
        out = {
            "metric": ("clips/sec, InternVideo2-1B stage-1 recipe step incl. frozen InternVL-6B + VideoMAE-g teachers, 16x224^2 clips, bf16 (whole job)"
                       if args.with_teachers else "clips/sec, InternVideo2-1B stage-1 pretrain step 8x224^2 bf16 (whole job)") if args.model == "1B"
                      else ("clips/sec, InternVideo2-6B encoder pretrain step 16x224^2 (whole job)" if args.model == "6B"
                            else "clips/sec, InternVideo2-B/14 pretrain step 8x224^2 bf16 (whole job)"),
            "value": round(value, 2), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": (f"fp8 (e4m3 block GEMMs, {args.fp8_scaling} per-tensor scaling{', per-channel weight scales' if args.fp8_weight_scales == 'channel' else ''}; bf16 attention / norms, {args.residual} residual, fp32 optimizer state; "
                                            f"forward at this shape (16 x 224^2, L = 833, 48 blocks) against the reference's fp32 CPU forward: loss within 2e-4 relative, head outputs "
                                            f"3.3-5.3e-2 rel-L2 where bf16 GEMMs give 5e-3 -- tests/test_fullsize_gpu.py::test_6B_encoder_at_its_own_shape_bf16_and_fp8_match_the_reference_digest)"
                                            if args.fp8 else ("bf16 student step; frozen CLIP teacher's block GEMMs in fp8 e4m3 (opt-in --teacher-fp8; the reference "
                                                              "runs its teachers in bf16)" if (args.with_teachers and args.teacher_fp8) else "bf16")),
            "data": "synthetic", "checkpoint_num": args.checkpoint_num,
            "config": {"workload": (f"InternVideo2-{args.model} stage-1 recipe step (engine_for_pretraining.py:63-148): 16x224^2 clips -> frozen InternVL-6B CLIP "
                                    f"teacher (8 frames) + VideoMAE-g teacher (16 frames) -> attention-guided mask 0.8 -> visible targets -> student step "
                                    f"(fwd + fused distill loss + bwd + grad all-reduce + AdamW), L={L}, drop_path {args.drop_path}") if args.with_teachers else
                                   (f"InternVideo2-{args.model} stage-1 student step (fwd + fused distill loss + bwd + grad all-reduce + AdamW), "
                                    f"{T}x224^2, mask 0.8 -> L={L}, clip_return_layer 6, mae_return_layer 4, drop_path {args.drop_path}"),
                       "model": {"1B": "pretrain_internvideo2_1B_patch14_224", "6B": "pretrain_internvideo2_6B_patch14_224"}.get(args.model, "InternVideo2-B/14"),
                       "params": n_params, "global_batch": B * world, "per_gpu_batch": B, "seq_len": L, "parallelism": f"dp{world}",
                       "weights": "random init (reference init), " + ("random-weight teachers" if args.with_teachers else "synthetic teacher targets")},
            "clips_per_sec_per_gpu": round(value / world, 2),
            # priced on the FLOPs the step EXECUTED (DropPath skipping removes the dropped (block, branch, sample) triples); the figure the
            # same clips/s would mean if every triple were computed is printed beside it and is not a utilisation claim
            "mfma_frac_of_step": round(value / world * (dp["executed_flop_per_clip"] + (29.7e12 if args.with_teachers else 0.0)) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "mfma_frac_nominal_equivalent": round(value / world * (spec["flop"] + (29.7e12 if args.with_teachers else 0.0)) / 1e12 / PEAK_BF16_TFLOPS, 4),
            "droppath_skip": dp,
            "encoder_fwd_bwd_frac": (encoder or {}).get("encoder_fwd_bwd_frac"),
            "encoder": encoder,
            "loss": round(loss_val, 5),
            # host time of one step's enqueue = the MEDIAN over the timed steps.  The mean is not a launch cost: the host runs ahead of the
            # GPU until the runtime's queue pushes back (~14 replayed steps deep), after which every call waits for a GPU step to retire --
            # 20 steps of 391 ms then "cost" 107 ms of host time each (BENCH_r03), 10 steps cost 0.5 ms each.  Both are reported.
            "host_enqueue_ms_per_step": round(sorted(host_step)[len(host_step) // 2] * 1e3, 2),
            "host_enqueue_mean_ms_incl_queue_backpressure": round(t_enqueued / args.steps * 1e3, 2),
            "host_steps_before_backpressure": next((i for i, t in enumerate(host_step) if t > 0.25 * elapsed / args.steps), len(host_step)),
            "teachers": ("InternVL_CLIP 6B (48 x 3200, 25 heads, 257-token frame sequences) + VideoMAE-g (40 x 1408, 2048 tokens), random weights, "
                         "24.6 + 5.1 TFLOP forward per clip (SURVEY.md 8(a) a19)" + ("; CLIP teacher block GEMMs in fp8 (e4m3, opt-in: --teacher-fp8)" if args.teacher_fp8 else "")) if args.with_teachers else None,
            "launch_mode": "hip graph replay + eager AdamW" if graphed else
                           ({"graph": "hip graph replay, then bucketed RCCL all-reduce (no overlap), eager AdamW",
                             "graph-overlap": "hip graph replay incl. the bucketed RCCL collectives on the side stream (overlapped), eager AdamW",
                             "graph-segments": "chain of hip graphs cut at the gradient buckets, eager RCCL all-reduce of each bucket on the side "
                                               "stream between them (overlapped with the following segments' backward), eager AdamW",
                             "eager": "eager launches, bucketed RCCL all-reduce overlapped with backward"}.get(dist_mode, "eager")),
            "residual_stream": args.residual,
            "hbm": hbm,
            "dist_mode": dist_mode, "dist_note": dist_note,
            "rccl_ranks": (dist.get_world_size() if (world > 1 or args.force_dist) else 1),
            "backend": (dist.get_backend() if (world > 1 or args.force_dist) else "none"),
            "shared_gpu": (True if args.share_gpu else None),
            "graph_segments": (len(engine._segments) if getattr(engine, "_segments", None) else None),
            "reduce_buckets": len(engine.reduce_log),
            "roofline": roofline,
            "other_kernels": other,
            "b32": b32,
            "attn_kernel": {0: "auto (32x32x16 MFMA)", 1: "16x16x32 MFMA", 2: "32x32x16 MFMA"}[args.attn_kernel],
            "reduce": f"{args.reduce_mode}/{args.reduce_dtype}" if (world > 1 or args.force_dist) else "n/a",
        }
        if world > 1 or args.force_dist:
            out["scaling_model"] = scaling_model(engine, max(world, 1), elapsed / args.steps)
        else:
            # written BEFORE any multi-GPU run exists, so that the driver's 1 -> 8 curve can be read against it: the same
            # arithmetic for N = 2, 4, 8 with this run's step time and the bucket plan every rank would build
            out["scaling_model_predictions"] = {str(n): scaling_model(engine, n, elapsed / args.steps, reduce_mode="allreduce", reduce_dtype="fp32")
                                                for n in (2, 4, 8)}
            if contention is not None:                          # measured on this GPU: what collective kernels beside the step cost it
                out["scaling_model_predictions"]["comm_contention_measured"] = contention
            strag = droppath_straggler(model, B, L, elapsed / args.steps * 1e3, dp)
            out["scaling_model_predictions"]["droppath_skip_straggler"] = strag
            # one number per N: the ideal-links model x the contention measured here (fp32 wire through 16 CUs: the default reduction; it is
            # measured with the N = 8 wire bytes and applied unchanged to N = 2, 4, which move 4/7 and 6/7 of them) x the straggler model
            comb = {}
            for n in ("2", "4", "8"):
                e = out["scaling_model_predictions"][n]["predicted_scaling_efficiency_all_links"]
                if contention and "efficiency_with_contention" in contention:
                    e *= contention["efficiency_with_contention"]["fp32_wire_16_cus"]
                if strag:
                    e *= strag["efficiency_factor"][n]
                comb[n] = {"efficiency": round(e, 4), "speedup": round(int(n) * e, 3)}
            out["scaling_model_predictions"]["with_measured_contention_and_straggler_model"] = comb
        if world == 1 and args.model == "1B" and not (args.no_secondary or args.with_teachers or args.fp8 or args.force_dist):
            out["secondary"] = secondary_lines()
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(spec, args.cpu_iters)
            except Exception as e:       # never lose the GPU number to a host-side problem
                out["cpu_baseline"] = {"error": repr(e)}
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or args.force_dist:
        # Teardown (tools/rccl_teardown_probe.py, profiles/r5_rccl_teardown_probe_1rank_v1.json: 5 trials per cell on a
        # 1-rank RCCL group).  dist.destroy_process_group() itself never aborted, in any mode or order; the SIGABRT of round 3 happens at
        # INTERPRETER EXIT, after a clean teardown, and only when collectives were captured INSIDE a HIP graph (graph-overlap: 2 of 10; graph-
        # segments -- the default -- and eager: 0 of 20).  So the default modes leave the way a training script does: graphs, then streams, then
        # the communicator (engine.close(); barrier; destroy), guarded by a watchdog because a teardown that hangs must not turn a finished,
        # printed measurement into a killed job; graph-overlap keeps the hard exit (and --hard-exit forces it for any mode).
        hard = args.hard_exit or dist_mode == "graph-overlap"
        if dist.get_backend() != "nccl" or not hard:
            import threading
            t_kill = threading.Timer(90.0, lambda: os._exit(0))      # the line is out: never let teardown cost the run
            t_kill.daemon = True
            t_kill.start()
            try:
                engine.close()
                dist.barrier()
                torch.cuda.synchronize()
                dist.destroy_process_group()
            finally:
                t_kill.cancel()
        else:
            dist.barrier()
            torch.cuda.synchronize()
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(0)

if __name__ == "__main__":
    main()
