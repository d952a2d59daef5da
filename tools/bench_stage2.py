"""Stage-2 (BASELINE configs[3]) step on one MI355X: InternVideo2-1B vision tower (4 x 224^2, random mask 0.8 -> 206 tokens) + BERT-large text /
fusion tower, all four losses (UTA against synthetic teacher targets, VTC, VTM, MLM), forward + backward.  GPU box only.

    python tools/bench_stage2.py [--batch 64] [--steps 5] [--warmup 2]

One JSON line: clips/s, ms per step and its split (vision tower forward, text side forward, backward) from HIP events.  The towers run
in the drop-in (plain autograd) mode: the fused optimizer / gradient buckets of internvideo_amd.engine are built around the stage-1
student and are not part of this line -- it is the model-side cost of a stage-2 step (reference: multi_modality/tasks/pretrain.py:63-121
without `optimizer.step`).  Random-init weights, synthetic clips / token ids (no network), dropout 0 in the text tower."""
import argparse
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib  # noqa: E402

from internvideo_amd import functional as Fn, masking, xbert  # noqa: E402
from internvideo_amd.stage2 import InternVideo2_Stage2_visual  # noqa: E402

DEV = "cuda"


class Stage2WithSyntheticTeacher(InternVideo2_Stage2_visual):
    """the frozen InternVL-6B CLIP teacher replaced by random l2-normalised targets of its output shapes (K = 6 taps x visible tokens x
    3200, final 768); masks from the reference's random generator (multi_modality/models/mask.py:22-37)"""

    static = None                                        # --graph: (mask, middle targets, final target) refreshed by the caller between replays

    @torch.no_grad()
    def encode_teacher(self, image):
        if self.static is not None:
            return self.static
        B, C, T, H, W = image.shape
        mask = masking.with_cls_column(masking.random_masks(self.video_window_size, self.video_mask_ratio, B, image.device))
        n_vis = int((~mask[0]).sum())
        K = 6
        mid = torch.nn.functional.normalize(torch.randn(K, B, n_vis, 3200, device=image.device), dim=-1).to(torch.bfloat16)
        fin = torch.nn.functional.normalize(torch.randn(B, 768, device=image.device), dim=-1)
        return mask, mid, fin


def cpu_baseline_vision_tower():
    """the CPU leg of the stage-2 line, bounded: the oracle (port of the reference's unfused fp32 path) forward + backward of the 1B VISION tower on
    one stage-2 clip (4 x 224^2, 206 visible tokens) on this box's host cores.  The text / fusion tower's port lives in the oracle too but one
    BERT-large step on the host would take the sample past its budget; said here so that the ratio is not read as a whole-step figure."""
    import time
    from internvideo_amd.hostinfo import usable_cores
    from oracle import internvideo2_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = O.StudentConfig(embed_dim=1408, depth=40, num_heads=16, mlp_ratio=48 / 11, num_frames=4, clip_return_layer=6, has_mae=False,
                          sep_image_video_pos_embed=True)
    params = {k: v.requires_grad_(True) for k, v in O.synthetic_params(cfg, seed=3).items()}
    video, mask, _ = O.synthetic_batch(cfg, 1, (1024 - int(1024 * 0.8)) // 4, seed=0)
    ts = []
    for it in range(3):
        t0 = time.perf_counter()
        out = O.encoder_forward(params, video, mask, cfg)
        sum(out[k].float().pow(2).mean() for k in ("x_vis", "x_pool_vis", "x_clip_align", "x_align")).backward()
        for p_ in params.values():
            p_.grad = None
        ts.append(time.perf_counter() - t0)
    t = float(np.mean(ts[1:]))
    return dict(value=round(1.0 / t, 4), unit="clips/s", cores=cores, kind="port", scope="vision_tower_only", comparable_to_value=False,
                sample=f"CPU oracle, 1B vision tower ONLY (no text tower), fwd+bwd, 1 clip 4x224^2 L=205 (51 visible patches per frame + cls), 2 timed iterations after 1 warm-up, {t:.2f} s/clip")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--text-len", type=int, default=32)
    ap.add_argument("--graph", action="store_true", help="capture forward + backward of the whole stage-2 model into one HIP graph and replay it "
                    "(static inputs; the mask check, the temperature and the visible-token count are device-side, no host read is left)")
    ap.add_argument("--batch-text", action="store_true", help="one text-mode pass over [ids | masked ids] and one fusion pass over the VTM pairs + "
                    "the MLM rows (InternVideo2_Stage2_visual.batch_text_passes) instead of two passes each")
    ap.add_argument("--group-wgrad", action="store_true", help="weight gradients of the text / fusion tower's Linear layers as grouped GEMMs at the end "
                    "of the backward pass (functional.grouped_weight_grads; bypasses autograd hooks on those parameters)")
    ap.add_argument("--engine", action="store_true", help="a TRAINING step (multi_modality/tasks/pretrain.py:207-213 under scripts/pretraining/stage2/1B/"
                    "config.py:97-102): IVTrainEngine over the stage-2 model -- flat fp32 master / bf16 compute copies, text-tower gradients accumulated "
                    "into zeroed flat buffers, gradient clipping 3.0, fused AdamW (betas 0.9 / 0.98, wd 0.05), BERT's configured dropout (0.1) with the "
                    "device-side epoch; --graph then captures forward + backward with engine.capture_fn and AdamW stays eager")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gpus", type=int, default=1, help="N > 1 (BASELINE configs[3]: 8 GPUs, RCCL all-gather of the VTC negatives): one rank per GPU under "
                    "torch.distributed.run (bench.py --model stage2-1B --gpus N starts them); needs --engine: bucketed gradient reduction overlapped with "
                    "the vision tower's backward, the packed feature all-gather inside the forward; eager launches (a collective sits inside the "
                    "forward, so the single-graph capture is a 1-GPU mode)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a 1-GPU box: every rank on cuda:0, collectives over gloo staged through host memory")
    ap.add_argument("--reduce-mode", default="allreduce", choices=["allreduce", "zero1"])
    ap.add_argument("--dropout", type=float, default=None, help="hidden / attention dropout of the text tower (default: 0.1 = config_bert_large.json with "
                    "--engine, 0 without: a captured plain-autograd step cannot advance the masks)")
    ap.add_argument("--residual", default="bf16", choices=["bf16", "fp32"], help="residual stream of the vision tower: bf16 = what the reference's bf16 "
                    "recipe carries (use_half_precision / use_bf16 of the stage-2 config), fp32 = the parity setting")
    a = ap.parse_args()
    if a.dropout is None:
        a.dropout = 0.1 if a.engine else 0.0
    gw = Fn.grouped_weight_grads if (a.group_wgrad and not a.engine) else contextlib.nullcontext
    import torch.distributed as dist
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus or world == 1 and a.gpus == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    json_fd = None
    if world > 1:
        if not a.engine:
            raise SystemExit("bench_stage2 --gpus N needs --engine (the plain-autograd mode has no gradient reduction)")
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench as B_
        json_fd = B_._reserve_stdout()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime
        if a.share_gpu:
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
            B_._host_staged_collectives()
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=600))
        a.graph = False                                     # see --gpus
    torch.manual_seed(0)                                    # identical weights on every rank
    np.random.seed(0)
    ve = dict(name="pretrain_internvideo2_1b_patch14_224", img_size=224, num_frames=4, tubelet_size=1, patch_size=14, d_model=1408, clip_embed_dim=768,
              clip_teacher_embed_dim=3200, clip_teacher_final_dim=768, clip_norm_type="l2", clip_return_layer=6, clip_student_return_interval=1,
              pretrained=None, use_checkpoint=False, checkpoint_num=40, use_flash_attn=True, use_fused_rmsnorm=True, use_fused_mlp=True,
              sep_image_video_pos_embed=True, clip_teacher=None, clip_input_resolution=224, video_mask_type="random", video_mask_ratio=0.8,
              image_mask_type="random", image_mask_ratio=0.5)
    te = dict(name="bert_large", d_model=1024, fusion_layer=19,
              config=dict(hidden_dropout_prob=a.dropout, attention_probs_dropout_prob=a.dropout))       # config_bert_large.json otherwise
    config = dict(model=dict(vision_encoder=ve, text_encoder=te, multimodal=dict(enable=True), embed_dim=512, temp=0.07),
                  criterion=dict(loss_weight=dict(vtc=1.0, mlm=1.0, vtm=1.0, uta=1.0), vtm_hard_neg=True, mlm_masking_prob=0.5,
                                 distill_final_features=True, clip_loss_ratio=[1.0, 1.0]), gradient_checkpointing=False)
    tok = SimpleNamespace(pad_token_id=0, cls_token_id=101, mask_token_id=103)
    model = Stage2WithSyntheticTeacher(config, tok, True).to(DEV).train()
    model.batch_text_passes = bool(a.batch_text)
    model.vision_encoder.residual_dtype = a.residual
    n_vision = sum(p.numel() for p in model.vision_encoder.parameters())
    n_text = sum(p.numel() for p in model.text_encoder.parameters())
    B, L = a.batch, a.text_len
    torch.manual_seed(1000 + rank); np.random.seed(1000 + rank)          # per-rank data / masks / MLM draws (tasks/pretrain.py:332 seed + rank)
    rng = np.random.RandomState(rank)
    lens = rng.randint(8, L + 1, size=B)
    ids = np.zeros((B, L), dtype=np.int64)
    att = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        ids[b, 0] = 101
        ids[b, 1:lens[b]] = rng.randint(1000, 30522, size=lens[b] - 1)
        att[b, :lens[b]] = 1
    text = SimpleNamespace(input_ids=torch.from_numpy(ids).to(DEV), attention_mask=torch.from_numpy(att).to(DEV))
    image = torch.randn(B, 4, 3, 224, 224, device=DEV).to(torch.bfloat16)
    idx = torch.arange(B, device=DEV) + rank * B            # distinct clip ids over the job (the VTC targets compare them)
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    times, parts, losses = [], [], None
    graph = None
    engine = None
    if a.engine:
        from internvideo_amd.engine import IVTrainEngine
        engine = IVTrainEngine(model, lr=5e-5, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.05, max_grad_norm=3.0, reduce_mode=a.reduce_mode)
        engine.group_text_wgrads = bool(a.group_wgrad)
        g_out = {}

        def loss_fn():
            out = model(image, text, idx, media_type="video")
            g_out.update(out)
            return sum(out.values())
        if a.graph:
            model.static = None
            m0, t_mid, t_fin = model.encode_teacher(image.permute(0, 2, 1, 3, 4))
            model.static = (m0, t_mid, t_fin)
            model.vision_encoder.static_visible_tokens = int((~m0[0]).sum())
            text.attention_mask._ivh_kv_len = text.attention_mask.sum(1, dtype=torch.int32).contiguous()
            engine.capture_fn(loss_fn)
    if a.graph and not a.engine:
        # static side inputs: vision mask + teacher targets (refreshed between replays by copy_), text lengths next to the attention mask
        model.static = None
        m0, t_mid, t_fin = model.encode_teacher(image.permute(0, 2, 1, 3, 4))
        model.static = (m0, t_mid, t_fin)
        model.vision_encoder.static_visible_tokens = int((~m0[0]).sum())
        kv_len = text.attention_mask.sum(1, dtype=torch.int32).contiguous()
        text.attention_mask._ivh_kv_len = kv_len            # right-padded by construction: no host check inside the captured region
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                                 # warm-up on the capture stream's allocator
                model.zero_grad(set_to_none=True)
                with gw():
                    sum(model(image, text, idx, media_type="video").values()).backward()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        model.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        from internvideo_amd.engine import _cyclic_gc_paused
        with _cyclic_gc_paused(), torch.cuda.graph(graph):
            g_out = model(image, text, idx, media_type="video")
            g_total = sum(g_out.values())
            with gw():
                g_total.backward()
        torch.cuda.synchronize()
    if world > 1:
        import time
        for _ in range(a.warmup):
            engine.train_step_fn(loss_fn)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            engine.train_step_fn(loss_fn)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=DEV)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        ms_n = float(el.item()) / a.steps * 1e3
        losses_n = {k: float(v.detach()) for k, v in g_out.items()}
        if rank == 0:
            sm = B_.scaling_model(engine, world, ms_n * 1e-3)
            line = dict(metric="clips/sec, InternVideo2 stage-2 1B TRAINING step (vision 1B + BERT-large, UTA + VTC + VTM + MLM), whole job",
                        value=round(B * world / ms_n * 1e3, 2), unit="clips/s", n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=round(ms_n, 2),
                        higher_is_better=True, scaling="weak", vs_baseline=None, dtype="bf16", data="synthetic",
                        config=dict(workload="InternVideo2 stage-2 1B training step (scripts/pretraining/stage2/1B/config.py: 4x224^2, mask 0.8, max_txt_l 32; "
                                             "four losses; VTC negatives all-gathered over the ranks)", per_gpu_batch=B, global_batch=B * world,
                                    parallelism=f"dp{world}"),
                        losses=losses_n, launch_mode="eager forward + backward (bucketed reduction overlapped with the vision backward) + clip + fused AdamW",
                        reduce=f"{a.reduce_mode}/bf16", backend=dist.get_backend(), shared_gpu=(True if a.share_gpu else None),
                        reduce_buckets=len(engine.reduce_log), text_tower_dropout=a.dropout, scaling_model=sm)
            os.write(json_fd, (json.dumps(line) + "\n").encode())
        if dist.get_backend() != "nccl":
            dist.destroy_process_group()
        else:                                               # as bench.py: no communicator teardown after a finished measurement
            dist.barrier(); torch.cuda.synchronize(); sys.stdout.flush(); sys.stderr.flush()
            os._exit(0)
        return
    for it in range(a.warmup + a.steps):
        e0, e1, e2 = ev(), ev(), ev()
        if engine is not None:
            e0.record()
            if a.graph:
                engine.train_step_graphed()
            else:
                engine.train_step_fn(loss_fn)
            e1.record(); e2.record()
            out = g_out
        elif graph is not None:
            # a new batch would be copied into image / text.input_ids / text.attention_mask / kv_len / model.static here
            e0.record()
            graph.replay()
            e1.record(); e2.record()
            out = g_out
        else:
            model.zero_grad(set_to_none=True)
            e0.record()
            out = model(image, text, idx, media_type="video")
            e1.record()
            total = sum(out.values())
            with gw():
                total.backward()
            e2.record()
        torch.cuda.synchronize()
        if it >= a.warmup:
            times.append(e0.elapsed_time(e2))
            parts.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
        losses = {k: float(v.detach()) for k, v in out.items()}
    ms = float(np.median(times))
    # roofline of the step: per-launch HIP events around every GEMM / attention / row kernel of ONE eager step of the same workload (events cannot
    # be recorded inside a graph replay), against the 2.5 PFLOP/s dense bf16 MFMA peak
    from internvideo_amd import ops
    prof, kprof = [], []
    ops.GEMM_PROFILE, ops.KERNEL_PROFILE = prof, kprof
    if engine is not None:
        engine.train_step_fn(loss_fn)
    else:
        model.zero_grad(set_to_none=True)
        out_e = model(image, text, idx, media_type="video")
        with gw():
            sum(out_e.values()).backward()
    torch.cuda.synchronize()
    ops.GEMM_PROFILE = ops.KERNEL_PROFILE = None
    kinds = {}
    from bench import _dyn_scale                               # launches under a device-side row count (DropPath skipping): executed FLOPs
    for ent in prof:
        kern, a_kc, b_kc, fl, e0, e1 = ent[:6]
        fl = fl * _dyn_scale(ent[6] if len(ent) > 6 else None)
        k_ = kinds.setdefault((kern, a_kc, b_kc), [0.0, 0.0, 0])
        k_[0] += fl; k_[1] += e0.elapsed_time(e1) * 1e-3; k_[2] += 1
    role = {(1, 1): "forward NT", (1, 0): "dgrad", (0, 0): "wgrad", (0, 1): "TN"}
    name = lambda k_: f"{'gemm256_kernel' if k_[0] == 2 else 'gemm_bf16_kernel'}<{k_[1]},{k_[2]}> ({role[k_[1:]]})"   # noqa: E731
    tot_fl = sum(v[0] for v in kinds.values()); tot_t = sum(v[1] for v in kinds.values())
    att_fl = sum(ent[1] * _dyn_scale(ent[5] if len(ent) > 5 else None) for ent in kprof if ent[2] != "B")
    dom = max(kinds, key=lambda k_: kinds[k_][1])
    roofline = dict(bound="mfma", kernel=name(dom), events_from="1 eager step of the same workload after the timed region",
                    achieved=round(kinds[dom][0] / kinds[dom][1] / 1e12, 1), peak=2500.0, unit="TFLOP/s",
                    frac=round(kinds[dom][0] / kinds[dom][1] / 2.5e15, 4), launches=kinds[dom][2], traffic=None,
                    gemm_family=dict(achieved=round(tot_fl / tot_t / 1e12, 1), frac=round(tot_fl / tot_t / 2.5e15, 4),
                                     time_share_of_step=round(tot_t / (ms * 1e-3), 3), gemm_launches=sum(v[2] for v in kinds.values()),
                                     by_kernel={name(k_): dict(tflops=round(v[0] / v[1] / 1e12, 1), launches=v[2],
                                                               avg_launch_us=round(v[1] / v[2] * 1e6, 1)) for k_, v in kinds.items()}))
    cpu = None
    if a.engine and not a.no_cpu_baseline:
        cpu = cpu_baseline_vision_tower()
    print(json.dumps(dict(metric=("clips/sec, InternVideo2 stage-2 1B TRAINING step (vision 1B + BERT-large, UTA + VTC + VTM + MLM): forward + backward + "
                                  "gradient clipping + fused AdamW, 1 GPU") if a.engine else
                                 "clips/sec, InternVideo2 stage-2 1B step (vision 1B + BERT-large, UTA + VTC + VTM + MLM), forward + backward, 1 GPU",
                          value=round(B / ms * 1e3, 2), unit="clips/s", ms_per_step=round(ms, 2),
                          forward_ms=None if a.graph else round(float(np.median([p[0] for p in parts])), 2),
                          backward_ms=None if a.graph else round(float(np.median([p[1] for p in parts])), 2), batch=B, vision_tokens=206, text_len=L,
                          params_vision=n_vision, params_text=n_text, losses=losses, dtype="bf16", data="synthetic", residual_stream=a.residual,
                          launch_mode=(("HIP graph replay of forward + backward, eager clip + fused AdamW (IVTrainEngine)" if a.graph else
                                        "eager forward + backward + clip + fused AdamW (IVTrainEngine)") if a.engine else
                                       ("HIP graph replay of forward + backward (no fused optimizer)" if a.graph else
                                        "eager autograd (no HIP graph, no fused optimizer)")),
                          text_tower_dropout=a.dropout, cpu_baseline=cpu,
                          optimizer=("fused AdamW lr 5e-5 betas (0.9, 0.98) wd 0.05, max_grad_norm 3.0 (stage2/1B/config.py:97-102)" if a.engine else None), batched_text_passes=bool(a.batch_text), grouped_text_weight_grads=bool(a.group_wgrad),
                          peak_mem_gb=round(torch.cuda.max_memory_allocated() / 1e9, 1),
                          mfma_flop_per_step=round((tot_fl + att_fl) / 1e12, 2), mfma_frac_of_step=round((tot_fl + att_fl) / (ms * 1e-3) / 2.5e15, 4),
                          higher_is_better=True, n_gpus=1, steps=a.steps, warmup=a.warmup, scaling="weak", vs_baseline=None,
                          config=dict(workload="InternVideo2 stage-2 1B step (multi_modality/scripts/pretraining/stage2/1B/config.py: 4x224^2, mask 0.8, "
                                               "max_txt_l 32; all four losses), forward + backward", per_gpu_batch=B, parallelism="dp1"),
                          roofline=roofline)), flush=True)


if __name__ == "__main__":
    main()
