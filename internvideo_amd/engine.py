"""Native data-parallel training engine for the InternVideo2 student -- and, through the same flat buffers, for the stage-2 video-text
model (`InternVideo2_Stage2_visual`: vision tower + BERT text / fusion tower + heads) -- one process per MI355X, RCCL over xGMI.

Replaces what DeepSpeed 0.10.1 does for the reference recipe (InternVideo2/single_modality/run_pretraining.py:363-375,
utils.py:814-908: bf16 engine, FusedAdam adam_w_mode, gradient clipping 3.0, ZeRO-1 bucketed gradient reduction) with a
design sized for 288 GB of HBM per GPU:

  * every parameter lives in ONE flat fp32 master buffer; Linear/Conv matrices additionally have a flat bf16 compute
    copy (what the MFMA kernels read) and a flat bf16 gradient buffer that the wgrad GEMMs write directly
    (`param.main_grad`); vectors / positional tables keep fp32 gradients.  Nothing is re-cast or copied per step.
  * the flat gradient buffers are laid out in BACKWARD order (heads, block depth-1 ... block 0, patch embed), so the
    gradients finished so far always form a contiguous prefix: buckets are plain slices, reduced in place by RCCL on a
    side HIP stream while the remaining blocks' backward runs (hook from BlockStackFn after every block).  The bucket plan
    is fixed at construction (same on every rank, every step).
  * three reductions of the matrix gradients over the ranks (`reduce_mode` / `reduce_dtype`):
      "allreduce" + "fp32"  (default) the bucket is widened to an fp32 communication buffer first: exact fp32 accumulation of the ranks' bf16
                            gradients -- what DeepSpeed's bf16 engine does for the reference recipe -- at twice the bytes of the bf16 form
                            (2 (W-1)/W x 4.28 GB per step and GPU: ~7 ms over 7 xGMI links, hidden behind backward except for the last bucket).
      "allreduce" + "bf16"  all-reduce of the bf16 buckets in place (DDP-equivalent, run_pretraining.py:378).  Least traffic per
                            collective call (ring: 2 (W-1)/W x 2.14 GB), but the W-way sum is rounded to bf16 inside RCCL (measured on two
                            gloo ranks: 1e-4 ... 6e-3 relative per bucket, tests/test_host_logic.py); opt-in.
      "zero1"               the ZeRO-1 role of the reference recipe (utils.py:863-871, scripts/pretraining/1B_pt.sh:65) mapped onto
                            xGMI's point-to-point mesh: ONE all-to-all of bf16 gradient shards per bucket (every pair of GPUs
                            exchanges its 1/W slice directly over its own link: (W-1)/W x 2.14 GB out per GPU, spread over 7
                            links, no ring), the W received pieces are accumulated in fp32 by `ivh_shard_sum_bf16` (exact fp32
                            sum, bf16 on the wire), AdamW updates only this rank's 1/W of master / moments / bf16 copy, and one
                            all-gather per bucket redistributes the bf16 compute copy.  AdamW streams 28 B x params / W.
  * the weight-decay split follows optim_factory.get_parameter_groups (:56-98): 1-D params, *.bias and the
    no_weight_decay() names are not decayed -- which is exactly the fp32 "vector" region -- so one fused AdamW launch
    per region (per bucket shard in zero1 mode) updates master, moments and the bf16 copy.
  * global grad-norm clipping (utils.py:860-861) is computed on the device (deterministic two-stage reduction; in zero1 mode
    the shard norms meet in one scalar all-reduce) and consumed by the AdamW kernel through a device scalar: the step
    never synchronises with the host.
"""
from __future__ import annotations

import contextlib
import gc
import weakref
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import ops

BF16, F32 = torch.bfloat16, torch.float32


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


def _unregister_dropout_epoch(t):
    """drop the library's pointer to a dead engine's dropout epoch (only if it is still the registered one)"""
    try:
        from . import xbert
        if xbert._DROP_EPOCH[0] is t:
            xbert.set_dropout_epoch(None)
    except Exception:      # noqa: BLE001  (interpreter shutdown)
        pass


@contextlib.contextmanager
def _cyclic_gc_paused():
    """No cyclic garbage collection while a stream is capturing.  A collection that the ~10^4 Python allocations of a captured step trigger
    runs finalizers of whatever garbage it finds ON THE CAPTURING THREAD; one that makes a runtime call which is illegal during capture
    (releasing a graph's memory pool, querying an event) throws out of a destructor and takes the process down -- seen once in round 5:
    `Fatal Python error: Aborted ... Garbage-collecting ... _attach_split_ws ... capture_step`, in a suite that had passed with the same kernels
    an hour earlier; whether it happens is a matter of allocation counts.  The usual victim is a dead engine of an earlier test: engine and
    model reference each other (the per-block hook), so a dropped engine -- its CUDAGraph and the graph's private pool with it -- waits for the
    cyclic collector, and torch.cuda.graph no longer collects on entry (torch 2.10: only with torch.compiler.config.force_cudagraph_gc).
    Here what is already garbage is collected before the capture begins and what becomes garbage during it right after; reference counting
    frees tensors during it as before."""
    was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class IVTrainEngine:
    def __init__(self, model, lr: float = 1.5e-4, betas=(0.9, 0.98), eps: float = 1e-6, weight_decay: float = 0.05,
                 max_grad_norm: float = 3.0, process_group=None, bucket_bytes: int = 96 << 20, overlap: bool = True,
                 clip_loss_ratio=(1.0, 1.0), mae_loss_ratio: float = 1.0, wgrad_stream: bool = False,
                 force_comm: bool = False, reduce_mode: str = "allreduce", reduce_dtype: str = "fp32",
                 check_finite: bool = False, lr_scales=None, layer_decay: Optional[float] = None, dropout_epoch: Optional[bool] = None):
        """lr_scales: callable(parameter name) -> lr_scale, the per-group factor of the reference's layer-wise lr decay
        (optim_factory.get_parameter_groups `lr_scale`); layer_decay: shorthand that builds it the way run_finetuning.py:548-549 does
        (values layer_decay ** (depth + 1 - i), layer ids of optim_factory.get_num_layer_for_vit)."""
        if reduce_mode not in ("allreduce", "zero1") or reduce_dtype not in ("bf16", "fp32"):
            raise ValueError("reduce_mode must be 'allreduce' or 'zero1', reduce_dtype 'bf16' or 'fp32'")
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.clip_loss_ratio, self.mae_loss_ratio = clip_loss_ratio, mae_loss_ratio
        self.pg = process_group
        inited = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if inited else 1
        self.rank = dist.get_rank(process_group) if inited else 0
        # force_comm: run the bucketed RCCL reduction (side stream, per-block hooks) even on a 1-rank group -- how the multi-GPU
        # code path is exercised on a single-GPU box (tests, `bench.py --force-dist`)
        self.comm = self.world > 1 or (force_comm and inited)
        self.overlap = overlap and self.comm
        self.bucket_bytes = bucket_bytes
        self.reduce_mode, self.reduce_dtype = reduce_mode, reduce_dtype
        self.zero1 = self.comm and reduce_mode == "zero1"
        self.step_count = 0
        self._consolidated_at = 0                              # a fresh engine holds complete state on every rank
        # the reference's per-step guard (engines/engine_for_pretraining.py:151-161): all-gather the loss over the ranks and stop the
        # job when any rank sees NaN / Inf.  It needs the loss on the host (one sync per step), so it is opt-in; off = no host sync.
        self.check_finite = bool(check_finite)
        self.all_loss_mean: Optional[float] = None
        dev = next(model.parameters()).device
        self.device = dev
        skip = set(model.no_weight_decay()) if hasattr(model, "no_weight_decay") else set()

        # ---- which module owns the block stack ---------------------------------------------------------------------------
        # the student itself, or the vision tower of the stage-2 model (multi_modality/models/internvideo2_stage2_visual.py:17-120): there
        # every parameter outside the vision tower's block stack / patch embedding (text + fusion tower, projection heads, temperature, the
        # vision decoders) belongs to autograd nodes that run BEFORE the tower's single backward node and may be used by several of them
        # (tied word embeddings, several passes through the same layers): those gradients are ACCUMULATED into buffers zeroed per step
        # (`p._ivh_accum`), they are final when the tower's backward begins, and their buckets are the first to be reduced.
        if hasattr(model, "blocks"):
            self.tower, self.tower_prefix = model, ""
        elif hasattr(getattr(model, "vision_encoder", None), "blocks"):
            self.tower, self.tower_prefix = model.vision_encoder, "vision_encoder."
        else:
            raise ValueError("IVTrainEngine: the model (or its .vision_encoder) must own a `blocks` stack")
        self.accumulate_outside_tower = bool(self.tower_prefix)
        # the ~150 Linear layers of the text / fusion tower are separate autograd nodes with 16-64 tiles each: their weight gradients are
        # queued and launched as grouped GEMMs (functional.grouped_weight_grads) -- resolved into the flat buffers when the vision tower's
        # backward begins, i.e. before the first bucket is reduced, so grouping and data parallelism coexist
        self.group_text_wgrads = self.accumulate_outside_tower
        tp = self.tower_prefix

        # ---- ordering: backward order -------------------------------------------------------------------------------
        # frozen parameters (requires_grad False) stay outside the engine: no gradient buffer, no optimizer state, never decayed
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
        depth = len(self.tower.blocks)

        def order_key(item):
            name = item[0]
            if name.startswith(tp + "blocks."):
                return (1, depth - 1 - int(name[len(tp):].split(".")[1]))
            if name.startswith(tp + "patch_embed") or name in (tp + "cls_token", tp + "pos_embed"):
                return (2, 0)
            return (0, 0)                                     # decoders / projector / text tower: their grads are ready first

        named.sort(key=order_key)
        mats, vecs = [], []
        for name, p in named:
            decay = not (p.dim() == 1 or name.endswith(".bias") or name in skip)
            (mats if decay else vecs).append((name, p))
        self.mat_params, self.vec_params = mats, vecs
        if lr_scales is None and layer_decay is not None and layer_decay < 1.0:
            from .schedules import LayerDecayValueAssigner
            assigner = LayerDecayValueAssigner.for_depth(depth, layer_decay)
            # the layer ids are those of the TOWER's parameter names (optim_factory.get_num_layer_for_vit matches "blocks." / "patch_embed" at
            # the start of the name): strip the tower's prefix (stage-2 `vision_encoder.`), or nothing would match and every layer would
            # silently train at scale 1; parameters outside the tower (text tower, heads) are the last layer: scale 1
            lr_scales = lambda name: assigner.get_scale(assigner.get_layer_id(name[len(tp):] if (tp and name.startswith(tp)) else name))   # noqa: E731
        self.lr_scales = lr_scales

        def layout(items, total_align):
            offs, n = [], 0
            for _, p in items:
                offs.append(n)
                n += _align(p.numel())
            return offs, _align(n, total_align)

        # zero1: every bucket (hence the whole matrix region) splits into `world` shards of whole 64-element groups
        self.shard_quantum = 64 * self.world if self.zero1 else 64
        self.mat_off, n_mat = layout(mats, 1024 * (self.world if self.zero1 else 1))
        self.vec_off, n_vec = layout(vecs, 1024)
        self.n_mat, self.n_vec = n_mat, n_vec
        self.master = torch.zeros(n_mat + n_vec, dtype=F32, device=dev)
        self.exp_avg = torch.zeros_like(self.master)
        self.exp_avg_sq = torch.zeros_like(self.master)
        self.shadow = torch.zeros(n_mat, dtype=BF16, device=dev)
        self.grad_mat = torch.zeros(n_mat, dtype=BF16, device=dev)
        self.grad_vec = torch.zeros(n_vec, dtype=F32, device=dev)
        def outside(name):
            return self.accumulate_outside_tower and order_key((name, None)) == (0, 0)
        for (name, p), off in zip(mats, self.mat_off):
            n = p.numel()
            self.master[off:off + n].copy_(p.detach().reshape(-1).float())
            p.data = self.master[off:off + n].view(p.shape)
            p._ivh_bf16 = self.shadow[off:off + n].view(p.shape[0], -1) if p.dim() >= 2 else self.shadow[off:off + n]
            p.main_grad = self.grad_mat[off:off + n].view(p.shape)
            p._ivh_accum = outside(name)
        for (name, p), off in zip(vecs, self.vec_off):
            n = p.numel()
            o = n_mat + off
            self.master[o:o + n].copy_(p.detach().reshape(-1).float())
            p.data = self.master[o:o + n].view(p.shape)
            p.main_grad = self.grad_vec[off:off + n].view(p.shape)
            p._ivh_accum = outside(name)
        self.shadow.copy_(self.master[:n_mat])                # initial bf16 compute copy
        # layer-wise lr decay: one (end offset, scale) table per region, runs of equal scale merged (a block's parameters are adjacent in
        # backward order, so the 1B classifier has 42 segments per region); consumed by ivh_adamw_step_scaled
        self._lr_seg_mat = self._lr_segments(mats, self.mat_off, n_mat) if lr_scales is not None else None
        self._lr_seg_vec = self._lr_segments(vecs, self.vec_off, n_vec) if lr_scales is not None else None
        # A gradient may reach ANY managed parameter through plain autograd (`.grad`) instead of a kernel that fills main_grad: the separable
        # positional tables and the frame-averaged image tables (composed with torch ops: the tower's own `pos_embed` / `clip_pos_embed` in
        # image steps of a model without sep_image_video_pos_embed -- ADVICE r4), everything outside the tower of the stage-2 model (e.g. the
        # temperature).  Which leaves are reached that way depends on the step (image / video), so every managed parameter is looked at.
        self._autograd_params = list(mats + vecs)
        # The GEMMs read `shadow`, the optimizer writes it.  Anything ELSE that writes parameters (model.load_state_dict after the engine
        # was built -- the reference's resume order, utils.py:568-647 -- or an in-place edit of p.data) changes `master` only: refresh
        # the copy from a load_state_dict post-hook, and let callers that edit parameters by hand call sync_shadow() themselves.
        # (The hooks hold the engine WEAKLY: a strong reference would close a model -> hook -> engine -> model cycle, and a dropped engine --
        # its flat buffers, its HIP graph and the graph's private pool -- would wait for the cyclic collector instead of going with its last
        # reference; see _cyclic_gc_paused for what a collection at the wrong moment costs.)
        me = weakref.ref(self)
        if hasattr(model, "register_load_state_dict_post_hook"):
            def _resync(module, incompatible):
                eng = me()
                if eng is not None:
                    eng.sync_shadow()
            self._lsd_hook = model.register_load_state_dict_post_hook(_resync)
        # zero1: p.data are views of the fp32 master buffer, whose non-owned shards go stale with the first step -- a model-level
        # checkpoint / eval must not silently mix updated and initial weights
        if self.zero1 and self.world > 1 and hasattr(model, "register_state_dict_pre_hook"):
            def _need_consolidated(module, prefix, keep_vars):
                eng = me()
                if eng is not None and not eng.consolidated:
                    raise RuntimeError("model.state_dict() under IVTrainEngine(reduce_mode='zero1'): the fp32 weights are sharded over the "
                                       "ranks; call engine.consolidate() on EVERY rank first")
            self._sd_hook = model.register_state_dict_pre_hook(_need_consolidated)
        # block index -> end offset (exclusive) of its matrices in grad_mat (prefix finished once that block's backward ran)
        self.block_end: Dict[int, int] = {}
        for (name, p), off in zip(mats, self.mat_off):
            if name.startswith(tp + "blocks."):
                i = int(name[len(tp):].split(".")[1])
                self.block_end[i] = max(self.block_end.get(i, 0), off + _align(p.numel()))
        self.head_end = min((off for (name, _), off in zip(mats, self.mat_off) if name.startswith(tp + "blocks.")), default=0)
        # ---- bucket plan: (lo, hi) slices of the matrix region in reduction order; the first `bucket_trigger[i]` buckets may be
        # launched as soon as block i has finished its backward (the rest, up to n_mat, at the end of backward)
        self.buckets: List[Tuple[int, int]] = []
        self.bucket_trigger: Dict[int, int] = {}
        lo, q = 0, self.shard_quantum
        for i in range(depth - 1, -1, -1):
            hi = lo + (self.block_end.get(i, lo) - lo) // q * q
            if (hi - lo) * 2 >= bucket_bytes:
                self.buckets.append((lo, hi))
                lo = hi
            self.bucket_trigger[i] = len(self.buckets)
        if lo < n_mat:
            self.buckets.append((lo, n_mat))
        self._next_bucket = 0
        # once a bucket has gone to the wire, a later write into it is silently lost on every other rank.  The parameters of a
        # bucket are marked `_ivh_closed` when its reduction is launched (multi-rank engines only) and every delivery path of a gradient
        # (functional._ret_grad / _wgrad_defer / _end_of_backward, _fold_autograd_grads) refuses to touch a closed parameter.
        self._bucket_params: Dict[int, list] = {}
        for (name, p), off in zip(mats, self.mat_off):
            for lo_, hi_ in self.buckets:
                if lo_ <= off < hi_:
                    self._bucket_params.setdefault(lo_, []).append(p)
        self._closed: list = []
        self._sumsq = torch.zeros(1, dtype=F32, device=dev)
        self._sq_scratch = torch.empty(4096, dtype=F32, device=dev)
        self._clip = None
        self.grad_norm = torch.zeros(1, dtype=F32, device=dev)
        self.reduce_log: List[Tuple[int, int]] = []
        self._defer_reduce = False
        self._seg_capture = None                               # live state of a segmented capture (capture_step(segmented=True))
        self._segments = None                                  # [(graph, [(lo, hi)], reduce the vector region too)] in replay order
        self.comm_stream = torch.cuda.Stream(device=dev) if (self.comm and dev.type == "cuda") else None
        # communication buffers
        self.grad_comm32 = torch.zeros(n_mat, dtype=F32, device=dev) if (self.comm and not self.zero1 and reduce_dtype == "fp32") else None
        self.a2a_recv = torch.zeros(n_mat, dtype=BF16, device=dev) if self.zero1 else None
        self.grad_shard32 = torch.zeros(n_mat // self.world, dtype=F32, device=dev) if self.zero1 else None
        # optional: weight-gradient GEMMs on their own stream, filling the CUs the dgrad chain leaves idle (functional._wgrad).
        # Off by default since the four wgrads of a block go out as one grouped launch that fills the GPU by itself
        # (measured on the 1B step: 140.1 ms without the stream, 142.6 ms with it; before grouping it was worth 20 ms).
        self.wgrad_stream = torch.cuda.Stream(device=dev) if (wgrad_stream and dev.type == "cuda") else None
        self.tower.grad_ready_hook = self._block_hook() if self.overlap else None
        # device-side dropout epoch (include/internvideo_hip.h ivh_set_dropout_epoch): on by default for a model with a text tower (BERT's
        # dropout 0.1, config_bert_large.json:5,8), so that its graph-captured step draws fresh masks on every replay
        self.dropout_epoch = None
        if (self.accumulate_outside_tower if dropout_epoch is None else dropout_epoch) and dev.type == "cuda":
            from . import xbert
            self.dropout_epoch = torch.zeros(1, dtype=torch.int32, device=dev)
            xbert.set_dropout_epoch(self.dropout_epoch)
            weakref.finalize(self, _unregister_dropout_epoch, self.dropout_epoch)   # an engine that goes away takes its registration with it

    def _lr_segments(self, items, offs, total):
        ends, scales = [], []
        for k, (name, _) in enumerate(items):
            sc = float(self.lr_scales(name))
            end = offs[k + 1] if k + 1 < len(items) else total
            if scales and scales[-1] == sc:
                ends[-1] = end
            else:
                ends.append(end); scales.append(sc)
        if not ends:
            return None
        if len(ends) > 1024:
            raise ValueError(f"layer-wise lr decay: {len(ends)} segments, the kernel's table holds 1024")
        return (torch.tensor(ends, dtype=torch.int64, device=self.device), torch.tensor(scales, dtype=F32, device=self.device))

    def lr_scale_of(self, name: str) -> float:
        """the lr_scale the engine applies to parameter `name` (1.0 without layer-wise decay)"""
        return 1.0 if self.lr_scales is None else float(self.lr_scales(name))

    def sync_shadow(self):
        """re-derive the bf16 compute copy of every matrix from the fp32 master buffer (after parameters were written by anything
        other than optimizer_step: checkpoint loads into the model, manual edits)"""
        self.shadow.copy_(self.master[:self.n_mat])

    # ---- gradient reduction -------------------------------------------------------------------------------------------
    def _shard(self, lo: int, hi: int) -> Tuple[int, int]:
        """this rank's slice (start, length) of bucket [lo, hi) (zero1)"""
        c = (hi - lo) // self.world
        return lo + self.rank * c, c

    def _reduce_bucket(self, lo: int, hi: int):
        """the collective(s) of one bucket, on the current stream"""
        g = self.grad_mat[lo:hi]
        if self.zero1:
            recv = self.a2a_recv[lo:hi]
            dist.all_to_all_single(recv, g, group=self.pg)     # piece r of every rank's bucket lands on rank r
            out = self.grad_shard32[lo // self.world:hi // self.world]
            if recv.is_cuda:
                ops.shard_sum_bf16(recv, self.world, out)
            else:                                              # host tensors only occur in the gloo tests of the reduction logic
                torch.sum(recv.view(self.world, -1), dim=0, dtype=F32, out=out)
        elif self.grad_comm32 is not None:
            c32 = self.grad_comm32[lo:hi]
            c32.copy_(g)                                       # widen, then an exact fp32 all-reduce
            dist.all_reduce(c32, group=self.pg)
        else:
            dist.all_reduce(g, group=self.pg)

    def _close_bucket(self, lo: int):
        """from here on nothing may write the gradients of the bucket that starts at `lo` (see __init__)"""
        for p in self._bucket_params.get(lo, ()):
            p._ivh_closed = True
            self._closed.append(p)

    def _launch_reduce(self, lo: int, hi: int):
        if hi <= lo:
            return
        self._close_bucket(lo)
        if self._seg_capture is not None:                      # segmented capture: the graph is cut here, the collective stays eager
            if self.wgrad_stream is not None:                  # the bucket's matrices are written on the wgrad stream: join before the cut
                torch.cuda.current_stream().wait_stream(self.wgrad_stream)
            self._seg_cut([(lo, hi)], vec=False)
            self.reduce_log.append((lo, hi))
            return
        if self.comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.comm_stream.wait_event(ev)
            if self.wgrad_stream is not None:                  # the bucket's matrices are written on the wgrad stream
                ev2 = torch.cuda.Event()
                ev2.record(self.wgrad_stream)
                self.comm_stream.wait_event(ev2)
            with torch.cuda.stream(self.comm_stream):
                self._reduce_bucket(lo, hi)
        else:                                                  # host tensors (gloo): same bucketing, no stream
            self._reduce_bucket(lo, hi)
        self.reduce_log.append((lo, hi))

    def _block_hook(self):
        """the per-block backward hook handed to the tower, holding the engine weakly (no model -> engine cycle)"""
        me = weakref.ref(self)

        def hook(*a, **kw):
            eng = me()
            return eng._on_block_done(*a, **kw) if eng is not None else None
        return hook

    def _on_block_done(self, i: int):
        """called by BlockStackFn.backward after block i: gradients of blocks >= i (and the heads) are final."""
        upto = self.bucket_trigger.get(i, 0)
        if self._next_bucket == 0 and upto > 0 and self.accumulate_outside_tower:
            self._fold_autograd_grads()                        # stage 2: the first buckets hold what autograd may have delivered as .grad
        while self._next_bucket < upto:
            self._launch_reduce(*self.buckets[self._next_bucket])
            self._next_bucket += 1

    def _reduce_vec(self):
        dist.all_reduce(self.grad_vec, group=self.pg)

    def _finish_reduce(self):
        from . import functional as Fn
        Fn._wgrad_flush(force=True)                            # weight gradients still queued for a grouped launch
        if self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)
        if not self.comm or self._defer_reduce:
            return
        while self._next_bucket < len(self.buckets):
            self._launch_reduce(*self.buckets[self._next_bucket])
            self._next_bucket += 1
        if self._seg_capture is not None:                      # the vector region's all-reduce ends the last segment
            self._seg_cut([], vec=True, last=True)
            return
        if self.comm_stream is not None:
            with torch.cuda.stream(self.comm_stream):
                self._reduce_vec()
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            self._reduce_vec()

    def reduce_all_now(self):
        """the whole gradient reduction on the current stream, bucket by bucket, after backward has finished (no overlap): the
        companion of a graph-captured step on a multi-rank group (capture_step(defer_reduce=True))."""
        if not self.comm:
            return
        for lo, hi in self.buckets:
            self._reduce_bucket(lo, hi)
        self._reduce_vec()

    # ---- gradient accumulation (run_pretraining.py:42,375 `--update_freq` = DeepSpeed's gradient_accumulation_steps) ------------------------
    def accumulate(self):
        """add the gradients of the micro-step that just ran its backward (flat bf16 matrix buffer, fp32 vector buffer: both are rewritten by
        every backward) to fp32 accumulators; nothing is reduced over the ranks until the boundary (optimizer_step(accumulated=True))."""
        from . import functional as Fn
        if self.zero1:
            raise NotImplementedError("IVTrainEngine: gradient accumulation with reduce_mode='zero1' is not built")
        Fn._wgrad_flush(force=True)                            # weight gradients still queued for a grouped launch
        if self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)
        if getattr(self, "_acc_mat", None) is None:
            self._acc_mat = torch.zeros(self.n_mat, dtype=F32, device=self.device)
            self._acc_vec = torch.zeros(self.n_vec, dtype=F32, device=self.device)
            self._acc_n = 0
        self._acc_mat.add_(self.grad_mat)
        self._acc_vec.add_(self.grad_vec)
        self._acc_n += 1

    # ---- optimizer ------------------------------------------------------------------------------------------------------
    def optimizer_step(self, lr: Optional[float] = None, weight_decay: Optional[float] = None, accumulated: bool = False):
        """accumulated=True: the step of a gradient-accumulation boundary -- the fp32 sums of accumulate() are all-reduced over the ranks
        (one exact fp32 sum per region, no overlap) and take the place of this micro-step's buffers; they are zeroed afterwards."""
        lr = self.lr if lr is None else lr
        wd = self.weight_decay if weight_decay is None else weight_decay
        self.step_count += 1
        gs = 1.0 / self.world
        clip = None
        n_mat, W = self.n_mat, self.world
        mat_grad = self.grad_comm32 if self.grad_comm32 is not None else self.grad_mat
        vec_grad = self.grad_vec
        if accumulated:
            if getattr(self, "_acc_mat", None) is None or self._acc_n == 0:
                raise RuntimeError("IVTrainEngine.optimizer_step(accumulated=True) without a preceding accumulate()")
            if self.comm:
                dist.all_reduce(self._acc_mat, group=self.pg)
                dist.all_reduce(self._acc_vec, group=self.pg)
            mat_grad, vec_grad = self._acc_mat, self._acc_vec
        if self.max_grad_norm and self.max_grad_norm > 0:
            if self.zero1:                                     # shard norms meet in one scalar all-reduce; the vector region is replicated
                ops.sqnorm(self.grad_shard32, self._sumsq, False, self._sq_scratch)
                dist.all_reduce(self._sumsq, group=self.pg)
            else:
                ops.sqnorm(mat_grad, self._sumsq, False, self._sq_scratch)
            ops.sqnorm(vec_grad, self._sumsq, True, self._sq_scratch)
            clip, nrm = ops.clip_coef(self._sumsq, self.max_grad_norm * W)
            self.grad_norm = nrm / W
        b1, b2 = self.betas
        if self.zero1:
            for lo, hi in self.buckets:
                s0, c = self._shard(lo, hi)
                ops.adamw_step(self.master[s0:s0 + c], self.exp_avg[s0:s0 + c], self.exp_avg_sq[s0:s0 + c],
                               self.grad_shard32[lo // W:lo // W + c], self.shadow[s0:s0 + c], lr, b1, b2, self.eps, wd, self.step_count, gs, clip,
                               lr_segments=self._lr_seg_mat, seg_base=s0)
            # redistribute the bf16 compute copy: one in-place all-gather per bucket, on the communication stream, under the
            # (replicated) AdamW of the vector region; the next forward waits for it
            if self.comm_stream is not None:
                self.comm_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.comm_stream):
                    self._gather_buckets(self.shadow)
            else:
                self._gather_buckets(self.shadow)
        else:
            ops.adamw_step(self.master[:n_mat], self.exp_avg[:n_mat], self.exp_avg_sq[:n_mat], mat_grad, self.shadow,
                           lr, b1, b2, self.eps, wd, self.step_count, gs, clip, lr_segments=self._lr_seg_mat)
        ops.adamw_step(self.master[n_mat:], self.exp_avg[n_mat:], self.exp_avg_sq[n_mat:], vec_grad, None,
                       lr, b1, b2, self.eps, 0.0, self.step_count, gs, clip, lr_segments=self._lr_seg_vec)
        if accumulated:
            self._acc_mat.zero_(); self._acc_vec.zero_()
            self._acc_n = 0
        if self.zero1 and self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        from . import functional as Fn
        Fn.WEIGHT_EPOCH += 1                                   # cached fp8 copies of the weights are stale now

    def _gather_buckets(self, buf: torch.Tensor):
        """all-gather every bucket of a flat matrix-region buffer from the ranks' shards, in place"""
        for lo, hi in self.buckets:
            s0, c = self._shard(lo, hi)
            dist.all_gather_into_tensor(buf[lo:hi], buf[s0:s0 + c], group=self.pg)

    def consolidate(self):
        """zero1: fp32 master and moments of the matrix region are only current on the rank that owns the shard; gather them
        (checkpointing, or before reading parameters through the model).  COLLECTIVE: every rank must call it (3 x #buckets
        all-gathers) -- call it on all ranks, then let rank 0 alone call state_dict() / model.state_dict() (the reference's
        save_on_master pattern, utils.py:568-647).  No-op otherwise."""
        if self.zero1 and self.world > 1:
            for buf in (self.master, self.exp_avg, self.exp_avg_sq):
                self._gather_buckets(buf[:self.n_mat])
        self._consolidated_at = self.step_count

    @property
    def consolidated(self) -> bool:
        """True when every rank holds current fp32 master weights / moments (always, except zero1 after a step without consolidate())"""
        return not (self.zero1 and self.world > 1) or getattr(self, "_consolidated_at", -1) == self.step_count

    # ---- one training step ------------------------------------------------------------------------------------------------
    def backward(self, loss: torch.Tensor):
        """loss.backward() with the weight-gradient GEMMs routed to the engine's wgrad stream."""
        from . import functional as Fn
        if self.wgrad_stream is not None:
            self.wgrad_stream.wait_stream(torch.cuda.current_stream())   # last step's optimizer read the gradient buffers
        import contextlib
        prev, Fn.WGRAD_STREAM = Fn.WGRAD_STREAM, self.wgrad_stream
        try:
            with (Fn.grouped_weight_grads() if self.group_text_wgrads else contextlib.nullcontext()):
                loss.backward()
        finally:
            Fn.WGRAD_STREAM = prev
        self._fold_autograd_grads()

    def _fold_autograd_grads(self):
        """parameters whose gradient reaches them through plain autograd instead of a kernel that writes main_grad (the separable positional
        tables of sep_pos_embed: the joint table is composed with torch ops; the stage-2 temperature): fold .grad into the engine's buffers"""
        for _, p in self._autograd_params:
            if p.grad is not None:
                if getattr(p, "_ivh_closed", False):          # its bucket went to the wire already: the contribution would be lost on N > 1 ranks
                    raise RuntimeError("IVTrainEngine: a gradient reached a parameter after its bucket was reduced")
                p.main_grad.add_(p.grad.reshape(p.main_grad.shape).to(p.main_grad.dtype))
                p.grad = None

    # ---- HIP-graph mode ---------------------------------------------------------------------------------------------------
    def capture_step(self, video: torch.Tensor, mask: torch.Tensor, targets, L: Optional[int] = None, warmup: int = 2,
                     defer_reduce: bool = False, capture_comm: bool = False, segmented: bool = False):
        """Capture mask -> indices + forward + fused loss + backward (both streams) of one step into a HIP graph.  The ~2200 kernel
        launches of a step cost ~120 ms of Python / ctypes / allocator time when issued one by one -- as long as the GPU work
        itself; replayed from a graph they cost the GPU front-end ~1 us each.  `video`, `mask` and `targets` become the graph's
        static inputs: write the next batch INTO them (copy_) before each `train_step_graphed()`.  The optimizer launches stay
        outside the graph (their step / lr arguments change every step).  On a multi-rank group:
          capture_comm=True   the bucketed collectives are captured WITH the step, on the communication stream, forked / joined by
                              events exactly as in eager mode: overlap with backward is kept and the host enqueues nothing per step;
          segmented=True      the step is captured as a CHAIN of HIP graphs cut at every point where a bucket becomes final (the
                              per-block hook): replay = graph 0, eager all-reduce of bucket 0 on the communication stream, graph 1, ...
                              The collectives are ordinary RCCL calls (nothing exotic is asked of the runtime), they overlap with the
                              following segments' backward exactly as in eager mode, and the host enqueues ~2 x #buckets calls per step
                              instead of ~2200 launches.  The default multi-rank mode of bench.py;
          defer_reduce=True   no collective is captured; the buckets are reduced after each replay, without overlap (fallback for
                              an RCCL / runtime combination that cannot capture collectives)."""
        from .internvideo2_pretrain import build_gather_indices

        def loss_fn():
            vis_inv = build_gather_indices(mask, self.device, L=L, check=False) if mask.is_cuda else None
            return self.model.forward_loss(video, mask, targets, self.clip_loss_ratio, self.mae_loss_ratio, vis_inv=vis_inv)
        return self.capture_fn(loss_fn, warmup=warmup, defer_reduce=defer_reduce, capture_comm=capture_comm, segmented=segmented)

    def capture_fn(self, loss_fn, warmup: int = 2, defer_reduce: bool = False, capture_comm: bool = False, segmented: bool = False):
        """capture_step for any model the engine manages: `loss_fn()` -> loss (device scalar) or (loss, aux); everything it reads must be
        static tensors (refresh them with copy_ between replays).  Same multi-rank modes as capture_step."""
        if self.comm and not (defer_reduce or capture_comm or segmented):
            raise RuntimeError("capture_step on a multi-rank group needs segmented=True (a chain of graphs with eager collectives between "
                               "them, overlapped), capture_comm=True (collectives inside the graph, overlapped) or defer_reduce=True "
                               "(graph replay, then the bucketed reduction without overlap)")
        from . import functional as Fn
        segmented = bool(self.comm and segmented and not capture_comm)
        self._segments = None
        self._defer_reduce = bool(self.comm and defer_reduce and not capture_comm and not segmented)
        self.tower.grad_ready_hook = self._block_hook() if (self.overlap and not self._defer_reduce) else None

        def body():
            self.zero_grad()
            self._begin_step_on_device()
            out = loss_fn()
            loss, parts = (out[0], out[1]) if isinstance(out, tuple) else (out, None)
            self.backward(loss)
            self._finish_reduce()
            return loss.detach(), parts

        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                         # eager warm-up on a side stream, as torch.cuda.graph wants it
            for _ in range(warmup):
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if segmented:
            return self._capture_segmented(body, side)
        self._graph = torch.cuda.CUDAGraph()
        Fn.WGRAD_KEEPALIVE = []
        Fn.FP8_CAPTURE_CACHE = {}                             # fp8 weights are quantised INSIDE the graph, once per captured step
        try:
            with _cyclic_gc_paused(), torch.cuda.graph(self._graph):
                self._graph_out = body()
        finally:
            keep, Fn.WGRAD_KEEPALIVE = Fn.WGRAD_KEEPALIVE, None
            Fn.FP8_CAPTURE_CACHE = None
        del keep                                              # their memory stays in the graph's private pool
        return self._graph_out

    # -- segmented capture: a chain of graphs, the collectives between them stay eager ----------------------------------------
    def _seg_begin(self):
        st = self._seg_capture
        g = torch.cuda.CUDAGraph()
        # "relaxed": the per-block hook runs on autograd's device thread, so a segment may end on another thread than it began on
        g.capture_begin(pool=st["pool"], capture_error_mode="relaxed")
        st["graph"] = g

    def _seg_cut(self, buckets, vec: bool, last: bool = False):
        st = self._seg_capture
        st["graph"].capture_end()
        st["plan"].append((st["graph"], list(buckets), vec))
        st["graph"] = None
        if not last:
            self._seg_begin()

    def _capture_segmented(self, body, stream):
        from . import functional as Fn
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        self._seg_capture = dict(pool=torch.cuda.graph_pool_handle(), plan=[], graph=None)
        Fn.WGRAD_KEEPALIVE = []
        Fn.FP8_CAPTURE_CACHE = {}
        stream.wait_stream(torch.cuda.current_stream())
        try:
            with _cyclic_gc_paused(), torch.cuda.stream(stream):
                self._seg_begin()
                out = body()                                   # _finish_reduce ends the last segment
                if self._seg_capture["graph"] is not None:     # (a body that never reached _finish_reduce)
                    self._seg_cut([], vec=False, last=True)
        except BaseException:
            g = self._seg_capture.get("graph")
            if g is not None:
                try:
                    g.capture_end()
                except Exception:      # noqa: BLE001
                    pass
            self._seg_capture = None
            raise
        finally:
            keep, Fn.WGRAD_KEEPALIVE = Fn.WGRAD_KEEPALIVE, None
            Fn.FP8_CAPTURE_CACHE = None
        del keep
        self._segments, self._seg_capture = self._seg_capture["plan"], None
        torch.cuda.current_stream().wait_stream(stream)
        self._graph, self._graph_out = None, out
        return out

    def _replay_segments(self):
        cur = torch.cuda.current_stream()
        self.reduce_log.clear()
        for g, buckets, vec in self._segments:
            g.replay()
            if not buckets and not vec:
                continue
            ev = torch.cuda.Event()
            ev.record(cur)
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                for lo, hi in buckets:
                    self._reduce_bucket(lo, hi)
                    self.reduce_log.append((lo, hi))
                if vec:
                    self._reduce_vec()
        cur.wait_stream(self.comm_stream)

    def _guard_finite(self, loss: torch.Tensor):
        """engines/engine_for_pretraining.py:151-161: every rank's loss is gathered; any NaN / Inf stops the job (the reference calls
        sys.exit(1); SystemExit(1) here, after the same message).  Also keeps the all-rank mean the reference logs."""
        v = loss.detach().reshape(1).float()
        if self.world > 1:
            parts = [torch.zeros_like(v) for _ in range(self.world)]
            dist.all_gather(parts, v, group=self.pg)
            v = torch.cat(parts)
        host = v.cpu()
        self.all_loss_mean = float(host.mean())
        isnan, isinf = bool(torch.isnan(host).any()), bool(torch.isinf(host).any())
        if isnan or isinf:
            print(" ========== loss_isnan = {},  loss_isinf = {} ========== ".format(isnan, isinf))
            raise SystemExit(1)

    def train_step_graphed(self, lr: Optional[float] = None, weight_decay: Optional[float] = None):
        """replay the captured step on the current contents of the static inputs, then AdamW.  -> (loss, parts) device scalars."""
        if self._segments is not None:
            self._replay_segments()
        else:
            self._graph.replay()
            if self._defer_reduce:
                self.reduce_all_now()
        if self.check_finite:                                 # before the update is applied, as the reference aborts before model.step()
            self._guard_finite(self._graph_out[0])
        self.optimizer_step(lr, weight_decay)
        return self._graph_out

    def zero_grad(self):
        """only the fp32 vector region accumulates (positional tables shared by several decoders); matrices are overwritten -- except, in
        the stage-2 model, the matrices outside the vision tower's block stack (accumulated: see __init__), which lie in [0, head_end)."""
        self.grad_vec.zero_()
        if self.accumulate_outside_tower and self.head_end > 0:
            self.grad_mat[:self.head_end].zero_()
        self._next_bucket = 0
        self.reduce_log.clear()
        for p in self._closed:
            p._ivh_closed = False
        self._closed.clear()

    def _begin_step_on_device(self):
        """per-step device-side state: the dropout epoch (csrc/common.h DropCfg: every mask is hash(call-site seed + epoch * K, element)),
        advanced by a kernel so that a captured step draws fresh masks on every replay"""
        ep = getattr(self, "dropout_epoch", None)
        if ep is not None:
            ep.add_(1)

    def train_step_fn(self, loss_fn, lr: Optional[float] = None, weight_decay: Optional[float] = None):
        """one training step of any model the engine manages (the stage-2 model: multi_modality/tasks/pretrain.py:207-213 `loss_dict =
        model(...); loss = sum(loss_dict.values()); model.backward(loss); model.step()`): `loss_fn()` -> loss or (loss, aux)."""
        self.zero_grad()
        self._begin_step_on_device()
        out = loss_fn()
        loss = out[0] if isinstance(out, tuple) else out
        if self.check_finite:
            self._guard_finite(loss)
        self.backward(loss)
        self._finish_reduce()
        if self._defer_reduce:
            self.reduce_all_now()
        self.optimizer_step(lr, weight_decay)
        return out

    def train_step(self, video: torch.Tensor, mask: torch.Tensor, targets, vis_inv=None, lr: Optional[float] = None,
                   weight_decay: Optional[float] = None):
        """forward + fused distillation loss + backward + gradient reduction + AdamW.  Returns the loss as a device
        scalar.  No host sync unless `check_finite` (the reference's per-step NaN / Inf guard, _guard_finite)."""
        self.zero_grad()
        self._begin_step_on_device()
        loss, parts = self.model.forward_loss(video, mask, targets, self.clip_loss_ratio, self.mae_loss_ratio, vis_inv=vis_inv)
        if self.check_finite:
            self._guard_finite(loss)
        self.backward(loss)
        self._finish_reduce()
        if self._defer_reduce:                                # an engine whose captured step defers the reduction reduces here too:
            self.reduce_all_now()                             # an eager step must never apply unreduced gradients scaled by 1 / world
        self.optimizer_step(lr, weight_decay)             # per-step lr / weight decay (engine_for_pretraining.py:56-61); None = the constructor's
        return loss.detach(), parts

    def close(self):
        """Call before `dist.destroy_process_group()`.  Releases what keeps the communicator's device resources referenced -- captured HIP
        graphs (a graph that captured RCCL kernels, capture_comm=True, holds them until it is destroyed), the segment chain, work still
        queued on the communication / weight-gradient streams -- in the order graphs -> streams -> (the caller's) communicator.  Destroying
        the group while a graph with captured collectives is alive is what aborted in ~1 of 10 runs of the round-3 1-rank tests; with this
        order the teardown probe (tools/rccl_teardown_probe.py) measures the abort rate per mode.  The engine stays usable in eager mode."""
        import gc
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        self._segments, self._seg_capture = None, None
        self._graph = None
        self._graph_out = None
        gc.collect()
        if self.device.type == "cuda":
            for st in (self.comm_stream, self.wgrad_stream):
                if st is not None:
                    st.synchronize()
            torch.cuda.synchronize(self.device)

    def layout_fingerprint(self) -> str:
        """digest of (parameter name, offset, size) of both flat regions: the flat master / moment buffers of a checkpoint mean nothing under
        another parameter order or alignment"""
        import hashlib
        h = hashlib.sha256()
        for items, offs in ((self.mat_params, self.mat_off), (self.vec_params, self.vec_off)):
            for (name, p), off in zip(items, offs):
                h.update(f"{name}:{off}:{p.numel()};".encode())
        h.update(f"|{self.n_mat}|{self.n_vec}".encode())
        return h.hexdigest()

    def state_dict(self):
        """pure read (no collective): safe to call on rank 0 only.  zero1 on several ranks: raises unless consolidate() ran on ALL ranks
        since the last optimizer step (a rank-0-only gather would deadlock the job; stale shards would silently mix old and new weights)."""
        if not self.consolidated:
            raise RuntimeError("IVTrainEngine(reduce_mode='zero1'): master weights / moments are sharded over the ranks; call "
                               "engine.consolidate() on EVERY rank before state_dict() / model.state_dict() (then rank 0 may save alone)")
        sd = {"master": self.master, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_count,
              "layout": self.layout_fingerprint()}
        if self.dropout_epoch is not None:
            sd["dropout_epoch"] = self.dropout_epoch
        return sd

    def load_state_dict(self, sd):
        if "layout" in sd and sd["layout"] != self.layout_fingerprint():
            raise RuntimeError("IVTrainEngine.load_state_dict: the checkpoint's flat buffers were laid out for another parameter order / set "
                               "(names, offsets or sizes differ); load the named `module` state_dict instead and start the moments afresh")
        if tuple(sd["master"].shape) != tuple(self.master.shape):
            raise RuntimeError(f"IVTrainEngine.load_state_dict: flat buffer of {tuple(sd['master'].shape)} elements, this engine holds {tuple(self.master.shape)}")
        self.master.copy_(sd["master"]); self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])
        if self.dropout_epoch is not None and "dropout_epoch" in sd:
            self.dropout_epoch.copy_(sd["dropout_epoch"])
        self._consolidated_at = self.step_count               # a loaded checkpoint is whole on every rank
        self.sync_shadow()
