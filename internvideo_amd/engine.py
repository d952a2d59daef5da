"""Native data-parallel training engine for the InternVideo2 student (one process per MI355X, RCCL over xGMI).

Replaces what DeepSpeed 0.10.1 does for the reference recipe (InternVideo2/single_modality/run_pretraining.py:363-375,
utils.py:814-908: bf16 engine, FusedAdam adam_w_mode, gradient clipping 3.0, bucketed gradient reduction) with a
design sized for 288 GB of HBM per GPU:

  * every parameter lives in ONE flat fp32 master buffer; Linear/Conv matrices additionally have a flat bf16 compute
    copy (what the MFMA kernels read) and a flat bf16 gradient buffer that the wgrad GEMMs write directly
    (`param.main_grad`); vectors / positional tables keep fp32 gradients.  Nothing is re-cast or copied per step.
  * the flat gradient buffers are laid out in BACKWARD order (heads, block depth-1 ... block 0, patch embed), so the
    gradients finished so far always form a contiguous prefix: buckets are plain slices, reduced in place by RCCL
    all-reduce on a side HIP stream while the remaining blocks' backward runs (hook from BlockStackFn after every block).
    xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU): few large buckets (default 256 MiB) keep RCCL's
    multi-ring/tree protocols at their bandwidth plateau; 2.14 GB of bf16 gradients hide under the ~100 ms backward.
  * the weight-decay split follows optim_factory.get_parameter_groups (:56-98): 1-D params, *.bias and the
    no_weight_decay() names are not decayed -- which is exactly the fp32 "vector" region -- so one fused AdamW launch
    per region updates master, moments and the bf16 copy (28 B/param of HBM traffic, no per-parameter kernels).
  * global grad-norm clipping (utils.py:860-861) is computed on the device (deterministic two-stage reduction) and
    consumed by the AdamW kernel through a device scalar: the step never synchronises with the host.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import ops

BF16, F32 = torch.bfloat16, torch.float32


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


class IVTrainEngine:
    def __init__(self, model, lr: float = 1.5e-4, betas=(0.9, 0.98), eps: float = 1e-6, weight_decay: float = 0.05,
                 max_grad_norm: float = 3.0, process_group=None, bucket_bytes: int = 256 << 20, overlap: bool = True,
                 clip_loss_ratio=(1.0, 1.0), mae_loss_ratio: float = 1.0, wgrad_stream: bool = False,
                 force_comm: bool = False):
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.clip_loss_ratio, self.mae_loss_ratio = clip_loss_ratio, mae_loss_ratio
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        # force_comm: run the bucketed RCCL reduction (side stream, per-block hooks) even on a 1-rank group -- how the multi-GPU
        # code path is exercised on a single-GPU box (tests, `bench.py --force-dist`)
        self.comm = self.world > 1 or (force_comm and dist.is_available() and dist.is_initialized())
        self.overlap = overlap and self.comm
        self.bucket_bytes = bucket_bytes
        self.step_count = 0
        dev = next(model.parameters()).device
        self.device = dev
        skip = set(model.no_weight_decay()) if hasattr(model, "no_weight_decay") else set()

        # ---- ordering: backward order -------------------------------------------------------------------------------
        named = list(model.named_parameters())
        depth = len(model.blocks)

        def order_key(item):
            name = item[0]
            if name.startswith("blocks."):
                return (1, depth - 1 - int(name.split(".")[1]))
            if name.startswith("patch_embed") or name in ("cls_token", "pos_embed"):
                return (2, 0)
            return (0, 0)                                     # decoders / projector: their grads are ready first

        named.sort(key=order_key)
        mats, vecs = [], []
        for name, p in named:
            decay = not (p.dim() == 1 or name.endswith(".bias") or name in skip)
            (mats if decay else vecs).append((name, p))
        self.mat_params, self.vec_params = mats, vecs

        def layout(items):
            offs, n = [], 0
            for _, p in items:
                offs.append(n)
                n += _align(p.numel())
            return offs, _align(n, 1024)

        self.mat_off, n_mat = layout(mats)
        self.vec_off, n_vec = layout(vecs)
        self.n_mat, self.n_vec = n_mat, n_vec
        self.master = torch.zeros(n_mat + n_vec, dtype=F32, device=dev)
        self.exp_avg = torch.zeros_like(self.master)
        self.exp_avg_sq = torch.zeros_like(self.master)
        self.shadow = torch.zeros(n_mat, dtype=BF16, device=dev)
        self.grad_mat = torch.zeros(n_mat, dtype=BF16, device=dev)
        self.grad_vec = torch.zeros(n_vec, dtype=F32, device=dev)
        for (name, p), off in zip(mats, self.mat_off):
            n = p.numel()
            self.master[off:off + n].copy_(p.detach().reshape(-1).float())
            p.data = self.master[off:off + n].view(p.shape)
            p._ivh_bf16 = self.shadow[off:off + n].view(p.shape[0], -1) if p.dim() >= 2 else self.shadow[off:off + n]
            p.main_grad = self.grad_mat[off:off + n].view(p.shape)
        for (name, p), off in zip(vecs, self.vec_off):
            n = p.numel()
            o = n_mat + off
            self.master[o:o + n].copy_(p.detach().reshape(-1).float())
            p.data = self.master[o:o + n].view(p.shape)
            p.main_grad = self.grad_vec[off:off + n].view(p.shape)
        self.shadow.copy_(self.master[:n_mat])                # initial bf16 compute copy
        # block index -> end offset (exclusive) of its matrices in grad_mat (prefix finished once that block's backward ran)
        self.block_end: Dict[int, int] = {}
        for (name, p), off in zip(mats, self.mat_off):
            if name.startswith("blocks."):
                i = int(name.split(".")[1])
                self.block_end[i] = max(self.block_end.get(i, 0), off + _align(p.numel()))
        self.head_end = min((off for (name, _), off in zip(mats, self.mat_off) if name.startswith("blocks.")), default=0)
        self._sumsq = torch.zeros(1, dtype=F32, device=dev)
        self._sq_scratch = torch.empty(4096, dtype=F32, device=dev)
        self._clip = None
        self.grad_norm = torch.zeros(1, dtype=F32, device=dev)
        self._reduced_upto = 0
        self.reduce_log: List[Tuple[int, int]] = []
        self.comm_stream = torch.cuda.Stream(device=dev) if (self.comm and dev.type == "cuda") else None
        # optional: weight-gradient GEMMs on their own stream, filling the CUs the dgrad chain leaves idle (functional._wgrad).
        # Off by default since the four wgrads of a block go out as one grouped launch that fills the GPU by itself
        # (measured on the 1B step: 140.1 ms without the stream, 142.6 ms with it; before grouping it was worth 20 ms).
        self.wgrad_stream = torch.cuda.Stream(device=dev) if (wgrad_stream and dev.type == "cuda") else None
        model.grad_ready_hook = self._on_block_done if self.overlap else None

    # ---- gradient reduction -------------------------------------------------------------------------------------------
    def _launch_reduce(self, lo: int, hi: int):
        if hi <= lo:
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
                dist.all_reduce(self.grad_mat[lo:hi], group=self.pg)
        else:                                                  # host tensors (gloo): same bucketing, no stream
            dist.all_reduce(self.grad_mat[lo:hi], group=self.pg)
        self.reduce_log.append((lo, hi))
        self._reduced_upto = hi

    def _on_block_done(self, i: int):
        """called by BlockStackFn.backward after block i: gradients of blocks >= i (and the heads) are final."""
        hi = self.block_end[i]
        if (hi - self._reduced_upto) * 2 >= self.bucket_bytes:
            self._launch_reduce(self._reduced_upto, hi)

    def _finish_reduce(self):
        from . import functional as Fn
        Fn._wgrad_flush(force=True)                            # weight gradients still queued for a grouped launch
        if self.wgrad_stream is not None:
            torch.cuda.current_stream().wait_stream(self.wgrad_stream)
        if not self.comm or getattr(self, "_defer_reduce", False):
            return
        self._launch_reduce(self._reduced_upto, self.n_mat)
        if self.comm_stream is not None:
            with torch.cuda.stream(self.comm_stream):
                dist.all_reduce(self.grad_vec, group=self.pg)
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            dist.all_reduce(self.grad_vec, group=self.pg)
        self._reduced_upto = 0

    # ---- optimizer ------------------------------------------------------------------------------------------------------
    def optimizer_step(self, lr: Optional[float] = None, weight_decay: Optional[float] = None):
        lr = self.lr if lr is None else lr
        wd = self.weight_decay if weight_decay is None else weight_decay
        self.step_count += 1
        gs = 1.0 / self.world
        clip = None
        if self.max_grad_norm and self.max_grad_norm > 0:
            ops.sqnorm(self.grad_mat, self._sumsq, False, self._sq_scratch)
            ops.sqnorm(self.grad_vec, self._sumsq, True, self._sq_scratch)
            clip, nrm = ops.clip_coef(self._sumsq, self.max_grad_norm * self.world)
            self.grad_norm = nrm / self.world
        b1, b2 = self.betas
        n_mat = self.n_mat
        ops.adamw_step(self.master[:n_mat], self.exp_avg[:n_mat], self.exp_avg_sq[:n_mat], self.grad_mat, self.shadow,
                       lr, b1, b2, self.eps, wd, self.step_count, gs, clip)
        ops.adamw_step(self.master[n_mat:], self.exp_avg[n_mat:], self.exp_avg_sq[n_mat:], self.grad_vec, None,
                       lr, b1, b2, self.eps, 0.0, self.step_count, gs, clip)

    # ---- one training step ------------------------------------------------------------------------------------------------
    def backward(self, loss: torch.Tensor):
        """loss.backward() with the weight-gradient GEMMs routed to the engine's wgrad stream."""
        from . import functional as Fn
        if self.wgrad_stream is not None:
            self.wgrad_stream.wait_stream(torch.cuda.current_stream())   # last step's optimizer read the gradient buffers
        prev, Fn.WGRAD_STREAM = Fn.WGRAD_STREAM, self.wgrad_stream
        try:
            loss.backward()
        finally:
            Fn.WGRAD_STREAM = prev

    # ---- HIP-graph mode (single GPU) ------------------------------------------------------------------------------------------
    def reduce_all_now(self):
        """the whole gradient reduction on the current stream, bucket by bucket, after backward has finished (no overlap): the
        companion of a graph-captured step on a multi-rank group (capture_step(defer_reduce=True))."""
        if not self.comm:
            return
        step = max(1, self.bucket_bytes // 2)
        for lo in range(0, self.n_mat, step):
            dist.all_reduce(self.grad_mat[lo:min(lo + step, self.n_mat)], group=self.pg)
        dist.all_reduce(self.grad_vec, group=self.pg)

    def capture_step(self, video: torch.Tensor, mask: torch.Tensor, targets, L: Optional[int] = None, warmup: int = 2,
                     defer_reduce: bool = False):
        """Capture mask -> indices + forward + fused loss + backward (both streams) of one step into a HIP graph.  The ~2200 kernel
        launches of a step cost ~120 ms of Python / ctypes / allocator time when issued one by one -- as long as the GPU work
        itself; replayed from a graph they cost the GPU front-end ~1 us each.  `video`, `mask` and `targets` become the graph's
        static inputs: write the next batch INTO them (copy_) before each `train_step_graphed()`.  The optimizer launches stay
        outside the graph (their step / lr arguments change every step).  Gradient reduction over ranks is not captured: use
        `train_step` when world_size > 1."""
        if self.comm and not defer_reduce:
            raise RuntimeError("capture_step: collectives are not captured; multi-GPU steps use train_step (eager, RCCL overlap) or "
                               "capture_step(defer_reduce=True) (graph replay, then the bucketed reduction without overlap)")
        from . import functional as Fn
        # defer_reduce: for hosts that cannot enqueue ~2200 launches per step as fast as the GPU retires them (several ranks sharing few
        # cores).  Forward + backward are replayed from the graph, the gradient buckets are reduced after it.  No collective is
        # issued during capture: the per-block hook is removed and _finish_reduce only flushes the queued weight gradients.
        self._defer_reduce = bool(self.comm)
        if self.comm:
            self.model.grad_ready_hook = None
        from .internvideo2_pretrain import build_gather_indices

        def body():
            self.zero_grad()
            vis_inv = build_gather_indices(mask, self.device, L=L, check=False) if mask.is_cuda else None
            loss, parts = self.model.forward_loss(video, mask, targets, self.clip_loss_ratio, self.mae_loss_ratio, vis_inv=vis_inv)
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
        self._graph = torch.cuda.CUDAGraph()
        Fn.WGRAD_KEEPALIVE = []
        try:
            with torch.cuda.graph(self._graph):
                self._graph_out = body()
        finally:
            keep, Fn.WGRAD_KEEPALIVE = Fn.WGRAD_KEEPALIVE, None
        del keep                                              # their memory stays in the graph's private pool
        return self._graph_out

    def train_step_graphed(self, lr: Optional[float] = None, weight_decay: Optional[float] = None):
        """replay the captured step on the current contents of the static inputs, then AdamW.  -> (loss, parts) device scalars."""
        self._graph.replay()
        if getattr(self, "_defer_reduce", False):
            self.reduce_all_now()
        self.optimizer_step(lr, weight_decay)
        return self._graph_out

    def zero_grad(self):
        """only the fp32 vector region accumulates (positional tables shared by several decoders); matrices are overwritten."""
        self.grad_vec.zero_()
        self._reduced_upto = 0
        self.reduce_log.clear()

    def train_step(self, video: torch.Tensor, mask: torch.Tensor, targets, vis_inv=None, lr: Optional[float] = None,
                   weight_decay: Optional[float] = None):
        """forward + fused distillation loss + backward + gradient all-reduce + AdamW.  Returns the loss as a device
        scalar (no host sync; the reference's per-step NaN check / .item() calls are left to the caller)."""
        self.zero_grad()
        loss, parts = self.model.forward_loss(video, mask, targets, self.clip_loss_ratio, self.mae_loss_ratio, vis_inv=vis_inv)
        self.backward(loss)
        self._finish_reduce()
        self.optimizer_step(lr, weight_decay)             # per-step lr / weight decay (engine_for_pretraining.py:56-61); None = the constructor's
        return loss.detach(), parts

    def state_dict(self):
        return {"master": self.master, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_count}

    def load_state_dict(self, sd):
        self.master.copy_(sd["master"]); self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])
        self.shadow.copy_(self.master[:self.n_mat])
