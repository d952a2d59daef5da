"""Autograd seams of the InternVideo2 student on top of the HIP kernels (internvideo_amd.ops).

Each torch.autograd.Function below owns a contiguous piece of the reference's forward
(InternVideo2/single_modality/models/internvideo2_pretrain.py, "P:") and its hand-derived backward; the pieces are
glued by PyTorch autograd only where tensors change hands between them (taps -> decoders).  No torch compute op
is on the block interiors: torch provides allocation, views and the autograd tape.  What torch itself still launches in a 1B step, by
the kernel trace (profiles/r5_step_sequence_eager_b128_v1.md): autograd's own accumulation where two decoders (and the pool) read one tap
(5 bf16 adds, 0.33 ms), four dtype copies at the stack's ends (0.3 ms), ~50 scalar-sized copies of the loss bookkeeping -- 0.25 % of the step.

Parameter conventions: matrices are consumed as bf16 (`mat()`), vectors as fp32 (`vec()`).  A parameter may carry
  ._ivh_bf16   : an up-to-date bf16 copy kept by the training engine (avoids a cast per step), and
  .main_grad   : a preallocated gradient buffer (bf16 for matrices / fp32 for vectors) that the backward kernels
                 write directly (the Function then returns None for that input) -- the native-engine mode;
  ._ivh_accum  : True = the parameter is used by several autograd nodes of one step (the stage-2 text / fusion tower: tied embeddings,
                 several passes through the same layers): its main_grad is zeroed by the engine at the start of the step and every
                 contribution is ADDED to it (never written in place by a kernel).
Without them the Functions cast on the fly and return gradients in the parameter's dtype (drop-in mode: works
with any torch optimizer / DDP / DeepSpeed wrapper, as SURVEY.md 8(b) B1/B2 require).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32


# Measurement aid (bench.py `encoder_fwd_bwd_frac`): while this is a list, the encoder's boundaries inside an EAGER step are recorded as
# (label, torch.cuda.Event) -- "enc_fwd_begin" (patch embed starts), "enc_fwd_end" (last block done), "enc_bwd_begin" (block stack's backward,
# after the decoders' queued weight gradients were flushed), "enc_bwd_end" (patch embed's backward done).  None = off (no event, no cost).
STEP_MARKS: Optional[list] = None
# bf16 stream: a tapped block's tap gradient joins dres inside the norm backward's loads (ivh_rmsnorm_add_bwd_bf16res dres_extra) instead of
# in a torch add of its own; IVH_TAP_ON_LOAD=0 restores the separate pass (A/B)
_TAP_ON_LOAD = __import__("os").environ.get("IVH_TAP_ON_LOAD", "1") != "0"


def _mark(label: str):
    if STEP_MARKS is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        STEP_MARKS.append((label, ev))


def mat(p: torch.Tensor) -> torch.Tensor:
    if p.dtype == BF16:
        return p.detach()
    sh = getattr(p, "_ivh_bf16", None)
    if sh is not None:
        return sh
    return p.detach().to(BF16)


def vec(p: torch.Tensor) -> torch.Tensor:
    d = p.detach()
    return d if d.dtype == F32 else d.float()


def _check_open(p: torch.Tensor):
    """engine mode on several ranks: a parameter whose gradient bucket has already been handed to the collective (engine._launch_reduce marks
    it `_ivh_closed`) must not receive another contribution -- it would reach this rank's optimizer only."""
    if getattr(p, "_ivh_closed", False):
        raise RuntimeError("a gradient was delivered to a parameter whose bucket had already been reduced: the autograd node that owns it ran "
                           "after the vision tower's backward began (IVTrainEngine assumes every node outside the tower runs before it)")


def _ret_grad(p: torch.Tensor, g: Optional[torch.Tensor], accumulate: bool = False):
    """Deliver gradient `g` of parameter `p`: into p.main_grad when the training engine provided one (returns None so
    that autograd does nothing), else back to autograd in the parameter's dtype / shape (drop-in mode)."""
    if g is None:
        return None
    mg = getattr(p, "main_grad", None)
    if mg is not None:
        _check_open(p)
        accumulate = accumulate or bool(getattr(p, "_ivh_accum", False))
        if g.data_ptr() != mg.data_ptr():
            if accumulate:
                mg.add_(g.reshape(mg.shape).to(mg.dtype))
            else:
                mg.copy_(g.reshape(mg.shape))
        return None
    return g.to(p.dtype).reshape(p.shape) if (g.dtype != p.dtype or g.shape != p.shape) else g


# Set by the training engine: a second HIP stream for the weight-gradient GEMMs.  A wgrad has few, long workgroups
# (e.g. 144 tiles x 209 K steps for fc1 on 256 CUs) and nothing downstream in backward consumes it, while the dgrad chain
# on the main stream leaves CUs idle in the last dispatch round of most of its GEMMs (318 tiles on 256 CUs): issued on its
# own stream the hardware dispatcher packs the wgrad workgroups into those gaps.  None = everything on the current stream.
WGRAD_STREAM: Optional["torch.cuda.Stream"] = None
# While a step is being captured into a HIP graph the caching allocator cannot be told about cross-stream use
# (Tensor.record_stream); the engine then sets this to a list and the operands are simply kept alive until the streams join.
WGRAD_KEEPALIVE: Optional[list] = None


def _wgrad(dy: torch.Tensor, x: torch.Tensor, p: torch.Tensor, k_dev: Optional[torch.Tensor] = None):
    """dW[n,k] = sum_m dy[m,n] x[m,k]  (both operands rows-contiguous: transposing LDS reads, no HBM transposes).
    Written straight into p.main_grad when present (and then, if the engine provided one, on the wgrad stream).
    k_dev: only the first *k_dev rows of dy / x exist (DropPath skipping: ops.gemm k_dev)."""
    if k_dev is not None:                                 # (the compacted block-stack path: always the 256^2 kernel, on the current stream)
        mg = getattr(p, "main_grad", None)
        if mg is not None:
            _check_open(p)
        if mg is not None and not getattr(p, "_ivh_accum", False) and mg.dtype == BF16 and mg.numel() == dy.shape[1] * x.shape[1]:
            ops.gemm(dy, x, a_kc=False, b_kc=False, out=mg.view(dy.shape[1], x.shape[1]), k_dev=k_dev)
            return mg
        g = ops.gemm(dy, x, a_kc=False, b_kc=False, k_dev=k_dev)
        if mg is not None:
            if getattr(p, "_ivh_accum", False):
                mg.add_(g.reshape(mg.shape).to(mg.dtype))
            else:
                mg.copy_(g.reshape(mg.shape))
            return mg
        return g
    mg = getattr(p, "main_grad", None)
    if mg is not None:
        _check_open(p)
    if mg is not None and getattr(p, "_ivh_accum", False):
        if dy.shape[0] % 8:
            pad = 8 - dy.shape[0] % 8
            dy, x = torch.nn.functional.pad(dy, (0, 0, 0, pad)), torch.nn.functional.pad(x, (0, 0, 0, pad))
        mg.add_(ops.gemm(dy, x, a_kc=False, b_kc=False).reshape(mg.shape).to(mg.dtype))
        return mg
    if mg is not None and mg.dtype in (BF16, F32) and mg.numel() == dy.shape[1] * x.shape[1]:
        out = mg.view(dy.shape[1], x.shape[1])
        st = WGRAD_STREAM
        if st is None:
            ops.gemm(dy, x, a_kc=False, b_kc=False, out=out, out_fp32=(mg.dtype == F32))
            return mg
        ev = torch.cuda.Event()
        ev.record()                                   # dy and x are complete on the current stream at this point
        st.wait_event(ev)
        with torch.cuda.stream(st):
            # beside the dgrad chain the goal is the fewest CU-microseconds, not the shortest launch: the 256^2 kernel does the
            # same FLOPs on 36-144 CUs where the 128^2 one would spread its tiles over every CU the other stream wants
            ops.gemm(dy, x, a_kc=False, b_kc=False, out=out, out_fp32=(mg.dtype == F32), kernel=2)
        if WGRAD_KEEPALIVE is not None:
            WGRAD_KEEPALIVE.append((dy, x))
        else:
            dy.record_stream(st); x.record_stream(st)  # the caching allocator must not recycle them under the side stream
        return mg
    if dy.shape[0] % 8:                               # rows-contiguous operands are read in 8-row groups: zero-pad a ragged row count
        pad = 8 - dy.shape[0] % 8
        dy, x = torch.nn.functional.pad(dy, (0, 0, 0, pad)), torch.nn.functional.pad(x, (0, 0, 0, pad))
    return ops.gemm(dy, x, a_kc=False, b_kc=False)


# Deferred weight gradients (engine mode: bf16 main_grad buffers).  A wgrad GEMM has few, long tiles (K = B*L): one transformer
# block's four have 426 tiles of the 256^2 kernel = 1.66 rounds of the 256 CUs, three blocks' twelve have 1278 = 4.99 rounds.
# Nothing in backward consumes a weight gradient, so they are queued with their operands and launched as grouped GEMMs
# (ops.gemm_grouped) whenever the queued tiles of one K fill the CUs (see _wgrad_group_size), and at the end of backward.
WGRAD_GROUP_MAX = 32                                      # problems per grouped launch (Gemm256Params::prob)
PATCH_WGRAD_SPLIT = __import__("os").environ.get("IVH_PATCH_WGRAD_SPLIT", "1") != "0"   # patch-embed weight gradient cut along the token axis (A/B: 0)
WGRAD_MIX_K = __import__("os").environ.get("IVH_WGRAD_MIX_K", "1") != "0"   # forced flushes group problems of different K (A/B: 0)
WGRAD_FILL = 0.95                                         # launch as soon as the queued tiles fill their last round of CUs this well
_N_CU = [0]


def _wgrad_tiles(q) -> int:
    return ((q[0].shape[1] + 255) // 256) * ((q[1].shape[1] + 255) // 256)


def _wgrad_group_size(same) -> int:
    """how many of the queued problems (same K, in queue order) to launch now: the largest prefix, in whole blocks of four GEMMs, whose
    256 x 256 tiles fill their last round of the CUs to WGRAD_FILL -- 12 (three blocks, 1278 tiles = 4.99 rounds) for the 1B model, 28
    (seven blocks, 756 tiles = 2.95 rounds) for ViT-B/14, 4 (one block, 1963 tiles = 7.67 rounds) for the 6B model.  0 = keep queueing."""
    if _N_CU[0] == 0:
        _N_CU[0] = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count or 256
    ncu = _N_CU[0]
    best = 0
    tiles = 0
    for i, q in enumerate(same[:WGRAD_GROUP_MAX]):
        tiles += _wgrad_tiles(q)
        n = i + 1
        if n % 4 == 0 and tiles / (-(-tiles // ncu) * ncu) >= WGRAD_FILL:
            best = n
    if best == 0 and len(same) >= WGRAD_GROUP_MAX:
        best = WGRAD_GROUP_MAX
    return best


def _mixing_pays(queue) -> bool:
    """would ONE grouped launch over problems of different K be shorter than one launch per K?  Rounds of the CUs x K steps: a launch per K
    group costs ceil(tiles_g / CUs) K_g each; the merged launch at most ceil(all tiles / CUs) K_max (a workgroup's tiles are a mix).  The
    1B decoders (468 tiles over 53376 rows + 288 over 53248) go from 2 + 2 rounds to 3; the stage-2 text tower's groups (K = 2048 and 6144,
    each several rounds) do not merge: measured 1.1 % slower merged."""
    if _N_CU[0] == 0:
        _N_CU[0] = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count or 256
    ncu = _N_CU[0]
    by_k: dict = {}
    for q in queue:
        by_k[q[0].shape[0]] = by_k.get(q[0].shape[0], 0) + _wgrad_tiles(q)
    apart = sum(-(-t // ncu) * k for k, t in by_k.items())
    merged = -(-sum(by_k.values()) // ncu) * max(by_k)
    return merged < 0.95 * apart


_wgrad_queue: list = []                                   # [(dy, x, out_view)]


class _PendingGrad:
    """a weight gradient queued for a grouped launch in drop-in (plain autograd) mode: the bf16 result buffer and the parameter it belongs
    to; BlockStackFn.backward resolves it -- after the flush that fills the buffer -- into the tensor it returns to autograd"""
    __slots__ = ("out", "p")

    def __init__(self, out, p):
        self.out, self.p = out, p

    def resolve(self):
        return _ret_grad(self.p, self.out)


_DEFER_DROPIN = [False]                                   # set by BlockStackFn.backward: it returns every gradient at its end, so it may defer

# Plain-autograd mode, weight gradients of SEPARATE autograd nodes (the ~150 Linear layers of the stage-2 text / fusion tower): each is a
# 1024-wide GEMM over 4-8k rows = 16-64 tiles for 256 CUs.  Inside `with grouped_weight_grads():` such a node queues its operands and
# returns None for the weight; when the backward pass ends (autograd's queue_callback) the queue is flushed as grouped launches and the
# results are accumulated into `.grad` by hand.  Opt-in because it bypasses autograd for those parameters: tensor hooks on them do not
# fire (DistributedDataParallel relies on such hooks) and torch.autograd.grad() does not see them.
_END_DEFER = [False]
_end_pending: list = []                                   # [(bf16 result buffer, [(parameter, first row, rows)])]
_end_task = [-1]                                          # autograd graph task the callback is registered with


class grouped_weight_grads:
    """context manager: weight gradients of Linear-like nodes are computed as grouped GEMMs at the end of the backward pass(es) run inside"""

    def __enter__(self):
        self.prev, _END_DEFER[0] = _END_DEFER[0], True
        return self

    def __exit__(self, *exc):
        _END_DEFER[0] = self.prev
        return False


def _end_of_backward():
    """autograd engine callback: every queued weight gradient -> grouped launches -> accumulated into .grad"""
    pend = list(_end_pending)
    _end_pending.clear()
    _end_task[0] = -1
    _wgrad_flush(force=True)
    for out, parts in pend:
        for p, r0, n in parts:
            if not p.requires_grad:                       # a frozen part of a concatenated weight (freeze_text / freeze_vision): no .grad
                continue
            mg = getattr(p, "main_grad", None)
            if mg is not None:                            # engine-managed: added to the flat buffer (zeroed at the start of the step)
                _check_open(p)
                mg.add_(out[r0:r0 + n].reshape(mg.shape).to(mg.dtype))
                continue
            g = out[r0:r0 + n].to(p.dtype).reshape(p.shape)
            p.grad = g if p.grad is None else p.grad + g


def resolve_end_pending():
    """engine mode: the weight gradients queued for the end of the backward pass are needed EARLIER -- the vision tower's backward (one
    autograd node, run after everything that consumes its outputs) starts the bucketed gradient reduction, and the text / fusion
    tower's buckets come first.  Called by BlockStackFn.backward; the autograd callback then finds nothing left."""
    if _end_pending and any(getattr(p, "main_grad", None) is not None for _, parts in _end_pending for p, _, _ in parts):
        _end_of_backward()


def _defer_to_end(dy: torch.Tensor, x: torch.Tensor, parts) -> bool:
    """queue dW = dy^T x (rows of the result belong to the parameters in `parts`: [(p, first row, rows)]) for the end of the running
    backward pass; False = not applicable (not inside `grouped_weight_grads()` / a backward pass, ragged rows, engine-managed gradients)"""
    if not _END_DEFER[0] or dy.shape[0] % 8 or torch._C._current_graph_task_id() == -1:
        return False
    has_mg = [getattr(p, "main_grad", None) is not None for p, _, _ in parts]
    if any(has_mg) and not (all(has_mg) and all(getattr(p, "_ivh_accum", False) for p, _, _ in parts)):
        return False                                      # engine-managed parameters that are written in place: not through this queue
    if not any(p.requires_grad for p, _, _ in parts):     # every part frozen: nothing to compute, nothing for autograd
        return True
    out = torch.empty((dy.shape[1], x.shape[1]), dtype=BF16, device=dy.device)
    task = torch._C._current_graph_task_id()
    if task != _end_task[0]:
        if _end_pending:                                  # left behind by a backward pass that raised before its callback ran: drop them
            global _wgrad_queue
            stale = {id(o) for o, _ in _end_pending}
            _wgrad_queue = [q for q in _wgrad_queue if id(q[2]) not in stale]
            _end_pending.clear()
        torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
        _end_task[0] = task
    _wgrad_queue.append((dy, x, out))
    _end_pending.append((out, list(parts)))
    return True


def _wgrad_defer(dy: torch.Tensor, x: torch.Tensor, p: torch.Tensor, k_dev: Optional[torch.Tensor] = None):
    """queue dW = dy^T x for a grouped launch: into `p.main_grad` when the training engine provided a bf16 one (-> None: nothing for
    autograd), into a temporary bf16 buffer when the caller collects its gradients at the end of its backward (-> _PendingGrad);
    otherwise compute it now and hand it to autograd.  k_dev: device-side row count of dy / x (DropPath skipping), kept with the queue entry."""
    mg = getattr(p, "main_grad", None)
    if mg is None and not p.requires_grad:                # frozen weight (freeze_text / freeze_vision, a frozen teacher): autograd is
        return None                                       # bypassed here, so it cannot drop the gradient for us -- no GEMM, no .grad
    if mg is not None:
        _check_open(p)
    if mg is None and _DEFER_DROPIN[0] and (dy.shape[0] % 8 == 0 or k_dev is not None):
        out = torch.empty((dy.shape[1], x.shape[1]), dtype=BF16, device=dy.device)
        _wgrad_queue.append((dy, x, out, k_dev))
        return _PendingGrad(out, p)
    if k_dev is None and (mg is None or getattr(p, "_ivh_accum", False)) and _defer_to_end(dy, x, [(p, 0, dy.shape[1])]):
        return None
    if mg is None or mg.dtype != BF16 or mg.numel() != dy.shape[1] * x.shape[1] or getattr(p, "_ivh_accum", False):
        return _ret_grad(p, _wgrad(dy, x, p, k_dev))
    _wgrad_queue.append((dy, x, mg.view(dy.shape[1], x.shape[1]), k_dev))
    return None


def _wgrad_flush(force: bool = False):
    """launch queued weight gradients of one K in groups that fill the CUs (`_wgrad_group_size`); everything that is queued when `force`."""
    global _wgrad_queue
    while _wgrad_queue:
        K = _wgrad_queue[0][0].shape[0]
        # a forced flush takes whatever is queued, whatever its K (the grouped kernel carries one K per problem): the decoders' weight
        # gradients over B L rows (CLIP branch) and B (L - 1) rows (MAE branch, no cls) fill 2.95 rounds of the CUs together where the two
        # K groups launched apart fill 1.83 + 1.13, i.e. four rounds
        same = [q for q in _wgrad_queue if q[0].shape[0] == K]
        if force and WGRAD_MIX_K and len(same) < len(_wgrad_queue) and _mixing_pays(_wgrad_queue[:WGRAD_GROUP_MAX]):
            same = list(_wgrad_queue)
        n_now = len(same[:WGRAD_GROUP_MAX]) if force else _wgrad_group_size(same)
        if n_now == 0:
            return
        batch = same[:n_now]
        ids = {id(q) for q in batch}
        _wgrad_queue = [q for q in _wgrad_queue if id(q) not in ids]
        st = WGRAD_STREAM

        def launch():
            if len(batch) > 1:
                ops.gemm_grouped(batch, a_kc=False, b_kc=False)
            else:
                q = batch[0]
                ops.gemm(q[0], q[1], a_kc=False, b_kc=False, out=q[2], k_dev=(q[3] if len(q) > 3 else None))
        if st is None:
            launch()
            continue
        ev = torch.cuda.Event()
        ev.record()
        st.wait_event(ev)
        with torch.cuda.stream(st):
            launch()
        for dy, x, *_ in batch:
            if WGRAD_KEEPALIVE is not None:
                WGRAD_KEEPALIVE.append((dy, x))
            else:
                dy.record_stream(st); x.record_stream(st)


def _vgrad(p: torch.Tensor, g: torch.Tensor):
    return g


def _act_d(act: str) -> str:
    """the derivative-exchanging flavour of an activation, when there is one: fc1's epilogue then stores gelu'(u) (a by-product of
    its erf) instead of u, and fc2's dgrad epilogue is a plain multiply"""
    return "gelu_erf_d" if act in ("gelu", "gelu_erf", "erf") else act


def _mg(p: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """the fp32 main_grad of a vector parameter, for kernels that can write the gradient there directly (no copy kernel)"""
    if p is None:
        return None
    mg = getattr(p, "main_grad", None)
    if getattr(p, "_ivh_accum", False):
        return None
    return mg if (mg is not None and mg.dtype == F32) else None


def _wants_grad(ctx, i: int, p: torch.Tensor) -> bool:
    """does input i (parameter p) of the running backward need a gradient?  (engine-managed parameters always do: main_grad is written)"""
    return bool(ctx.needs_input_grad[i]) or getattr(p, "main_grad", None) is not None


class LinearFn(torch.autograd.Function):
    """y = x W^T + b on bf16 rows (nn.Linear, P:158,160,341)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.dtype != BF16:
            x2 = x2.to(BF16)
        y = ops.gemm(x2, mat(w), bias=vec(b) if b is not None else None)
        ctx.save_for_backward(x2)
        ctx.w, ctx.b = w, b
        ctx.xshape, ctx.xdtype = x.shape, x.dtype
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        w, b = ctx.w, ctx.b
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        if dy2.dtype != BF16:
            dy2 = dy2.to(BF16)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dy2, mat(w), a_kc=True, b_kc=False).reshape(ctx.xshape).to(ctx.xdtype)
        dw = _wgrad_defer(dy2, x2, w) if _wants_grad(ctx, 1, w) else None
        db = _ret_grad(b, _vgrad(b, ops.colsum_bf16(dy2))) if (b is not None and _wants_grad(ctx, 2, b)) else None
        return dx, dw, db


# fp8 (e4m3) Linear -- BASELINE configs[4].  Weights are quantised (plain + transposed copy) once per optimizer step: the cache on the
# parameter is keyed on its storage, its version counter and WEIGHT_EPOCH, which the training engine bumps after every AdamW launch
# (the fused optimizer writes parameters from a kernel, invisible to the version counter).
WEIGHT_EPOCH = 0
# While a step is being captured into a HIP graph the cache above must NOT serve: a hit records no quantisation kernel, and every replay
# would then read the e4m3 weights frozen at capture time while AdamW keeps moving the master / bf16 copies (the model would silently stop
# training).  Under capture every weight is quantised inside the graph -- once per captured step (this dict, armed by the engine around
# its capture, keyed on the parameter) into buffers of the graph's private pool, so every replay re-quantises the CURRENT weights.
FP8_CAPTURE_CACHE: Optional[dict] = None


def fp8_weight(w: torch.Tensor, channel: bool = False):
    """(wq [N, K], wqt [K, N16], s_fwd, s_dgrad) of a weight matrix, cached on the parameter.  Per-tensor scaling: s_fwd = s_dgrad = one fp32
    scale.  `channel` (model.fp8_weight_scales = "channel"): wq carries one scale per output feature n (s_fwd [N]) and wqt one per input
    feature k (s_dgrad [K]) -- each GEMM's B operand is scaled along that GEMM's output dimension (ops.fp8_quantize_weight)."""
    def quantise():
        wb = mat(w)
        wb = wb.reshape(wb.shape[0], -1)
        if channel:
            return ops.fp8_quantize_weight(wb)
        q, qt, sc = ops.fp8_quantize(wb, want_transposed=True)
        return q, qt, sc, sc
    if w.is_cuda and torch.cuda.is_current_stream_capturing():
        cc = FP8_CAPTURE_CACHE
        c = cc.get((id(w), channel)) if cc is not None else None
        if c is None:
            c = quantise()
            if cc is not None:
                cc[(id(w), channel)] = c
        return c
    key = (w.data_ptr(), w._version, WEIGHT_EPOCH, channel)
    c = getattr(w, "_ivh_fp8", None)
    if c is None or c[0] != key:
        c = (key,) + tuple(quantise())
        w._ivh_fp8 = c
    return c[1:]


class Fp8History:
    """Delayed scaling for the fp8 block GEMMs (`model.fp8_scaling = "delayed"`): every activation / gradient quantisation site quantises
    with an amax carried over from the previous steps (one pass over the tensor, no amax pass) and records its own max|x| for the next ones.
    The carried value is a decaying maximum, amax <- max(this step's max|x|, decay * amax): it rises at once and falls slowly, which is what a
    window maximum does, without a host-side window index (the roll is three tiny kernels on HBM-resident arrays and is captured with the
    step into a HIP graph unchanged).  A site seen for the first time is quantised with current scaling and recorded."""

    def __init__(self, device, capacity: int = 4096, decay: float = 0.95):
        self.cur = torch.zeros(capacity, dtype=F32, device=device)            # amax to quantise with
        self.nxt = torch.zeros(capacity, dtype=torch.int32, device=device)    # bit patterns of the max|x| collected this step
        self.decay = float(decay)
        self.slot: dict = {}
        self.ready: set = set()
        self.collected: set = set()

    def roll(self):
        """start of a step: fold what the last step collected into the carried amax"""
        if not self.collected:
            return
        self.cur.mul_(self.decay)
        torch.maximum(self.cur, self.nxt.view(F32), out=self.cur)             # non-negative floats: the bit patterns ARE the values
        self.nxt.zero_()
        self.ready |= self.collected
        self.collected = set()

    def quantize(self, key, x: torch.Tensor, want_transposed: bool):
        i = self.slot.setdefault(key, len(self.slot))
        if i >= self.cur.numel():
            raise RuntimeError("Fp8History: more quantisation sites than slots")
        self.collected.add(i)
        if i in self.ready:
            return ops.fp8_quantize(x, want_transposed=want_transposed, amax_prev=self.cur[i:i + 1], amax_next=self.nxt[i:i + 1])
        return ops.fp8_quantize(x, want_transposed=want_transposed, amax_next=self.nxt[i:i + 1])


FP8_LINEAR_CHANNEL_SCALES = False       # Fp8LinearFn: per-channel weight scales (the block stack takes the switch from its meta)


class Fp8LinearFn(torch.autograd.Function):
    """y = x W^T + b with all three GEMMs (forward, dgrad, wgrad) on the fp8 MFMA path: per-tensor-scaled e4m3 operands, fp32
    accumulation, bf16 results.  x [.., K] bf16, W [N, K]; N and K multiples of 16 (every InternVideo2 width is)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.dtype != BF16:
            x2 = x2.to(BF16)
        need_w = w.requires_grad or getattr(w, "main_grad", None) is not None
        xq, xqt, sx = ops.fp8_quantize(x2.contiguous(), want_transposed=need_w)
        wq, wqt, sw, swt = fp8_weight(w, FP8_LINEAR_CHANNEL_SCALES)
        y = ops.gemm_fp8(xq, wq, sx, sw, bias=vec(b) if b is not None else None)
        ctx.save_for_backward(xqt, sx, wqt, swt)
        ctx.w, ctx.b = w, b
        ctx.xshape, ctx.xdtype = x.shape, x.dtype
        return y.reshape(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        xqt, sx, wqt, sw = ctx.saved_tensors
        w, b = ctx.w, ctx.b
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        if dy2.dtype != BF16:
            dy2 = dy2.to(BF16)
        dyq, dyqt, sd = ops.fp8_quantize(dy2, want_transposed=xqt is not None)
        dx = None
        if ctx.needs_input_grad[0]:                       # dX[m, k] = sum_n dY[m, n] W[n, k]: contraction over n on W's transposed copy
            dx = ops.gemm_fp8(dyq, wqt, sd, sw, k=dy2.shape[1]).reshape(ctx.xshape).to(ctx.xdtype)
        dw = None
        if xqt is not None:                               # dW[n, k] = sum_m dY[m, n] X[m, k]: contraction over the (zero-padded) token axis
            mg = getattr(w, "main_grad", None)
            out = mg.view(w.shape[0], -1) if (mg is not None and mg.dtype in (BF16, F32)) else None
            g = ops.gemm_fp8(dyqt, xqt, sd, sx, out=out, out_fp32=(out is not None and out.dtype == F32))
            dw = None if out is not None else _ret_grad(w, g)
        db = _ret_grad(b, _vgrad(b, ops.colsum_bf16(dy2))) if b is not None else None
        return dx, dw, db


class MlpFn(torch.autograd.Function):
    """y = fc2(gelu(fc1(x))) (Mlp P:220-244 / FusedMLP P:268-269 / MLP_Decoder head P:375-379).
    fc1's epilogue writes both the pre-activation u (for gelu') and g = gelu(u); fc2's dgrad epilogue multiplies by gelu'(u)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, act):
        x2 = x.reshape(-1, x.shape[-1])
        act = _act_d(act)
        g, u = ops.gemm(x2, mat(w1), bias=vec(b1), act=act, want_preact=True)      # u = gelu'(pre-activation) for "gelu_erf_d"
        y = ops.gemm(g, mat(w2), bias=vec(b2))
        ctx.save_for_backward(x2, u, g)
        ctx.p = (w1, b1, w2, b2)
        ctx.act, ctx.xshape = act, x.shape
        return y.reshape(*x.shape[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, u, g = ctx.saved_tensors
        w1, b1, w2, b2 = ctx.p
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        if dy2.dtype != BF16:
            dy2 = dy2.to(BF16)
        du = ops.gemm(dy2, mat(w2), a_kc=True, b_kc=False, dact_in=u, act=ctx.act)
        dw2 = _wgrad_defer(dy2, g, w2) if _wants_grad(ctx, 3, w2) else None      # queued for a grouped launch where a flush is guaranteed
        db2 = _ret_grad(b2, _vgrad(b2, ops.colsum_bf16(dy2))) if _wants_grad(ctx, 4, b2) else None   # (engine buffers / grouped_weight_grads())
        dx = ops.gemm(du, mat(w1), a_kc=True, b_kc=False).reshape(ctx.xshape) if ctx.needs_input_grad[0] else None
        dw1 = _wgrad_defer(du, x2, w1) if _wants_grad(ctx, 1, w1) else None
        db1 = _ret_grad(b1, _vgrad(b1, ops.colsum_bf16(du))) if _wants_grad(ctx, 2, b1) else None
        return dx, dw1, db1, dw2, db2, None


class PatchEmbedGatherFn(torch.autograd.Function):
    """PatchEmbed Conv3d (P:320-331) + cls cat + pos_embed add (P:634-656) + visible gather (P:659), computed for the
    visible tokens only: im2col of the kept tubelets -> MFMA GEMM -> fp32 residual stream rows."""

    @staticmethod
    def forward(ctx, video, vis_idx, inv_idx, proj_w, proj_b, cls_token, pos_embed, tubelet, patch):
        _mark("enc_fwd_begin")
        B, L = vis_idx.shape
        D = proj_w.shape[0]
        kreal = proj_w[0].numel()
        kp = (kreal + 63) // 64 * 64
        wp = torch.zeros((D, kp), dtype=BF16, device=video.device)
        wp[:, :kreal] = mat(proj_w).reshape(D, kreal)
        cols = ops.patch_im2col(video, vis_idx, tubelet, patch, kp)
        tok = ops.gemm(cols, wp, bias=vec(proj_b))
        x0 = ops.assemble_tokens(tok, vec(cls_token).reshape(-1), vec(pos_embed).reshape(-1, D), vis_idx)
        ctx.save_for_backward(cols, vis_idx, inv_idx)
        ctx.p = (proj_w, proj_b, cls_token, pos_embed)
        ctx.meta = (B, L, D, kreal)
        return x0

    @staticmethod
    def backward(ctx, dx0):
        cols, vis_idx, inv_idx = ctx.saved_tensors
        proj_w, proj_b, cls_token, pos_embed = ctx.p
        B, L, D, kreal = ctx.meta
        dx0 = dx0.contiguous()
        dtok = ops.rows_to_bf16(dx0, B, L, 1)
        # dW = dtok^T cols is [D, Kp] = 6 x 3 tiles of 256^2 for the 1B model under a contraction over all B (L - 1) token rows: 18 workgroups
        # on 256 CUs (1.04 ms at B = 128 on the 128^2 kernel).  Cut along the token axis into S batch entries (a batched launch: S x 18 tiles)
        # whose bf16 partial products are summed in fp32: 0.2 ms
        rows = dtok.shape[0]
        S = next((s_ for s_ in (16, 12, 8, 6, 4, 3, 2) if rows % s_ == 0 and rows // s_ >= 2048), 1) if PATCH_WGRAD_SPLIT else 1
        if S > 1:
            dwp = ops.gemm(dtok.view(S, rows // S, D), cols.view(S, rows // S, cols.shape[1]), a_kc=False, b_kc=False).float().sum(0)
        else:
            dwp = ops.gemm(dtok, cols, a_kc=False, b_kc=False)                  # [D, Kp]
        dw = dwp[:, :kreal].reshape(proj_w.shape)
        db = ops.colsum_bf16(dtok)
        dpos = ops.pos_grad(dx0, 1, B, L, inv_idx, 0)                           # [N1, D]; row 0 == sum_b dx0[b,0] == dcls
        ret = (None, None, None, _ret_grad(proj_w, dw), _ret_grad(proj_b, _vgrad(proj_b, db)),
               _ret_grad(cls_token, _vgrad(cls_token, dpos[0].clone())), _ret_grad(pos_embed, _vgrad(pos_embed, dpos)), None, None)
        _mark("enc_bwd_end")
        return ret


BLOCK_PARAM_NAMES = ("norm1.weight", "attn.qkv.weight", "attn.q_norm.weight", "attn.k_norm.weight", "attn.proj.weight",
                     "attn.proj.bias", "ls1.gamma", "norm2.weight", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight",
                     "mlp.fc2.bias", "ls2.gamma")
NBP = len(BLOCK_PARAM_NAMES)


# DropPath skipping (BlockStackFn._block_forward, include/internvideo_hip.h "DropPath SAMPLE SKIPPING").  IVH_DROPPATH_SKIP=0 restores the
# compute-then-multiply-by-zero path (same-box A/B; the two agree to the rounding of the weight gradients' summation order).
DROPPATH_SKIP = __import__("os").environ.get("IVH_DROPPATH_SKIP", "1") != "0"


def dp_skip_applies(meta, M: int, D: int, params) -> bool:
    """can this block stack run its DropPath branches on the kept samples only?  Needs the kernels that take device-side counts: the 256^2
    bf16 GEMM (operands below 2 GiB, erf-GELU with the derivative exchanged) and the 32x32 attention kernels (head dim <= 128, 16-byte
    strides) -- otherwise the stack silently keeps the multiply-by-zero path (same results)."""
    mode = meta.get("dp_skip", "auto")
    if not DROPPATH_SKIP or mode is False or mode is None or meta.get("fp8"):
        return False
    if _act_d(meta["act"]) != "gelu_erf_d":
        return False
    H, L = meta["H"], meta["L"]
    hd = D // H
    if hd % 8 or hd > 128 or D % 8:                          # (above 96 the dK / dV pass runs on the 16x16 kernel, which takes the count too)
        return False
    widest = max(3 * D, int(params[8].shape[0]))                                             # qkv / fc1 output widths
    if M * widest * 2 >= (1 << 31) - (1 << 24):
        return False
    return mode is True or M * D >= 512 * 512               # "auto": small stacks stay on the per-shape kernel choice (128^2 GEMM tiles)


class BlockStackFn(torch.autograd.Function):
    """All transformer blocks (P:247-297, loop P:664-683) with the fused residual protocol of the reference's
    DropoutAddRMSNorm path (P:282-287): the fp32 residual stream is updated inside the norm kernels
    (res += droppath * layerscale * branch), LayerScale (P:131-146) and DropPath (P:264,274) never run as
    separate passes, q/k RMSNorm (P:198-206) is in place on the packed qkv buffer and attention reads it packed.
    Returns the residual-stream value after every block listed in `taps` (ascending; P:669-688)."""

    @staticmethod
    def _block_forward(res, branch, g_prev, rs_prev, prm, rowscale, i, meta, plan=None):
        """one block: -> the tuple `backward` consumes (res1, rstd1, n1, qkv, rq, rk, att, lse, b1, res2, rstd2, n2, u, g, b2, rs1, rs2, q8).
        plan = (slot int32 [depth, 2, B], count int32 [depth, 2, 2]) of ops.droppath_plan: DROPPATH SKIPPING -- the samples a branch's DropPath
        draw zeroes (P:264,274,283-286) are not computed: the norm in front of the branch writes its output compacted over the kept samples,
        every kernel of the branch runs on the kept rows (the count stays on the device: the step is a replayed HIP graph) and the next
        residual add reads the branch back through the same map.  n1, qkv, rq, rk, att, lse, b1 are then compacted by slot[i, 0] and n2, u, g,
        b2 by slot[i, 1] (rows behind the kept ones hold nothing); the incoming `branch` by slot[i - 1, 1]."""
        B, L, H, eps, act = meta["B"], meta["L"], meta["H"], meta["eps"], _act_d(meta["act"])
        fp8 = bool(meta.get("fp8"))
        (n1w, qkvw, qnw, knw, projw, projb, ls1, n2w, fc1w, fc1b, fc2w, fc2b, ls2) = prm
        q8 = {} if fp8 else None
        if plan is not None:
            slot, cnt = plan
            s_in = slot[i - 1, 1] if i > 0 else None                                # keep map of the incoming branch (the previous block's MLP)
            s1, s2 = slot[i, 0], slot[i, 1]
            nb1, m1, m2 = cnt[i, 0, 0:1], cnt[i, 0, 1:2], cnt[i, 1, 1:2]            # kept clips / rows of the attention branch, rows of the MLP branch
        else:
            s_in = s1 = s2 = nb1 = m1 = m2 = None

        def lin(name, x, w, bias=None, act_=None, want_preact=False, m_dev=None):
            """x W^T (+ bias, activation): bf16 MFMA GEMM, or -- meta["fp8"] -- per-tensor-scaled e4m3 operands on the MX MFMA path; the
            transposed fp8 copy of x is what the weight gradient contracts over, so it is saved instead of the bf16 activation"""
            if not fp8:
                return ops.gemm(x, mat(w), bias=bias, act=act_, want_preact=want_preact, m_dev=m_dev)
            hist = meta.get("fp8_hist")
            xq, xqt, sx = hist.quantize((i, name), x, True) if hist is not None else ops.fp8_quantize(x, want_transposed=True)
            wq, _, sw, _ = fp8_weight(w, bool(meta.get("fp8_wchan")))
            q8[name] = (xqt, sx)
            return ops.gemm_fp8(xq, wq, sx, sw, bias=bias, act=act_, want_preact=want_preact)

        if branch is None:
            res1 = res
            _, n1, rstd1 = ops.rmsnorm_add_fwd(res, None, None, None, L, vec(n1w), eps, want_res_out=False, y_slot=s1)
        else:
            res1, n1, rstd1 = ops.rmsnorm_add_fwd(res, branch, g_prev, rs_prev, L, vec(n1w), eps, branch_slot=s_in, y_slot=s1)
        qkv = lin("n1", n1, qkvw, m_dev=m1)
        rq, rk = ops.qk_rmsnorm_fwd(qkv, vec(qnw), vec(knw), eps, m_dev=m1)
        att, lse = ops.flash_attn_fwd_packed(qkv, B, L, H, nb_dev=nb1)
        b1 = lin("att", att, projw, bias=vec(projb), m_dev=m1)
        rs1 = rowscale[i, 0] if rowscale is not None else None
        rs2 = rowscale[i, 1] if rowscale is not None else None
        g1 = vec(ls1) if ls1 is not None else None
        res2, n2, rstd2 = ops.rmsnorm_add_fwd(res1, b1, g1, rs1, L, vec(n2w), eps, branch_slot=s1, y_slot=s2)
        g, u = lin("n2", n2, fc1w, bias=vec(fc1b), act_=act, want_preact=True, m_dev=m2)
        b2 = lin("g", g, fc2w, bias=vec(fc2b), m_dev=m2)
        if fp8:                                                 # the bf16 GEMM inputs are not needed again: their fp8 transposes are
            n1 = n2 = g = None
        return (res1, rstd1, n1, qkv, rq, rk, att, lse, b1, res2, rstd2, n2, u, g, b2, rs1, rs2, q8)

    @staticmethod
    def forward(ctx, x0, rowscale, meta, *params):
        """meta["checkpoint_num"] = n: the first n blocks keep only their outputs (res2, b2: 6 of the ~42 bytes per token and channel a
        block saves) and are recomputed in backward from the previous block's outputs -- `use_checkpoint` / `checkpoint_num` of the
        reference (P:323-327, `with_cp`: torch.utils.checkpoint around Block.forward).  DropPath's per-sample scales are an input
        (`rowscale`), so the recomputation is bit-identical to the first pass."""
        L, eps, taps = meta["L"], meta["eps"], meta["taps"]
        depth = len(params) // NBP
        n_cp = min(int(meta.get("checkpoint_num", 0) or 0), depth)
        saved: List[tuple] = []
        x0_dtype, x0_needs_grad = x0.dtype, x0.requires_grad
        res_bf16 = bool(meta.get("res_bf16"))
        tap_dtype = BF16 if (res_bf16 and meta.get("taps_bf16")) else x0_dtype      # taps_bf16: the consumers read the stream's own rows
        if res_bf16 and x0.dtype != BF16:                      # meta["res_bf16"]: the stream between the blocks is bf16 (the reference's
            x0 = x0.to(BF16)                                   # own bf16 recipe, P:283-286); taps leave the stack in the caller's type
        res, branch, g_prev, rs_prev = x0, None, None, None
        outs = {}
        if meta.get("fp8_hist") is not None:                   # delayed scaling: last step's amax values join the window
            meta["fp8_hist"].roll()
        plan = None
        if rowscale is not None and dp_skip_applies(meta, x0.shape[0], x0.shape[1], params):
            plan = ops.droppath_plan(rowscale, L)              # keep maps + kept counts of every (block, branch), on the device
            if meta.get("dp_count_acc") is not None:           # measurement aid (bench.py): kept counts summed over the steps, on the device
                meta["dp_count_acc"].add_(plan[1])
        for i in range(depth):
            prm = params[i * NBP:(i + 1) * NBP]
            st = BlockStackFn._block_forward(res, branch, g_prev, rs_prev, prm, rowscale, i, meta, plan)
            if branch is not None and (i - 1) in taps:
                outs[i - 1] = st[0] if st[0].dtype == tap_dtype else st[0].to(tap_dtype)
            ls2 = prm[12]
            res, branch, g_prev, rs_prev = st[9], st[14], (vec(ls2) if ls2 is not None else None), st[16]
            if i < n_cp:                                       # keep (res2, b2, rs1, rs2) only; slots as in the full tuple
                st = (None,) * 9 + (st[9],) + (None,) * 4 + (st[14], st[15], st[16], None)
            saved.append(st)
        final, _, _ = ops.rmsnorm_add_fwd(res, branch, g_prev, rs_prev, L, None, eps,       # x = x + residual (P:685-688)
                                          branch_slot=(plan[0][depth - 1, 1] if plan is not None else None))
        outs[depth - 1] = final if final.dtype == tap_dtype else final.to(tap_dtype)
        _mark("enc_fwd_end")
        ctx.x0_dtype = x0_dtype
        ctx.saved = saved
        ctx.params = params
        ctx.meta = meta
        ctx.x0, ctx.rowscale, ctx.n_cp = (x0 if n_cp > 0 else None), rowscale, n_cp
        ctx.plan = plan
        ctx.has_x0_grad = x0_needs_grad
        return tuple(outs[t] for t in taps)

    @staticmethod
    def backward(ctx, *dtaps):
        meta, params, saved = ctx.meta, ctx.params, ctx.saved
        B, L, H, act, taps = meta["B"], meta["L"], meta["H"], _act_d(meta["act"]), meta["taps"]
        depth = len(params) // NBP
        hook = meta.get("grad_ready_hook")
        tapgrad = {t: g for t, g in zip(taps, dtaps) if g is not None}
        grads: List[Optional[torch.Tensor]] = [None] * len(params)
        M = B * L
        D = saved[0][9].shape[1]
        RT = BF16 if meta.get("res_bf16") else F32             # type of the residual stream and of its gradient
        # final add:  T_last = res2 + rs2 * ls2 * b2
        dres = tapgrad.get(depth - 1)
        if dres is None:
            dres = torch.zeros((M, D), dtype=RT, device=saved[0][9].device)
        elif dres.dtype != RT:
            dres = dres.reshape(M, D).to(RT, memory_format=torch.contiguous_format)     # (a copy: the incoming gradient is never edited in place)
        else:
            dres = dres.reshape(M, D).clone(memory_format=torch.contiguous_format)   # updated in place below
        db2 = dg2 = dbias2 = None
        fp8 = bool(meta.get("fp8"))
        plan = ctx.plan                                         # DropPath skipping: keep maps / kept counts (see _block_forward)
        slot, cnt = plan if plan is not None else (None, None)

        def lin_bwd(dy, w, x, q8, name, dact=None, m_dev=None):
            """-> (dx = dy W [* gelu'], gradient of w to hand to autograd | None, bias-gradient partials of the bf16 dgrad epilogue | None)"""
            if not fp8:
                if dact is not None:
                    dx, cs = ops.gemm(dy, mat(w), a_kc=True, b_kc=False, dact_in=dact, act=act, want_colsum=True, m_dev=m_dev)
                else:
                    dx, cs = ops.gemm(dy, mat(w), a_kc=True, b_kc=False, m_dev=m_dev), None
                return dx, _wgrad_defer(dy, x, w, m_dev), cs
            hist = meta.get("fp8_hist")                                  # one quantisation feeds dgrad (plain) and wgrad (transposed copy)
            dyq, dyqt, sd = hist.quantize((i, "d:" + name), dy, True) if hist is not None else ops.fp8_quantize(dy, want_transposed=True)
            _, wqt, _, sw = fp8_weight(w, bool(meta.get("fp8_wchan")))
            dx = ops.gemm_fp8(dyq, wqt, sd, sw, k=dy.shape[1], dact_in=dact, act=(act if dact is not None else None))
            xqt, sx = q8[name]
            mg = getattr(w, "main_grad", None)
            if mg is None and not w.requires_grad:                        # frozen weight: autograd would discard the product
                return dx, None, None
            out = mg.view(w.shape[0], -1) if (mg is not None and mg.dtype in (BF16, F32)) else None
            gw = ops.gemm_fp8(dyqt, xqt, sd, sx, out=out, out_fp32=(out is not None and out.dtype == F32))
            return dx, (None if out is not None else _ret_grad(w, gw)), None

        pending_hooks: List[int] = []
        resolve_end_pending()                                                   # engine mode: the text / fusion tower's queued weight gradients
        _wgrad_flush(force=True)                                                # the decoders' weight gradients queued so far
        _mark("enc_bwd_begin")
        _DEFER_DROPIN[0] = True                                                 # drop-in mode: weight gradients grouped too, resolved below
        try:
            for i in range(depth - 1, -1, -1):
                (n1w, qkvw, qnw, knw, projw, projb, ls1, n2w, fc1w, fc1b, fc2w, fc2b, ls2) = params[i * NBP:(i + 1) * NBP]
                if i < ctx.n_cp:                                    # recompute this block from the previous block's outputs
                    if i == 0:
                        rin, bin_, gin, rsin = ctx.x0, None, None, None
                    else:
                        pls = params[(i - 1) * NBP + 12]
                        rin, bin_, gin, rsin = saved[i - 1][9], saved[i - 1][14], (vec(pls) if pls is not None else None), saved[i - 1][16]
                    with torch.no_grad():
                        saved[i] = BlockStackFn._block_forward(rin, bin_, gin, rsin, params[i * NBP:(i + 1) * NBP], ctx.rowscale, i, meta, plan)
                (res1, rstd1, n1, qkv, rq, rk, att, lse, b1, res2, rstd2, n2, u, g, b2, rs1, rs2, q8) = saved[i]
                base = i * NBP
                if plan is not None:
                    s1, s2 = slot[i, 0], slot[i, 1]                                 # keep maps of this block's attention / MLP branch
                    nb1, m1, m2 = cnt[i, 0, 0:1], cnt[i, 0, 1:2], cnt[i, 1, 1:2]
                else:
                    s1 = s2 = nb1 = m1 = m2 = None
                if i == depth - 1:
                    # backward of the final add (no norm output)
                    _, db2, _, dg2, dbias2 = ops.rmsnorm_add_bwd(None, dres, None, None, None, b2, vec(ls2) if ls2 is not None else None, rs2, L,
                                                                 dg_out=_mg(ls2), want_dbias=True, db_out=_mg(fc2b), branch_slot=s2)
                if ls2 is not None:
                    grads[base + 12] = _ret_grad(ls2, _vgrad(ls2, dg2))
                # ---- MLP branch
                du, grads[base + 10], du_cs = lin_bwd(db2, fc2w, g, q8, "g", dact=u, m_dev=m2)  # weight gradients (bf16 path): queued, launched in groups
                grads[base + 11] = _ret_grad(fc2b, _vgrad(fc2b, dbias2))          # column sum of db2: by-product of the residual backward
                dn2, grads[base + 8], _ = lin_bwd(du, fc1w, n2, q8, "n2", m_dev=m2)
                if du_cs is not None:                                               # fc1 bias gradient: by-product of the dgrad epilogue
                    grads[base + 9] = _ret_grad(fc1b, _vgrad(fc1b, ops.colsum_finish(du_cs, ops._f32_vec(_mg(fc1b), du_cs.shape[1]), m_dev=m2)))
                elif plan is not None:
                    raise ops.InternVideoHipError("DropPath skipping: the fc2 dgrad did not run on the 256x256 kernel (no bias-gradient partials)")
                else:
                    grads[base + 9] = _ret_grad(fc1b, _vgrad(fc1b, ops.colsum_bf16(du, out=_mg(fc1b))))
                del du
                dres, db1, dw2n, dg1, dbias1 = ops.rmsnorm_add_bwd(dn2, dres, res2, rstd2, vec(n2w), b1, vec(ls1) if ls1 is not None else None, rs1, L,
                                                                   dw_out=_mg(n2w), dg_out=_mg(ls1), want_dbias=True, db_out=_mg(projb),
                                                                   y_slot=s2, branch_slot=s1)
                grads[base + 7] = _ret_grad(n2w, _vgrad(n2w, dw2n))
                if ls1 is not None:
                    grads[base + 6] = _ret_grad(ls1, _vgrad(ls1, dg1))
                # ---- attention branch
                datt, grads[base + 4], _ = lin_bwd(db1, projw, att, q8, "att", m_dev=m1)
                grads[base + 5] = _ret_grad(projb, _vgrad(projb, dbias1))
                dqkv = ops.flash_attn_bwd_packed(qkv, att, datt, lse, B, L, H, nb_dev=nb1)
                dwq, dwk = ops.qk_rmsnorm_bwd(qkv, dqkv, vec(qnw), vec(knw), rq, rk, dwq_out=_mg(qnw), dwk_out=_mg(knw), m_dev=m1)
                grads[base + 2] = _ret_grad(qnw, _vgrad(qnw, dwq))
                grads[base + 3] = _ret_grad(knw, _vgrad(knw, dwk))
                dn1, grads[base + 1], _ = lin_bwd(dqkv, qkvw, n1, q8, "n1", m_dev=m1)
                del dqkv
                # res1 of block i is the tap T_{i-1}
                dtap = None
                if i > 0 and (i - 1) in tapgrad:
                    if RT == F32:
                        ops.accum_rows(dres, tapgrad[i - 1].reshape(M, D).contiguous(), B, L, 0, True)
                    elif _TAP_ON_LOAD and tapgrad[i - 1].dtype == BF16 and tapgrad[i - 1].is_contiguous():
                        dtap = tapgrad[i - 1].reshape(M, D)     # bf16 stream: joins dres inside the norm backward's loads (fp32 sum, one rounding)
                    else:
                        dres.add_(tapgrad[i - 1].reshape(M, D))
                if i > 0:
                    pls2 = params[(i - 1) * NBP + 12]
                    prs2 = saved[i - 1][16]
                    pb2 = saved[i - 1][14]
                    pfc2b = params[(i - 1) * NBP + 11]
                    dres, db2n, dw1n, dg2n, dbias2n = ops.rmsnorm_add_bwd(dn1, dres, res1, rstd1, vec(n1w), pb2,
                                                                          vec(pls2) if pls2 is not None else None, prs2, L,
                                                                          dw_out=_mg(n1w), dg_out=_mg(pls2), want_dbias=True, db_out=_mg(pfc2b),
                                                                          dres_extra=dtap, y_slot=s1,
                                                                          branch_slot=(slot[i - 1, 1] if plan is not None else None))
                    db2, dg2, dbias2 = db2n, dg2n, dbias2n
                else:
                    dres, _, dw1n, _ = ops.rmsnorm_add_bwd(dn1, dres, res1, rstd1, vec(n1w), None, None, None, L, want_dbranch=False,
                                                           dw_out=_mg(n1w), y_slot=s1)
                grads[base + 0] = _ret_grad(n1w, _vgrad(n1w, dw1n))
                saved[i] = None                                                     # free this block's activations
                pending_hooks.append(i)
                _wgrad_flush(force=(i == 0))                                        # whenever the queued tiles fill the CUs, and at the end
                if not _wgrad_queue:                                                # the gradients of every block seen so far are final
                    if hook is not None:
                        for j in pending_hooks:
                            hook(j)
                    pending_hooks.clear()
        finally:
            _DEFER_DROPIN[0] = False
        grads = [g.resolve() if isinstance(g, _PendingGrad) else g for g in grads]   # the forced flush at block 0 has filled every buffer
        ctx.saved = None
        ctx.x0 = None
        if ctx.has_x0_grad and dres.dtype != ctx.x0_dtype:
            dres = dres.to(ctx.x0_dtype)
        return (dres if ctx.has_x0_grad else None, None, None, *grads)


def _decoder_tail_fwd(y, nw, nb, eps, target, norm_none: bool):
    """decoder tail on bf16 rows y [M, C]: LayerNorm -> l2 (norm_type 'l2', P:358-359) or LayerNorm only ('none', P:360-361).
    target None -> (features bf16 [M, C], stats, None); else -> (sum_rows(2 - 2 <s, t>) as a 1-element fp32 tensor, stats, ds | None):
    with 'l2' the features never reach HBM (ln_l2 kernel), with 'none' they are materialised and the loss rows come from cosine_rows."""
    if not norm_none:
        if target is None:
            out, stats, _ = ops.ln_l2_fwd(y, vec(nw), vec(nb), eps)
            return out, stats, None
        _, stats, rows = ops.ln_l2_fwd(y, vec(nw), vec(nb), eps, want_out=False, target=target.reshape(-1, target.shape[-1]))
        return ops.sum_rows(rows, 1.0), stats, None
    out, _, stats = ops.layernorm_fwd(y, vec(nw), vec(nb), eps)
    if target is None:
        return out, stats, None
    tg = target.reshape(-1, target.shape[-1]).contiguous()
    rows, ds = ops.cosine_rows(out, tg if tg.dtype in (BF16, F32) else tg.float(), dscale=1.0, want_grad=True)
    return ops.sum_rows(rows, 1.0), stats, ds


def _decoder_tail_bwd(y, nw, nb, stats, dout, target, norm_none: bool, ds):
    """-> (dy bf16 [M, C], dnw fp32, dnb fp32); the norm's column sums land in its parameters' main_grad directly when the engine provides one"""
    if not norm_none:
        if target is None:
            do = dout.reshape(-1, dout.shape[-1]).contiguous()
            if do.dtype not in (BF16, F32):
                do = do.float()
            return ops.ln_l2_bwd(y, vec(nw), vec(nb), stats, do, None, 0.0, dw_out=_mg(nw), db_out=_mg(nb))
        # d(sum_rows(2 - 2 <s,t>)) = -2 t per row, times the upstream scalar (read on the device: no host sync)
        return ops.ln_l2_bwd(y, vec(nw), vec(nb), stats, None, target.reshape(-1, target.shape[-1]), -2.0,
                             dscale_dev=dout.reshape(1).float().contiguous(), dw_out=_mg(nw), db_out=_mg(nb))
    if target is None:
        do = dout.reshape(-1, dout.shape[-1]).contiguous()
        do = do if do.dtype == BF16 else do.to(BF16)
    else:
        do = (ds.float() * dout.reshape(1).float()).to(BF16)
    dx, dw, db, _, _ = ops.layernorm_bwd(y, vec(nw), stats, do)
    return dx.to(BF16), dw, db


class PosDecoderFn(torch.autograd.Function):
    """Decoder input (tap + pos_embed[~mask], P:713-714 / P:736-737) -> Linear_Decoder (P:334-365) or MLP_Decoder
    (P:368-403) -> LayerNorm -> l2.  `skip` = 1 drops the cls row (MAE branch, P:681)."""

    @staticmethod
    def forward(ctx, tap, pos, vis_idx, inv_idx, skip, ln_eps, mlp, target, *p):
        """target None -> returns the l2-normalised features (drop-in forward, P:744).
        target (bf16|fp32 [B, L-skip, C], l2-normalised teacher features) -> returns sum_rows(2 - 2 <s, t>) as a 1-element
        fp32 tensor: the distillation loss term of engines/engine_for_pretraining.py:131-148 before the mean, without
        ever writing the (B, L, C) student features to HBM."""
        B, L = vis_idx.shape
        D = tap.shape[-1]
        mlp, norm_none = bool(int(mlp) & 1), bool(int(mlp) & 2)                 # bit 0: MLP_Decoder, bit 1: norm_type 'none' (P:358-363)
        posv = vec(pos).reshape(-1, D)
        xin = ops.add_pos_gather(tap, posv, vis_idx, skip)                      # bf16 [B*(L-skip), D]
        if mlp:
            w0, b0, w2, b2, nw, nb = p
            h, u = ops.gemm(xin, mat(w0), bias=vec(b0), act="gelu_erf_d", want_preact=True)
            y = ops.gemm(h, mat(w2), bias=vec(b2))
        else:
            w0, b0, nw, nb = p
            h = u = None
            y = ops.gemm(xin, mat(w0), bias=vec(b0))
        ret, stats, ds = _decoder_tail_fwd(y, nw, nb, ln_eps, target, norm_none)
        if target is None:
            ret = ret.reshape(B, L - skip, -1)
        ctx.save_for_backward(xin, h, u, y, stats, vis_idx, inv_idx, target, ds)
        ctx.p, ctx.pos = p, pos
        ctx.meta = (B, L, D, skip, mlp, norm_none)
        ctx.tap_dtype = tap.dtype
        return ret

    @staticmethod
    def backward(ctx, dout):
        xin, h, u, y, stats, vis_idx, inv_idx, target, ds = ctx.saved_tensors
        B, L, D, skip, mlp, norm_none = ctx.meta
        p = ctx.p
        nw, nb = p[-2], p[-1]
        dy, dnw, dnb = _decoder_tail_bwd(y, nw, nb, stats, dout, target, norm_none, ds)
        if mlp:
            w0, b0, w2, b2 = p[:4]
            du = ops.gemm(dy, mat(w2), a_kc=True, b_kc=False, dact_in=u, act="gelu_erf_d")
            gw2 = _wgrad_defer(dy, h, w2); gb2 = _ret_grad(b2, _vgrad(b2, ops.colsum_bf16(dy, out=_mg(b2))))
            dxin = ops.gemm(du, mat(w0), a_kc=True, b_kc=False)
            gw0 = _wgrad_defer(du, xin, w0); gb0 = _ret_grad(b0, _vgrad(b0, ops.colsum_bf16(du, out=_mg(b0))))
            pg = (gw0, gb0, gw2, gb2)
        else:
            w0, b0 = p[:2]
            dxin = ops.gemm(dy, mat(w0), a_kc=True, b_kc=False)
            pg = (_wgrad_defer(dy, xin, w0), _ret_grad(b0, _vgrad(b0, ops.colsum_bf16(dy, out=_mg(b0)))))
        if ctx.tap_dtype == BF16:                                # a bf16 tap: its gradient is the dgrad's output itself (behind `skip` zero rows)
            dtap = dxin if skip == 0 else ops.rows_shift_bf16(dxin, B, L, skip)
        else:
            dtap = torch.empty((B * L, D), dtype=F32, device=dxin.device)
            ops.accum_rows(dtap, dxin, B, L, skip, False)
        pos = ctx.pos
        mg = getattr(pos, "main_grad", None)
        if mg is not None:          # K decoders share one table: accumulate in place (the engine zeroes it every step)
            ops.pos_grad(dxin, 1, B, L - skip, inv_idx, skip, dpos=mg.view(-1, D), accumulate=True)
            gpos = None
        else:
            gpos = _ret_grad(pos, ops.pos_grad(dxin, 1, B, L - skip, inv_idx, skip))
        return (dtap, gpos, None, None, None, None, None, None, *pg,
                _ret_grad(nw, _vgrad(nw, dnw)), _ret_grad(nb, _vgrad(nb, dnb)))


class StreamToBf16Fn(torch.autograd.Function):
    """fp32 residual-stream rows [B*L, D] -> bf16 (B, L, D): the `x_vis` output of the stage-2 vision encoder
    (multi_modality/models/backbones/internvideo2/internvideo2.py:640-647), which the reference returns in the model dtype."""

    @staticmethod
    def forward(ctx, x, B, L):
        ctx.meta = (B, L, x.shape[-1])
        return ops.rows_to_bf16(x.contiguous(), B, L, 0).view(B, L, x.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        B, L, D = ctx.meta
        dy2 = dy.reshape(B * L, D).contiguous()
        if dy2.dtype not in (BF16, F32):
            dy2 = dy2.float()
        dx = torch.empty((B * L, D), dtype=F32, device=dy.device)
        ops.accum_rows(dx, dy2, B, L, 0, False)
        return dx, None, None


class LnL2Fn(torch.autograd.Function):
    """LayerNorm -> l2 (tail of Linear_Decoder; `norm_none`: LayerNorm only, norm_type 'none') on bf16 rows; with `target` returns
    sum_rows(2 - 2 <s, t>) instead."""

    @staticmethod
    def forward(ctx, y, nw, nb, eps, target, norm_none=False):
        y2 = y.reshape(-1, y.shape[-1]).contiguous()
        ret, stats, ds = _decoder_tail_fwd(y2, nw, nb, eps, target, bool(norm_none))
        if target is None:
            ret = ret.reshape(y.shape)
        ctx.save_for_backward(y2, stats, target, ds)
        ctx.p, ctx.yshape, ctx.norm_none = (nw, nb), y.shape, bool(norm_none)
        return ret

    @staticmethod
    def backward(ctx, dout):
        y2, stats, target, ds = ctx.saved_tensors
        nw, nb = ctx.p
        dy, dnw, dnb = _decoder_tail_bwd(y2, nw, nb, stats, dout, target, ctx.norm_none, ds)
        return dy.reshape(ctx.yshape), _ret_grad(nw, _vgrad(nw, dnw)), _ret_grad(nb, _vgrad(nb, dnb)), None, None, None


class AttnPoolFn(torch.autograd.Function):
    """AttentionPoolingBlock / CrossAttention (P:18-114): mean query, three LayerNorms, q/k/v Linear with separate
    bias parameters, 1-query multi-head attention (the flash kernel with Lq = 1), output projection."""

    @staticmethod
    def forward(ctx, x, B, L, H, ln_eps, nqw, nqb, nkw, nkb, nvw, nvb, qw, qb, kw, kb, vw, vb, pw, pb):
        D = x.shape[-1]
        hd = D // H
        ctx.xdtype = x.dtype
        if x.dtype != F32:                                                      # a bf16 tap of the residual stream: this block works on fp32 rows
            x = x.float()
        xm = ops.token_mean_fwd(x, B, L)                                        # [B, D] fp32
        qin, _, qstats = ops.layernorm_fwd(xm, vec(nqw), vec(nqb), ln_eps)
        kin, vin, kvstats = ops.layernorm_fwd(x, vec(nkw), vec(nkb), ln_eps, vec(nvw), vec(nvb))
        q = ops.gemm(qin, mat(qw), bias=vec(qb))                                # [B, D]
        kv = torch.empty((2, B * L, D), dtype=BF16, device=x.device)
        ops.gemm(kin, mat(kw), bias=vec(kb), out=kv[0])
        ops.gemm(vin, mat(vw), bias=vec(vb), out=kv[1])
        q4 = q.view(B, 1, H, hd)
        k4, v4 = kv[0].view(B, L, H, hd), kv[1].view(B, L, H, hd)
        if hd <= 128:
            o, lse = ops.flash_attn_fwd(q4, k4, v4)                             # scale hd^-0.5 (P:30,70)
        else:                                                                   # 6B: 16 heads over 3200 -> hd = 200
            o, lse = ops.pool_attn_fwd(q.view(B, H, hd), k4, v4)
        o2 = o.view(B, D)
        y = ops.gemm(o2, mat(pw), bias=vec(pb))
        ctx.save_for_backward(x, xm, qstats, kvstats, qin, kin, vin, q, kv, o, lse)
        ctx.p = (nqw, nqb, nkw, nkb, nvw, nvb, qw, qb, kw, kb, vw, vb, pw, pb)
        ctx.meta = (B, L, H, hd, D)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, xm, qstats, kvstats, qin, kin, vin, q, kv, o, lse = ctx.saved_tensors
        nqw, nqb, nkw, nkb, nvw, nvb, qw, qb, kw, kb, vw, vb, pw, pb = ctx.p
        B, L, H, hd, D = ctx.meta
        dy2 = dy.contiguous().to(BF16)
        o2 = o.view(B, D)
        do = ops.gemm(dy2, mat(pw), a_kc=True, b_kc=False)
        gpw = _ret_grad(pw, _wgrad(dy2, o2, pw)); gpb = _ret_grad(pb, _vgrad(pb, ops.colsum_bf16(dy2)))
        q4 = q.view(B, 1, H, hd)
        k4, v4 = kv[0].view(B, L, H, hd), kv[1].view(B, L, H, hd)
        if hd <= 128:
            dq4, dkv = ops.flash_attn_bwd(q4, k4, v4, o, do.view(B, 1, H, hd), lse)
        else:
            dq4, dkv = ops.pool_attn_bwd(q.view(B, H, hd), k4, v4, do.view(B, H, hd), lse)
        dq = dq4.view(B, D)
        dk, dv = dkv[0].view(B * L, D), dkv[1].view(B * L, D)
        dqin = ops.gemm(dq, mat(qw), a_kc=True, b_kc=False)
        gqw = _ret_grad(qw, _wgrad(dq, qin, qw)); gqb = _ret_grad(qb, _vgrad(qb, ops.colsum_bf16(dq)))
        dkin = ops.gemm(dk, mat(kw), a_kc=True, b_kc=False)
        gkw = _wgrad_defer(dk, kin, kw); gkb = _ret_grad(kb, _vgrad(kb, ops.colsum_bf16(dk)))      # long K, 36 tiles: joins the decoders' grouped launch
        dvin = ops.gemm(dv, mat(vw), a_kc=True, b_kc=False)
        gvw = _wgrad_defer(dv, vin, vw); gvb = _ret_grad(vb, _vgrad(vb, ops.colsum_bf16(dv)))
        dx, dnkw, dnkb, dnvw, dnvb = ops.layernorm_bwd(x, vec(nkw), kvstats, dkin, vec(nvw), dvin)
        dxm, dnqw, dnqb, _, _ = ops.layernorm_bwd(xm, vec(nqw), qstats, dqin)
        ops.token_mean_bwd(dxm, dx, B, L)
        if dx.dtype != ctx.xdtype:
            dx = dx.to(ctx.xdtype)
        return (dx, None, None, None, None,
                _ret_grad(nqw, _vgrad(nqw, dnqw)), _ret_grad(nqb, _vgrad(nqb, dnqb)),
                _ret_grad(nkw, _vgrad(nkw, dnkw)), _ret_grad(nkb, _vgrad(nkb, dnkb)),
                _ret_grad(nvw, _vgrad(nvw, dnvw)), _ret_grad(nvb, _vgrad(nvb, dnvb)),
                gqw, gqb, gkw, gkb, gvw, gvb, gpw, gpb)


# ---------------------------------------------------------------------------------------------------------------------------
# forward-only (teacher / evaluation) versions: nothing is saved, every intermediate is released as soon as it is consumed
# ---------------------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def embed_all_tokens(video, proj_w, proj_b, cls_token, pos_embed, tubelet: int, patch: int, per_frame: bool = False):
    """Tubelet patch embed of EVERY token + cls + positional table -> fp32 residual-stream rows.
    per_frame=False: (B, 1 + T*h*w) sequences with one cls per clip (the student without a mask).
    per_frame=True : (B*T', 1 + h*w) sequences, one per frame, each with its own cls row and the per-frame table
                     (the CLIP teacher, internvl_clip_vision.py:415-421).  -> (x0 [S*L, D], S, L)"""
    B, Cc, T, Hh, Ww = video.shape
    D = proj_w.shape[0]
    Tn, gh, gw = T // tubelet, Hh // patch, Ww // patch
    N1 = 1 + Tn * gh * gw
    idx = torch.arange(N1, dtype=torch.int32, device=video.device).unsqueeze(0).expand(B, N1).contiguous()
    kreal = proj_w[0].numel()
    kp = (kreal + 63) // 64 * 64
    wp = torch.zeros((D, kp), dtype=BF16, device=video.device)
    wp[:, :kreal] = mat(proj_w).reshape(D, kreal)
    cols = ops.patch_im2col(video, idx, tubelet, patch, kp)                     # rows in (b, t, h, w) order
    tok = ops.gemm(cols, wp, bias=vec(proj_b))
    del cols
    if per_frame:
        S, L = B * Tn, 1 + gh * gw
        idx = torch.arange(L, dtype=torch.int32, device=video.device).unsqueeze(0).expand(S, L).contiguous()
    else:
        S, L = B, N1
    if pos_embed.shape[-2] != L:
        raise ValueError(f"positional table has {pos_embed.shape[-2]} rows, the sequences have {L} tokens")
    x0 = ops.assemble_tokens(tok, vec(cls_token).reshape(-1), vec(pos_embed).reshape(-1, D), idx)
    return x0, S, L


@torch.no_grad()
def frozen_fp8_weight(w: torch.Tensor):
    """(wq [N, K] e4m3, one scale per output feature) of a FROZEN weight: quantised once (per-channel scales cost nothing at inference)
    and cached on the parameter for as long as its storage and version stand"""
    key = (w.data_ptr(), w._version)
    c = getattr(w, "_ivh_fp8_frozen", None)
    if c is None or c[0] != key:
        wb = mat(w)
        q, _, sr, _ = ops.fp8_quantize_weight(wb.reshape(wb.shape[0], -1))
        c = (key, q, sr)
        w._ivh_fp8_frozen = c
    return c[1], c[2]


@torch.no_grad()
def block_stack_infer(x0, block_params: Sequence, S: int, L: int, H: int, eps: float, act: str, taps: Sequence[int], fp8: bool = False):
    """The fused-residual block loop of BlockStackFn.forward without the autograd bookkeeping.  block_params: per block the 13
    tensors of Block.flat_params().  -> {tap index: fp32 [S*L, D] residual-stream value after that block}; the last block is
    always tapped.  fp8 (opt-in, frozen teachers only): the four GEMMs of every block on the e4m3 MFMA path -- activations quantised per
    tensor on the fly, weights once with per-channel scales (frozen_fp8_weight); norms, attention and the residual stream unchanged."""
    depth = len(block_params)
    want = set(taps) | {depth - 1}
    outs = {}
    res, branch, g_prev = x0, None, None
    def lin(x, w, bias=None, act=None):
        if not fp8:
            return ops.gemm(x, mat(w), bias=bias, act=act)
        xq, _, sx = ops.fp8_quantize(x)
        wq, sw = frozen_fp8_weight(w)
        return ops.gemm_fp8(xq, wq, sx, sw, bias=bias, act=act)

    for i, prm in enumerate(block_params):
        (n1w, qkvw, qnw, knw, projw, projb, ls1, n2w, fc1w, fc1b, fc2w, fc2b, ls2) = prm
        if branch is None:
            res1 = res
            _, n1, _ = ops.rmsnorm_add_fwd(res, None, None, None, L, vec(n1w), eps, want_res_out=False)
        else:
            res1, n1, _ = ops.rmsnorm_add_fwd(res, branch, g_prev, None, L, vec(n1w), eps)
            if (i - 1) in want:
                outs[i - 1] = res1
        qkv = lin(n1, qkvw)
        del n1
        ops.qk_rmsnorm_fwd(qkv, vec(qnw), vec(knw), eps)
        att, _ = ops.flash_attn_fwd_packed(qkv, S, L, H)
        del qkv
        b1 = lin(att, projw, bias=vec(projb))
        del att
        res2, n2, _ = ops.rmsnorm_add_fwd(res1, b1, vec(ls1) if ls1 is not None else None, None, L, vec(n2w), eps)
        del b1
        g = lin(n2, fc1w, bias=vec(fc1b), act=act)
        del n2
        branch = lin(g, fc2w, bias=vec(fc2b))
        del g
        res, g_prev = res2, (vec(ls2) if ls2 is not None else None)
    final, _, _ = ops.rmsnorm_add_fwd(res, branch, g_prev, None, L, None, eps)
    outs[depth - 1] = final
    return outs


@torch.no_grad()
def attn_pool_infer(x, S: int, L: int, H: int, ln_eps: float, nqw, nqb, nkw, nkb, nvw, nvb, qw, qb, kw, kb, vw, vb, pw, pb,
                    want_attn: bool = False):
    """AttnPoolFn.forward without autograd; want_attn adds the head-averaged attention over the patch keys (fp32 [S, L-1])."""
    D = x.shape[-1]
    hd = D // H
    xm = ops.token_mean_fwd(x, S, L)
    qin, _, _ = ops.layernorm_fwd(xm, vec(nqw), vec(nqb), ln_eps)
    kin, vin, _ = ops.layernorm_fwd(x, vec(nkw), vec(nkb), ln_eps, vec(nvw), vec(nvb))
    q = ops.gemm(qin, mat(qw), bias=vec(qb))
    kv = torch.empty((2, S * L, D), dtype=BF16, device=x.device)
    ops.gemm(kin, mat(kw), bias=vec(kb), out=kv[0])
    ops.gemm(vin, mat(vw), bias=vec(vb), out=kv[1])
    del kin, vin
    k4, v4 = kv[0].view(S, L, H, hd), kv[1].view(S, L, H, hd)
    if hd <= 128:
        o, _ = ops.flash_attn_fwd(q.view(S, 1, H, hd), k4, v4)
    else:
        o, _ = ops.pool_attn_fwd(q.view(S, H, hd), k4, v4)
    y = ops.gemm(o.view(S, D), mat(pw), bias=vec(pb))
    attn = ops.pool_attn_map(q.view(S, H, hd), k4, skip=1) if want_attn else None
    return y, attn


# ---------------------------------------------------------------------------------------------------------------------------
# LayerNorm pre-norm blocks (VideoMAE encoder / decoder, InternVideo1/Pretrain/VideoMAE/modeling_finetune.py:75-181 "MF:";
# the VideoMAE teacher of InternVideo2, single_modality/models/videomae.py:60-132)
# ---------------------------------------------------------------------------------------------------------------------------
LN_BLOCK_PARAM_NAMES = ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.q_bias", "attn.v_bias", "attn.proj.weight",
                        "attn.proj.bias", "gamma_1", "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias",
                        "mlp.fc2.weight", "mlp.fc2.bias", "gamma_2")
NLP = len(LN_BLOCK_PARAM_NAMES)


def _qkv_bias(qb, vb):
    """MF:108-113: `cat(q_bias, zeros_like(v_bias), v_bias)` (k has no bias); None when the block has no qkv bias"""
    if qb is None:
        return None
    q, v = vec(qb), vec(vb)
    return torch.cat((q, torch.zeros_like(v), v)).contiguous()


def _mlp_operands(fc1w, fc1b, fc2w):
    """bf16 fc1 / fc2 operands with the hidden width padded to a multiple of 8 when it is not one (VideoMAE-giant's decoder:
    int(512 * 48/11) = 2234, MP:500-518): the MFMA GEMMs move 16-byte row chunks.  Padded fc1 rows / bias entries and fc2 columns are
    zero, so the padded hidden units are gelu(0) = 0 and contribute nothing; gradients are sliced back.  -> (w1, b1, w2, Hm)"""
    Hm = fc1w.shape[0]
    w1, b1, w2 = mat(fc1w), vec(fc1b), mat(fc2w)
    if Hm % 8 == 0:
        return w1, b1, w2, Hm
    Hp = (Hm + 7) // 8 * 8
    w1p = torch.zeros((Hp, w1.shape[1]), dtype=BF16, device=w1.device); w1p[:Hm] = w1
    b1p = torch.zeros((Hp,), dtype=F32, device=w1.device); b1p[:Hm] = b1
    w2p = torch.zeros((w2.shape[0], Hp), dtype=BF16, device=w1.device); w2p[:, :Hm] = w2
    return w1p, b1p, w2p, Hm


class LNBlockStackFn(torch.autograd.Function):
    """depth x [ x += dp(gamma_1 * attn(LN(x))) ; x += dp(gamma_2 * mlp(LN(x))) ]  (MF:170-181) on an fp32 token stream [B*L, D]:
    LayerNorm -> qkv GEMM (+ q / v bias) -> flash attention -> proj GEMM -> residual-add kernel (LayerScale, DropPath fused) ->
    LayerNorm -> fc1 (+ bias, erf-GELU, gelu' by-product) -> fc2 -> residual add.  Returns the stream after the last block."""

    @staticmethod
    def forward(ctx, x0, rowscale, meta, *params):
        B, L, H, eps = meta["B"], meta["L"], meta["H"], meta["eps"]
        depth = len(params) // NLP
        saved = []
        res = x0
        for i in range(depth):
            (n1w, n1b, qkvw, qb, vb, projw, projb, g1, n2w, n2b, fc1w, fc1b, fc2w, fc2b, g2) = params[i * NLP:(i + 1) * NLP]
            rs1 = rowscale[i, 0] if rowscale is not None else None
            rs2 = rowscale[i, 1] if rowscale is not None else None
            n1, _, st1 = ops.layernorm_fwd(res, vec(n1w), vec(n1b), eps)
            qkv = ops.gemm(n1, mat(qkvw), bias=_qkv_bias(qb, vb))
            att, lse = ops.flash_attn_fwd_packed(qkv, B, L, H)
            b1 = ops.gemm(att, mat(projw), bias=vec(projb))
            res2, _, _ = ops.rmsnorm_add_fwd(res, b1, vec(g1) if g1 is not None else None, rs1, L, None, eps)
            n2, _, st2 = ops.layernorm_fwd(res2, vec(n2w), vec(n2b), eps)
            w1, bb1, w2, _ = _mlp_operands(fc1w, fc1b, fc2w)
            g, u = ops.gemm(n2, w1, bias=bb1, act="gelu_erf_d", want_preact=True)
            b2 = ops.gemm(g, w2, bias=vec(fc2b))
            res3, _, _ = ops.rmsnorm_add_fwd(res2, b2, vec(g2) if g2 is not None else None, rs2, L, None, eps)
            saved.append((res, st1, n1, qkv, att, lse, b1, res2, st2, n2, u, g, b2, rs1, rs2))
            res = res3
        ctx.saved, ctx.params, ctx.meta = saved, params, meta
        ctx.has_x0_grad = x0.requires_grad
        return res

    @staticmethod
    def backward(ctx, dout):
        meta, params, saved = ctx.meta, ctx.params, ctx.saved
        B, L, H = meta["B"], meta["L"], meta["H"]
        depth = len(params) // NLP
        grads: List[Optional[torch.Tensor]] = [None] * len(params)
        dres = dout.contiguous().float().clone()                              # updated in place below
        for i in range(depth - 1, -1, -1):
            (n1w, n1b, qkvw, qb, vb, projw, projb, g1, n2w, n2b, fc1w, fc1b, fc2w, fc2b, g2) = params[i * NLP:(i + 1) * NLP]
            (res, st1, n1, qkv, att, lse, b1, res2, st2, n2, u, g, b2, rs1, rs2) = saved[i]
            base = i * NLP
            D = res.shape[1]
            # ---- MLP branch: res3 = res2 + rs2 * g2 * b2
            _, db2, _, dg2 = ops.rmsnorm_add_bwd(None, dres, None, None, None, b2, vec(g2) if g2 is not None else None, rs2, L)
            if g2 is not None:
                grads[base + 14] = _ret_grad(g2, dg2)
            w1, _, w2, Hm = _mlp_operands(fc1w, fc1b, fc2w)
            du = ops.gemm(db2, w2, a_kc=True, b_kc=False, dact_in=u, act="gelu_erf_d")
            dn2 = ops.gemm(du, w1, a_kc=True, b_kc=False)
            if w1.shape[0] == Hm:
                grads[base + 12] = _ret_grad(fc2w, _wgrad(db2, g, fc2w))
                grads[base + 10] = _ret_grad(fc1w, _wgrad(du, n2, fc1w))
                grads[base + 11] = _ret_grad(fc1b, ops.colsum_bf16(du))
            else:                                                              # padded hidden width: slice the gradients back
                grads[base + 12] = _ret_grad(fc2w, ops.gemm(db2, g, a_kc=False, b_kc=False)[:, :Hm])
                grads[base + 10] = _ret_grad(fc1w, ops.gemm(du, n2, a_kc=False, b_kc=False)[:Hm])
                grads[base + 11] = _ret_grad(fc1b, ops.colsum_bf16(du)[:Hm])
            grads[base + 13] = _ret_grad(fc2b, ops.colsum_bf16(db2))
            del du
            _, dw2n, db2n, _, _ = ops.layernorm_bwd(res2, vec(n2w), st2, dn2, dx=dres, accumulate=True)
            grads[base + 8], grads[base + 9] = _ret_grad(n2w, dw2n), _ret_grad(n2b, db2n)
            # ---- attention branch: res2 = res + rs1 * g1 * b1
            _, db1, _, dg1 = ops.rmsnorm_add_bwd(None, dres, None, None, None, b1, vec(g1) if g1 is not None else None, rs1, L)
            if g1 is not None:
                grads[base + 7] = _ret_grad(g1, dg1)
            datt = ops.gemm(db1, mat(projw), a_kc=True, b_kc=False)
            grads[base + 5] = _ret_grad(projw, _wgrad(db1, att, projw))
            grads[base + 6] = _ret_grad(projb, ops.colsum_bf16(db1))
            dqkv = ops.flash_attn_bwd_packed(qkv, att, datt, lse, B, L, H)
            if qb is not None:
                dbias = ops.colsum_bf16(dqkv)
                grads[base + 3] = _ret_grad(qb, dbias[:D].clone())
                grads[base + 4] = _ret_grad(vb, dbias[2 * D:].clone())
            dn1 = ops.gemm(dqkv, mat(qkvw), a_kc=True, b_kc=False)
            grads[base + 2] = _ret_grad(qkvw, _wgrad(dqkv, n1, qkvw))
            del dqkv
            _, dw1n, db1n, _, _ = ops.layernorm_bwd(res, vec(n1w), st1, dn1, dx=dres, accumulate=True)
            grads[base + 0], grads[base + 1] = _ret_grad(n1w, dw1n), _ret_grad(n1b, db1n)
            saved[i] = None
        ctx.saved = None
        return (dres if ctx.has_x0_grad else None, None, None, *grads)


@torch.no_grad()
def ln_block_stack_infer(x0, block_params: Sequence, B: int, L: int, H: int, eps: float, taps: Sequence[int] = (), final_norm=None,
                         heads_as_sequence: bool = False):
    """forward-only LN block loop; -> {tap: fp32 stream after that block} (last block always).  final_norm = (w, b, eps): the stream
    after the LAST block is replaced by its LayerNorm before it is tapped (videomae.py:300-303).
    heads_as_sequence: the attention exactly as single_modality/models/videomae.py:91-96 codes it -- q, k, v are handed to
    flash_attn_func as (B, H, N, hd) while its contract is (batch, seqlen, nheads, headdim), so every token attends over its own H
    head slots (seqlen = H, nheads = N) and the (B, H, N, hd) result is reinterpreted as (B, N, H*hd) by the reshape that follows.
    The strided attention entry point runs that layout in place (no transposes)."""
    depth = len(block_params)
    want = set(taps) | {depth - 1}
    outs = {}
    res = x0
    for i, prm in enumerate(block_params):
        (n1w, n1b, qkvw, qb, vb, projw, projb, g1, n2w, n2b, fc1w, fc1b, fc2w, fc2b, g2) = prm
        n1, _, _ = ops.layernorm_fwd(res, vec(n1w), vec(n1b), eps)
        qkv = ops.gemm(n1, mat(qkvw), bias=_qkv_bias(qb, vb))
        del n1
        if heads_as_sequence:
            hd = qkv.shape[1] // (3 * H)
            q5 = qkv.view(B, L, 3, H, hd)
            o, _ = ops.flash_attn_fwd(q5[:, :, 0].permute(0, 2, 1, 3), q5[:, :, 1].permute(0, 2, 1, 3), q5[:, :, 2].permute(0, 2, 1, 3))
            att = o.view(B * L, H * hd)                                       # (B, H, N, hd) memory read as (B, N, H*hd): videomae.py:96
        else:
            att, _ = ops.flash_attn_fwd_packed(qkv, B, L, H)
        del qkv
        b1 = ops.gemm(att, mat(projw), bias=vec(projb))
        del att
        res2, _, _ = ops.rmsnorm_add_fwd(res, b1, vec(g1) if g1 is not None else None, None, L, None, eps)
        n2, _, _ = ops.layernorm_fwd(res2, vec(n2w), vec(n2b), eps)
        w1, bb1, w2, _ = _mlp_operands(fc1w, fc1b, fc2w)
        g = ops.gemm(n2, w1, bias=bb1, act="gelu_erf")
        del n2
        b2 = ops.gemm(g, w2, bias=vec(fc2b))
        del g
        res, _, _ = ops.rmsnorm_add_fwd(res2, b2, vec(g2) if g2 is not None else None, None, L, None, eps)
        if i in want:
            outs[i] = res
    if final_norm is not None:
        w, b, e = final_norm
        y, _, _ = ops.layernorm_fwd(outs[depth - 1], vec(w), vec(b), e)
        outs[depth - 1] = y                                                   # bf16
    return outs


class LayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x) on rows: x fp32|bf16 [M, C] -> bf16 (MP:138 encoder.norm, MP:264-266 decoder.norm)"""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        if x2.dtype not in (F32, BF16):
            x2 = x2.float()
        y, _, stats = ops.layernorm_fwd(x2, vec(w), vec(b), eps)
        ctx.save_for_backward(x2, stats)
        ctx.p, ctx.xshape, ctx.xdtype = (w, b), x.shape, x.dtype
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, stats = ctx.saved_tensors
        w, b = ctx.p
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        if dy2.dtype != BF16:
            dy2 = dy2.to(BF16)
        dx, dw, db, _, _ = ops.layernorm_bwd(x2, vec(w), stats, dy2)
        return dx.reshape(ctx.xshape).to(ctx.xdtype), _ret_grad(w, dw), _ret_grad(b, db), None


class PatchEmbedVisibleFn(torch.autograd.Function):
    """cls-free tubelet patch embed of the visible tokens + fixed positional table (MP:125-133): im2col of the kept cubes -> MFMA
    GEMM -> fp32 stream rows [B*Nvis, D].  vis_idx int32 [B, 1+Nvis] (token id + 1, leading pseudo-cls 0)."""

    @staticmethod
    def forward(ctx, video, vis_idx, proj_w, proj_b, pos, tubelet, patch):
        D = proj_w.shape[0]
        kreal = proj_w[0].numel()
        kp = (kreal + 63) // 64 * 64
        wp = torch.zeros((D, kp), dtype=BF16, device=video.device)
        wp[:, :kreal] = mat(proj_w).reshape(D, kreal)
        cols = ops.patch_im2col(video, vis_idx, tubelet, patch, kp)
        tok = ops.gemm(cols, wp, bias=vec(proj_b))
        x0 = ops.assemble_tokens_nocls(tok, pos, vis_idx)
        ctx.save_for_backward(cols)
        ctx.p, ctx.meta = (proj_w, proj_b), (vis_idx.shape[0], vis_idx.shape[1] - 1, kreal)
        return x0

    @staticmethod
    def backward(ctx, dx0):
        (cols,) = ctx.saved_tensors
        proj_w, proj_b = ctx.p
        B, Nvis, kreal = ctx.meta
        dtok = ops.rows_to_bf16(dx0.contiguous(), B, Nvis, 0)
        dwp = ops.gemm(dtok, cols, a_kc=False, b_kc=False)
        return (None, None, _ret_grad(proj_w, dwp[:, :kreal].reshape(proj_w.shape)), _ret_grad(proj_b, ops.colsum_bf16(dtok)),
                None, None, None)


class MaeDecoderInputFn(torch.autograd.Function):
    """MP:381-389: cat([x_vis + pos[~mask], mask_token + pos[mask]], dim=1) as fp32 stream rows [B*N, Cd]"""

    @staticmethod
    def forward(ctx, xvis, mask_token, pos, vis_idx, msk_idx):
        out = ops.mae_decoder_input(xvis.contiguous(), vec(mask_token).reshape(-1), pos, vis_idx, msk_idx)
        ctx.p = mask_token
        ctx.meta = (vis_idx.shape[0], vis_idx.shape[1] - 1, msk_idx.shape[1])
        return out

    @staticmethod
    def backward(ctx, dout):
        B, Nvis, Nmask = ctx.meta
        dout = dout.contiguous()
        dxvis = ops.rows_window(dout, B, Nvis + Nmask, 0, Nvis)
        dmask = ops.colsum_bf16(ops.rows_window(dout, B, Nvis + Nmask, Nvis, Nmask))
        return dxvis, _ret_grad(ctx.p, dmask), None, None, None


class RowsWindowFn(torch.autograd.Function):
    """rows [start, start+count) of every clip of an fp32 stream [B*L, D] as bf16 (decoder tail `x[:, -return_token_num:]`, MP:264)"""

    @staticmethod
    def forward(ctx, x, B, L, start, count):
        ctx.meta = (B, L, start, count)
        return ops.rows_window(x.contiguous(), B, L, start, count)

    @staticmethod
    def backward(ctx, dy):
        B, L, start, count = ctx.meta
        dy2 = dy.contiguous()
        if dy2.dtype not in (BF16, F32):
            dy2 = dy2.float()
        return ops.rows_window_bwd(dy2, B, L, start, count), None, None, None, None


class MseLossFn(torch.autograd.Function):
    """nn.MSELoss()(pred, target) (ME:53,101-106): mean over every element, fp32; the gradient 2 (pred - target) / n is produced in
    the forward pass and scaled by the upstream scalar in backward."""

    @staticmethod
    def forward(ctx, pred, target):
        p2 = pred.reshape(-1, pred.shape[-1]).contiguous()
        t2 = target.reshape(-1, target.shape[-1]).contiguous().float()
        n = p2.numel()
        rows, dpred = ops.mse_rows(p2, t2, dscale=1.0 / n, want_grad=True)
        ctx.save_for_backward(dpred)
        ctx.pshape, ctx.pdtype = pred.shape, pred.dtype
        return ops.sum_rows(rows, 1.0 / n).reshape(())

    @staticmethod
    def backward(ctx, dloss):
        (dpred,) = ctx.saved_tensors
        return (dpred.float() * dloss.float()).to(ctx.pdtype).reshape(ctx.pshape), None


class CosineAlignLossFn(torch.autograd.Function):
    """(2 - 2 * (s * t).sum(-1)).mean() over every leading dim, for materialised l2-normalised features
    (multi_modality/models/criterions.py:480-482; engines/engine_for_pretraining.py:131-136).  Gradient flows to s only."""

    @staticmethod
    def forward(ctx, s, t):
        s2 = s.reshape(-1, s.shape[-1]).contiguous()
        t2 = t.reshape(-1, t.shape[-1]).contiguous()
        if s2.dtype not in (BF16, F32):
            s2 = s2.float()
        if t2.dtype not in (BF16, F32):
            t2 = t2.float()
        M = s2.shape[0]
        rows, ds = ops.cosine_rows(s2, t2, dscale=1.0 / M, want_grad=True)
        ctx.save_for_backward(ds)
        ctx.sshape, ctx.sdtype = s.shape, s.dtype
        return ops.sum_rows(rows, 1.0 / M).reshape(())

    @staticmethod
    def backward(ctx, dloss):
        (ds,) = ctx.saved_tensors
        return (ds.float() * dloss.float()).to(ctx.sdtype).reshape(ctx.sshape), None
