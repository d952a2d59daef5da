"""Tensor-level wrappers over the C ABI (one Python function per entry point of include/internvideo_hip.h).

Device memory, the current HIP stream and the caching allocator come from PyTorch-ROCm; all arithmetic is in
libinternvideo_hip.so.  Every wrapper validates shapes/dtypes on the host (the reference's `assert`s at the same
seam, e.g. models/flash_attention_class.py:35-37) and raises InternVideoHipError if the library rejects a call.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import lib as _L
from .lib import GemmDesc, InternVideoHipError, call, ptr, stream_ptr

BF16, F32 = torch.bfloat16, torch.float32
GEMM_PROFILE = None      # set to a list by bench.py to collect (kernel, a_kc, b_kc, flops, start_event, end_event) per launch
KERNEL_PROFILE = None    # set to a list by bench.py: (name, algorithmic work, "B" | "FLOP", start_event, end_event) per launch of the row / attention / optimizer kernels


def _pcall(name: str, work: float, unit: str, fname: str, *args, dyn=None) -> None:
    """call() with per-launch HIP events on the launch stream when bench.py is profiling (algorithmic bytes / FLOPs supplied by the wrapper).
    dyn = (count tensor, nominal value): the launch follows a device-side count; the profiler scales `work` by count / nominal afterwards."""
    if KERNEL_PROFILE is None:
        call(fname, *args)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    call(fname, *args)
    e1.record()
    KERNEL_PROFILE.append((name, float(work), unit, e0, e1, dyn))
# "gelu_erf_d": GELU(erf) whose `preact` output / `dact_in` input is gelu'(pre-activation) itself (include/internvideo_hip.h, act = 3)
ACT = {None: 0, "none": 0, "gelu": 1, "gelu_erf": 1, "erf": 1, "gelu_tanh": 2, "tanh": 2, "gelu_erf_d": 3}


def _chk(t: torch.Tensor, dtype, name: str, inner_contig: bool = True):
    if not t.is_cuda:
        raise InternVideoHipError(f"{name} must live in HBM (got device {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise InternVideoHipError(f"{name} must be {dtype} (got {t.dtype})")
    if inner_contig and t.numel() > 0 and t.stride(-1) != 1:
        raise InternVideoHipError(f"{name}: innermost dimension must be contiguous")
    return t


def _chk_rows(t: torch.Tensor, dtype, name: str):
    """row kernels address dense [M, D] rows (row pitch = D): a row-strided view would be read with the wrong pitch"""
    _chk(t, dtype, name)
    if not t.is_contiguous():
        raise InternVideoHipError(f"{name}: rows must be dense (contiguous [M, D]); got strides {tuple(t.stride())}")
    return t


def set_gemm_kernel(choice: int) -> None:
    """0 = per-shape heuristic (default), 1 = 128^2 4-wave kernel, 2 = 256^2 8-wave ping-pong kernel (tests / benchmarks)."""
    call("ivh_set_gemm_kernel", int(choice))


def gemm(a: torch.Tensor, b: torch.Tensor, *, a_kc: bool = True, b_kc: bool = True,
         bias: Optional[torch.Tensor] = None, act=None, want_preact: bool = False,
         dact_in: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
         out_fp32: bool = False, alpha: float = 1.0, kernel: int = 0, want_colsum: bool = False,
         m_dev: Optional[torch.Tensor] = None, k_dev: Optional[torch.Tensor] = None):
    """C[m,n] = epi(alpha * sum_k A(m,k) B(n,k)).  a: [M,K] (a_kc) or [K,M]; b: [N,K] (b_kc) or [K,N];
    optional leading batch dimension on both (b may be un-batched only if a is).
    m_dev / k_dev (int32 [1] in HBM): the real row count / contraction length, read by the kernel (<= the shape's: DropPath skipping,
    ivh_gemm_desc.m_dev / k_dev) -- rows at or past *m_dev of `out` (and of the pre-activation copy) are left untouched."""
    _L.require_gpu()
    _chk(a, BF16, "a"); _chk(b, BF16, "b")
    batched = a.dim() == 3
    if batched and b.dim() != 3:
        raise InternVideoHipError("batched gemm needs a batched b")
    a2 = a if batched else a.unsqueeze(0)
    b2 = b if batched else b.unsqueeze(0)
    nb = a2.shape[0]
    if a_kc:
        M, K = a2.shape[1], a2.shape[2]
    else:
        K, M = a2.shape[1], a2.shape[2]
    if b_kc:
        N, Kb = b2.shape[1], b2.shape[2]
    else:
        Kb, N = b2.shape[1], b2.shape[2]
    if K != Kb or b2.shape[0] != nb:
        raise InternVideoHipError(f"gemm: contraction mismatch A {tuple(a.shape)} (a_kc={a_kc}) vs B {tuple(b.shape)} (b_kc={b_kc})")
    odt = F32 if out_fp32 else BF16
    if out is None:
        out = torch.empty((nb, M, N) if batched else (M, N), dtype=odt, device=a.device)
    _chk(out, odt, "out")
    o2 = out if batched else out.unsqueeze(0)
    if tuple(o2.shape) != (nb, M, N):
        raise InternVideoHipError(f"gemm: out has shape {tuple(out.shape)}, expected {(nb, M, N)}")
    d = GemmDesc()
    d.A, d.B, d.C = a2.data_ptr(), b2.data_ptr(), o2.data_ptr()
    d.lda, d.ldb, d.ldc = a2.stride(1), b2.stride(1), o2.stride(1)
    d.M, d.N, d.K, d.a_kc, d.b_kc = M, N, K, int(a_kc), int(b_kc)
    d.c_fp32 = int(out_fp32)
    d.alpha = float(alpha)
    d.batch = nb
    d.strideA, d.strideB, d.strideC = a2.stride(0), b2.stride(0), o2.stride(0)
    d.act = ACT[act]
    pre = None
    if bias is not None:
        _chk(bias, F32, "bias")
        bb = bias if batched else bias.unsqueeze(0)
        if bb.shape[-1] != N:
            raise InternVideoHipError("gemm: bias length != N")
        d.bias, d.stride_bias = bb.data_ptr(), (bb.stride(0) if bb.shape[0] > 1 else 0)
    if want_preact:
        pre = torch.empty((nb, M, N) if batched else (M, N), dtype=BF16, device=a.device)
        p2 = pre if batched else pre.unsqueeze(0)
        d.preact, d.ldp, d.stride_preact = p2.data_ptr(), p2.stride(1), p2.stride(0)
    if dact_in is not None:
        _chk(dact_in, BF16, "dact_in")
        q2 = dact_in if batched else dact_in.unsqueeze(0)
        if tuple(q2.shape) != (nb, M, N):
            raise InternVideoHipError("gemm: dact_in shape mismatch")
        d.dact_in, d.ldd, d.stride_dact = q2.data_ptr(), q2.stride(1), q2.stride(0)
        if d.act == 0:
            raise InternVideoHipError("gemm: dact_in needs act to select the GELU flavour")
    dyn = None
    if m_dev is not None:
        d.m_dev = _cnt(m_dev, "m_dev").data_ptr()
        dyn = (m_dev, M)
    if k_dev is not None:
        d.k_dev = _cnt(k_dev, "k_dev").data_ptr()
        dyn = (k_dev, K)
    part = None
    if want_colsum:                         # bias gradient as a by-product of the 256^2 dgrad epilogue; -> (out, part | None)
        if dact_in is not None and d.act == 3 and not batched and _L.load().ivh_gemm_select(C.byref(d)) == 2:
            part = torch.empty((2 * ((M + 255) // 256), N), dtype=F32, device=a.device)
            d.colsum_part = part.data_ptr()
        res = _gemm_launch(d, a_kc, b_kc, nb, M, N, K, out, pre, want_preact, dyn)
        return (res, part)
    if kernel:                              # this launch only: 1 = 128^2, 2 = 256^2 (falls back to 1 when 2 is not built for the epilogue)
        set_gemm_kernel(kernel)
        try:
            return _gemm_launch(d, a_kc, b_kc, nb, M, N, K, out, pre, want_preact, dyn)
        finally:
            set_gemm_kernel(0)
    return _gemm_launch(d, a_kc, b_kc, nb, M, N, K, out, pre, want_preact, dyn)


def _cnt(t: torch.Tensor, name: str) -> torch.Tensor:
    """a device-side count: one int32 in HBM"""
    if not t.is_cuda or t.dtype != torch.int32 or t.numel() != 1:
        raise InternVideoHipError(f"{name} must be one int32 in HBM, got {t.dtype} {tuple(t.shape)} on {t.device}")
    return t


def _attach_split_ws(d, dev, query="ivh_gemm_split_workspace"):
    """scratch for the K split of a mostly empty last tile round (ivh_gemm_desc.split_ws); the tensor must outlive the enqueue only"""
    need = getattr(_L.load(), query)(C.byref(d))
    if need <= 0:
        return None
    ws = torch.empty((need,), dtype=torch.uint8, device=dev)
    d.split_ws, d.split_ws_bytes = ws.data_ptr(), need
    return ws


def _gemm_launch(d, a_kc, b_kc, nb, M, N, K, out, pre, want_preact, dyn=None):
    """dyn = (count tensor, nominal value) of a launch with a device-side row count: the profiler scales the nominal FLOPs by it afterwards"""
    ws = _attach_split_ws(d, out.device) if (nb == 1 and a_kc) else None    # noqa: F841  (kept alive until the launch is enqueued)
    if GEMM_PROFILE is not None:            # bench.py: per-launch HIP events on the launch stream
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("ivh_gemm_bf16", C.byref(d), stream_ptr())
        e1.record()
        kern = _L.load().ivh_gemm_select(C.byref(d))
        GEMM_PROFILE.append((kern, int(a_kc), int(b_kc), 2.0 * nb * M * N * K, e0, e1, dyn))
    else:
        call("ivh_gemm_bf16", C.byref(d), stream_ptr())
    return (out, pre) if want_preact else out


def gemm_grouped(problems, *, a_kc: bool = False, b_kc: bool = False) -> None:
    """problems: list of (a, b, out) with out[m,n] = sum_k A(m,k) B(n,k) (bf16, 2-D, same K and layouts).  One persistent launch
    over all of them when the library can group them (ivh_gemm_grouped_bf16), else one launch each."""
    _L.require_gpu()
    n = len(problems)
    arr = (GemmDesc * n)()
    for i, prob in enumerate(problems):
        a, b, out = prob[0], prob[1], prob[2]
        kd = prob[3] if len(prob) > 3 else None              # device-side contraction length of this problem (ivh_gemm_desc.k_dev)
        _chk(a, BF16, "a"); _chk(b, BF16, "b"); _chk(out, BF16, "out")
        if a.dim() != 2 or b.dim() != 2 or out.dim() != 2:
            raise InternVideoHipError("gemm_grouped: 2-D operands only")
        M, K = (a.shape if a_kc else (a.shape[1], a.shape[0]))
        N, Kb = (b.shape if b_kc else (b.shape[1], b.shape[0]))
        if K != Kb or tuple(out.shape) != (M, N):
            raise InternVideoHipError(f"gemm_grouped: problem {i}: A {tuple(a.shape)} B {tuple(b.shape)} out {tuple(out.shape)} do not match")
        d = arr[i]
        d.A, d.B, d.C = a.data_ptr(), b.data_ptr(), out.data_ptr()
        d.lda, d.ldb, d.ldc = a.stride(0), b.stride(0), out.stride(0)
        d.M, d.N, d.K, d.a_kc, d.b_kc = M, N, K, int(a_kc), int(b_kc)
        d.alpha, d.batch = 1.0, 1
        if kd is not None:
            d.k_dev = _cnt(kd, "k_dev").data_ptr()
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("ivh_gemm_grouped_bf16", arr, n, stream_ptr())
        e1.record()
        fl = sum(2.0 * d.M * d.N * d.K for d in arr)
        # per problem: (nominal FLOPs, (count tensor, nominal K) | None)
        dyn = [(2.0 * d.M * d.N * d.K, ((q[3], d.K) if (len(q) > 3 and q[3] is not None) else None)) for d, q in zip(arr, problems)]
        GEMM_PROFILE.append((2 if _L.load().ivh_gemm_select(C.byref(arr[0])) == 2 else 1, int(a_kc), int(b_kc), fl, e0, e1,
                             dyn if any(r is not None for _, r in dyn) else None))
    else:
        call("ivh_gemm_grouped_bf16", arr, n, stream_ptr())


# ---- fp8 (e4m3) path: BASELINE configs[4] ------------------------------------------------------------------------------------
FP8 = torch.float8_e4m3fn


def fp8_quantize(x: torch.Tensor, want_transposed: bool = False, amax_prev: Optional[torch.Tensor] = None,
                 amax_next: Optional[torch.Tensor] = None):
    """x bf16 [M, K] -> (q e4m3 [M, K], qt e4m3 [K, M16] | None, scale fp32 [1]) with one scale per tensor: x ~ q * scale.
    qt is the transposed copy (M16 = M rounded up to 16, pad columns zero) that dgrad / wgrad contract over.
    Current scaling (default): scale = max|x| / 448, two passes over x.  Delayed scaling: `amax_prev` (fp32 [1] in HBM: the amax this call
    site saw on the previous step) sets the scale, this call's max|x| is max-ed into `amax_next` (int32 [1], the bit pattern; the caller
    zeroes it once per step): one pass over x.  With only `amax_next` given the call scales by its own amax and records it there."""
    _L.require_gpu()
    _chk(x, BF16, "x")
    if x.dim() != 2:
        raise InternVideoHipError("fp8_quantize: 2-D input")
    M, K = x.shape
    q = torch.empty((M, K), dtype=FP8, device=x.device)
    qt = torch.empty((K, (M + 15) // 16 * 16), dtype=FP8, device=x.device) if want_transposed else None
    scale = torch.empty((1,), dtype=F32, device=x.device)
    if amax_prev is not None:
        if amax_next is None or amax_prev.dtype != F32 or amax_next.dtype != torch.int32 or amax_prev.numel() != 1 or amax_next.numel() != 1:
            raise InternVideoHipError("fp8_quantize: delayed scaling needs amax_prev fp32 [1] and amax_next int32 [1]")
        call("ivh_fp8_quantize_delayed", ptr(x), x.stride(0), M, K, ptr(q), q.stride(0), ptr(qt), (qt.stride(0) if qt is not None else 0),
             ptr(amax_prev), ptr(scale), ptr(amax_next), stream_ptr())
        return q, qt, scale
    scratch = torch.empty((1,), dtype=torch.int32, device=x.device)
    call("ivh_fp8_quantize", ptr(x), x.stride(0), M, K, ptr(q), q.stride(0), ptr(qt), (qt.stride(0) if qt is not None else 0), ptr(scale), ptr(scratch), stream_ptr())
    if amax_next is not None:                               # first sighting of a call site: remember its amax for the next step
        amax_next.copy_(scratch)
    return q, qt, scale


def fp8_quantize_weight(w: torch.Tensor):
    """w bf16 [N, K] (a Linear weight) -> (q e4m3 [N, K], qt e4m3 [K, N16], scale_rows fp32 [N], scale_cols fp32 [K]): per-channel scales.
    w ~ q * scale_rows[:, None] (the forward GEMM's B operand) and w.T ~ qt[:, :N] * scale_cols[:, None] (the dgrad GEMM's B operand): each
    image is scaled along the OUTPUT dimension of the GEMM that reads it, the only dimension along which a scale leaves the contraction."""
    _L.require_gpu()
    _chk(w, BF16, "w")
    if w.dim() != 2:
        raise InternVideoHipError("fp8_quantize_weight: 2-D input")
    N, K = w.shape
    q = torch.empty((N, K), dtype=FP8, device=w.device)
    qt = torch.empty((K, (N + 15) // 16 * 16), dtype=FP8, device=w.device)
    sr = torch.empty((N,), dtype=F32, device=w.device)
    sc = torch.empty((K,), dtype=F32, device=w.device)
    scratch = torch.empty((N + K,), dtype=torch.int32, device=w.device)
    call("ivh_fp8_quantize_weight", ptr(w), w.stride(0), N, K, ptr(q), q.stride(0), ptr(qt), qt.stride(0), ptr(sr), ptr(sc), ptr(scratch), stream_ptr())
    return q, qt, sr, sc


def set_gemm_fp8_kernel(choice: int) -> None:
    """0 = per problem (the persistent 256^2 e4m3 kernel for large problems), 1 = always the 128^2 e4m3 kernel (tests / benchmarks)"""
    call("ivh_set_gemm_fp8_kernel", int(choice))


def gemm_fp8(a: torch.Tensor, b: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor, *, bias: Optional[torch.Tensor] = None, act=None,
             want_preact: bool = False, dact_in: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, out_fp32: bool = False,
             alpha: float = 1.0, k: Optional[int] = None):
    """C[m,n] = epi(alpha * scale_a * scale_b * sum_k a[m,k] b[n,k]); a [M, K*], b [N, K*] e4m3 (K-contiguous), contraction over the first
    `k` columns (default: all; the transposed copies of fp8_quantize carry zero pad columns, so contracting over all of them is exact).
    scale_b with more than one element = one scale per row of b (per output column n: fp8_quantize_weight), length N."""
    _L.require_gpu()
    if a.dtype != FP8 or b.dtype != FP8 or a.dim() != 2 or b.dim() != 2:
        raise InternVideoHipError("gemm_fp8: a, b must be 2-D float8_e4m3fn")
    M, K = a.shape
    N = b.shape[0]
    K = K if k is None else int(k)
    if b.shape[1] < K:
        raise InternVideoHipError("gemm_fp8: contraction mismatch")
    entry = "ivh_gemm_fp8"
    if scale_b.numel() != 1:
        if scale_b.numel() != N or scale_b.dtype != F32 or not scale_b.is_contiguous():
            raise InternVideoHipError(f"gemm_fp8: per-channel scale_b must be {N} contiguous fp32 values, got {tuple(scale_b.shape)} {scale_b.dtype}")
        entry = "ivh_gemm_fp8_cs"
    odt = F32 if out_fp32 else BF16
    if out is None:
        out = torch.empty((M, N), dtype=odt, device=a.device)
    d = GemmDesc()
    d.A, d.B, d.C = a.data_ptr(), b.data_ptr(), out.data_ptr()
    d.lda, d.ldb, d.ldc = a.stride(0), b.stride(0), out.stride(0)
    d.M, d.N, d.K, d.a_kc, d.b_kc = M, N, K, 1, 1
    d.c_fp32, d.alpha, d.batch, d.act = int(out_fp32), float(alpha), 1, ACT[act]
    pre = None
    if bias is not None:
        _chk(bias, F32, "bias")
        d.bias = bias.data_ptr()
    if want_preact:
        pre = torch.empty((M, N), dtype=BF16, device=a.device)
        d.preact, d.ldp = pre.data_ptr(), pre.stride(0)
    if dact_in is not None:
        _chk(dact_in, BF16, "dact_in")
        d.dact_in, d.ldd = dact_in.data_ptr(), dact_in.stride(0)
    ws = _attach_split_ws(d, out.device, "ivh_gemm_fp8_split_workspace")    # noqa: F841
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call(entry, C.byref(d), ptr(scale_a), ptr(scale_b), stream_ptr())
        e1.record()
        GEMM_PROFILE.append((8, 1, 1, 2.0 * M * N * K, e0, e1))
    else:
        call(entry, C.byref(d), ptr(scale_a), ptr(scale_b), stream_ptr())
    return (out, pre) if want_preact else out


def rmsnorm_add_fwd(res_in: Optional[torch.Tensor], branch: Optional[torch.Tensor], gamma: Optional[torch.Tensor],
                    rowscale: Optional[torch.Tensor], rows_per_sample: int, w: Optional[torch.Tensor], eps: float,
                    want_res_out: bool = True, res_dtype: Optional[torch.dtype] = None,
                    branch_slot: Optional[torch.Tensor] = None, y_slot: Optional[torch.Tensor] = None):
    """-> (res_out [M,D] or None, y bf16 [M,D] or None, rstd fp32 [M] or None).  The residual stream is fp32 or bf16: the type of
    `res_in` (or `res_dtype` when there is none); res_out has the same type.
    branch_slot / y_slot (int32 [M / rows_per_sample], droppath_plan): `branch` holds only the kept samples of its DropPath draw, compacted;
    y is written compacted over the kept samples of the branch that consumes it (rows behind them are left untouched)."""
    _L.require_gpu()
    ref = res_in if res_in is not None else branch
    M, D = ref.shape
    rt = res_in.dtype if res_in is not None else (res_dtype or F32)
    if rt not in (F32, BF16):
        raise InternVideoHipError(f"residual stream must be fp32 or bf16, got {rt}")
    if res_in is not None: _chk_rows(res_in, rt, "res_in")
    if branch is not None: _chk_rows(branch, BF16, "branch")
    for t, n in ((gamma, "gamma"), (rowscale, "rowscale"), (w, "w")):
        if t is not None: _chk(t, F32, n)
    dev = ref.device
    res_out = torch.empty((M, D), dtype=rt, device=dev) if want_res_out else None
    y = torch.empty((M, D), dtype=BF16, device=dev) if w is not None else None
    rstd = torch.empty((M,), dtype=F32, device=dev) if w is not None else None
    rb = 4 if rt == F32 else 2
    nbytes = M * D * ((rb if res_in is not None else 0) + (2 if branch is not None else 0) + (rb if want_res_out else 0) + (2 if w is not None else 0))
    if branch_slot is not None or y_slot is not None:
        ns = M // int(rows_per_sample)
        for t, n in ((branch_slot, "branch_slot"), (y_slot, "y_slot")):
            if t is not None and (not t.is_cuda or t.dtype != torch.int32 or t.numel() != ns or not t.is_contiguous()):
                raise InternVideoHipError(f"rmsnorm_add_fwd: {n} must be a contiguous int32 [{ns}] tensor in HBM")
        _pcall("rmsnorm_add_fwd", nbytes, "B", "ivh_rmsnorm_add_fwd_skip", ptr(res_in), int(rt == BF16), ptr(branch), ptr(gamma), ptr(rowscale),
               int(rows_per_sample), ptr(w), float(eps), M, D, ptr(res_out), ptr(y), ptr(rstd), ptr(branch_slot), ptr(y_slot), stream_ptr())
        return res_out, y, rstd
    _pcall("rmsnorm_add_fwd", nbytes, "B", "ivh_rmsnorm_add_fwd" if rt == F32 else "ivh_rmsnorm_add_fwd_bf16res", ptr(res_in), ptr(branch), ptr(gamma), ptr(rowscale), int(rows_per_sample), ptr(w),
           float(eps), M, D, ptr(res_out), ptr(y), ptr(rstd), stream_ptr())
    return res_out, y, rstd


def droppath_plan(rowscale: torch.Tensor, rows_per_sample: int):
    """rowscale fp32 [..., B] (0 = the sample's branch is dropped, P:264,274) -> (slot int32 [..., B], count int32 [..., 2]):
    slot = position of the sample among the kept ones of its set (-1: dropped); count = (kept samples, kept samples * rows_per_sample)."""
    _L.require_gpu()
    _chk(rowscale, F32, "rowscale")
    if not rowscale.is_contiguous():
        raise InternVideoHipError("droppath_plan: rowscale must be contiguous")
    B = rowscale.shape[-1]
    n_sets = rowscale.numel() // B
    slot = torch.empty(rowscale.shape, dtype=torch.int32, device=rowscale.device)
    count = torch.empty(tuple(rowscale.shape[:-1]) + (2,), dtype=torch.int32, device=rowscale.device)
    call("ivh_droppath_plan", ptr(rowscale), n_sets, B, int(rows_per_sample), ptr(slot), ptr(count), stream_ptr())
    return slot, count


def _f32_vec(t: Optional[torch.Tensor], n: int) -> Optional[torch.Tensor]:
    """`t` if it can receive an fp32 [n] result in place (contiguous fp32, n elements), else None"""
    if t is None or t.dtype != F32 or t.numel() != n or not t.is_contiguous():
        return None
    return t.view(n)


def norm_bwd_parts(M: int) -> int:
    return _L.load().ivh_norm_bwd_parts(int(M))


def colsum_finish(part: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False, m_dev: Optional[torch.Tensor] = None,
                  rows_per_unit: int = 256, parts_per_unit: int = 2) -> torch.Tensor:
    """m_dev: the partials come from a launch with a device-side row count (gemm(..., m_dev=, want_colsum=True)): only the rows of the
    tiles that ran are summed"""
    n_part, D = part.shape
    if out is None:
        out = torch.empty((D,), dtype=F32, device=part.device)
        accumulate = False
    if m_dev is not None:
        call("ivh_colsum_finish_dyn", ptr(part), n_part, D, ptr(out), int(accumulate), ptr(_cnt(m_dev, "m_dev")), int(rows_per_unit), int(parts_per_unit),
             stream_ptr())
        return out
    call("ivh_colsum_finish", ptr(part), n_part, D, ptr(out), int(accumulate), stream_ptr())
    return out


def colsum_finish_multi(parts, outs=None):
    """parts: list of fp32 [n_part, D] (same shape; None entries are skipped); outs: matching list of fp32 [D] buffers or None.
    One launch for all of them.  -> list of results aligned with `parts` (None where the part was None)."""
    idx = [i for i, q in enumerate(parts) if q is not None]
    res = [None] * len(parts)
    if not idx:
        return res
    n_part, D = parts[idx[0]].shape
    for i in idx:
        o = _f32_vec(outs[i], D) if outs is not None else None
        res[i] = o if o is not None else torch.empty((D,), dtype=F32, device=parts[i].device)
    if len(idx) == 1:
        i = idx[0]
        call("ivh_colsum_finish", ptr(parts[i]), n_part, D, ptr(res[i]), 0, stream_ptr())
        return res
    pa = (C.c_void_p * len(idx))(*[parts[i].data_ptr() for i in idx])
    oa = (C.c_void_p * len(idx))(*[res[i].data_ptr() for i in idx])
    call("ivh_colsum_finish_multi", pa, oa, len(idx), n_part, D, 0, stream_ptr())
    return res


def rmsnorm_add_bwd(dy: Optional[torch.Tensor], dres_out: Optional[torch.Tensor], res_out: Optional[torch.Tensor],
                    rstd: Optional[torch.Tensor], w: Optional[torch.Tensor], branch: Optional[torch.Tensor],
                    gamma: Optional[torch.Tensor], rowscale: Optional[torch.Tensor], rows_per_sample: int,
                    want_dbranch: bool = True, inplace_dres: bool = True,
                    dw_out: Optional[torch.Tensor] = None, dg_out: Optional[torch.Tensor] = None,
                    want_dbias: bool = False, db_out: Optional[torch.Tensor] = None, dres_extra: Optional[torch.Tensor] = None,
                    y_slot: Optional[torch.Tensor] = None, branch_slot: Optional[torch.Tensor] = None):
    """-> (dres_in [M,D], dbranch bf16 [M,D] | None, dw fp32 [D] | None, dgamma fp32 [D] | None) and, with want_dbias, a fifth
    element dbias fp32 [D] = column sum of dbranch (the bias gradient of the Linear that produced `branch`).
    dw_out / dg_out / db_out: fp32 [D] buffers the column sums are written to directly (e.g. a parameter's main_grad).
    The residual stream (dres_out, res_out, dres_in) is fp32 or bf16: the type of dres_out / res_out.
    dres_extra (bf16 stream only): a second gradient of the same rows (a feature tap's), added to dres_out in fp32 as the kernel loads it."""
    _L.require_gpu()
    ref = dy if dy is not None else dres_out
    M, D = ref.shape
    dev = ref.device
    n_part = norm_bwd_parts(M)
    rt = dres_out.dtype if dres_out is not None else res_out.dtype
    if rt not in (F32, BF16) or (res_out is not None and dy is not None and res_out.dtype != rt):
        raise InternVideoHipError(f"rmsnorm_add_bwd: residual-stream tensors must share one type (fp32 or bf16), got {rt} / "
                                  f"{None if res_out is None else res_out.dtype}")
    for t, dt_, n in ((dy, BF16, "dy"), (dres_out, rt, "dres_out"), (res_out, rt, "res_out"), (branch, BF16, "branch")):
        if t is not None: _chk_rows(t, dt_, n)
    if dres_extra is not None:
        if rt != BF16 or dres_out is None:
            raise InternVideoHipError("rmsnorm_add_bwd: dres_extra joins a bf16 dres_out (fp32 streams add their taps with accum_rows)")
        _chk_rows(dres_extra, BF16, "dres_extra")
        if dres_extra.shape != dres_out.shape:
            raise InternVideoHipError(f"rmsnorm_add_bwd: dres_extra {tuple(dres_extra.shape)} vs dres_out {tuple(dres_out.shape)}")
    rb = 4 if rt == F32 else 2
    dres_in = dres_out if (inplace_dres and dres_out is not None) else torch.empty((M, D), dtype=rt, device=dev)
    dbranch = torch.empty((M, D), dtype=BF16, device=dev) if want_dbranch else None
    dw_part = torch.empty((n_part, D), dtype=F32, device=dev) if dy is not None else None
    dg_part = torch.empty((n_part, D), dtype=F32, device=dev) if (want_dbranch and gamma is not None and branch is not None) else None
    db_part = torch.empty((n_part, D), dtype=F32, device=dev) if (want_dbias and want_dbranch) else None
    nbytes = M * D * ((2 if dy is not None else 0) + (rb if dres_out is not None else 0) + (rb if (res_out is not None and dy is not None) else 0) +
                      (2 if (branch is not None and dg_part is not None) else 0) + rb + (2 if want_dbranch else 0) + (2 if dres_extra is not None else 0))
    extra = () if rt == F32 else (ptr(dres_extra),)
    if y_slot is not None or branch_slot is not None:       # dy / branch / dbranch compacted over the kept samples of their DropPath draws
        ns = M // int(rows_per_sample)
        for t, n in ((branch_slot, "branch_slot"), (y_slot, "y_slot")):
            if t is not None and (not t.is_cuda or t.dtype != torch.int32 or t.numel() != ns or not t.is_contiguous()):
                raise InternVideoHipError(f"rmsnorm_add_bwd: {n} must be a contiguous int32 [{ns}] tensor in HBM")
        _pcall("rmsnorm_add_bwd", nbytes, "B", "ivh_rmsnorm_add_bwd_skip", ptr(dy), ptr(dres_out), int(rt == BF16), ptr(res_out), ptr(rstd), ptr(w), ptr(branch),
               ptr(gamma), ptr(rowscale), int(rows_per_sample), M, D, ptr(dres_in), ptr(dbranch), ptr(dw_part), ptr(dg_part), ptr(db_part), ptr(dres_extra),
               ptr(y_slot), ptr(branch_slot), stream_ptr())
    else:
        _pcall("rmsnorm_add_bwd", nbytes, "B", "ivh_rmsnorm_add_bwd" if rt == F32 else "ivh_rmsnorm_add_bwd_bf16res", ptr(dy), ptr(dres_out), ptr(res_out), ptr(rstd), ptr(w), ptr(branch), ptr(gamma),
               ptr(rowscale), int(rows_per_sample), M, D, ptr(dres_in), ptr(dbranch), ptr(dw_part), ptr(dg_part), ptr(db_part), *extra, stream_ptr())
    dw, dg, db = colsum_finish_multi([dw_part, dg_part, db_part], [dw_out, dg_out, db_out])
    if want_dbias:
        return dres_in, dbranch, dw, dg, db
    return dres_in, dbranch, dw, dg


def colsum_bf16(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bias gradient: out[n] = sum_m x[m, n]"""
    _L.require_gpu()
    _chk(x, BF16, "x")
    M, N = x.shape
    n = _L.load().ivh_colsum_scratch_floats(M, N)
    scratch = torch.empty((n,), dtype=F32, device=x.device)
    out = _f32_vec(out, N)
    if out is None:
        out = torch.empty((N,), dtype=F32, device=x.device)
    call("ivh_colsum_bf16", ptr(x), x.stride(0), M, N, ptr(out), ptr(scratch), stream_ptr())
    return out


def qk_rmsnorm_fwd(qkv: torch.Tensor, wq: torch.Tensor, wk: torch.Tensor, eps: float, m_dev: Optional[torch.Tensor] = None):
    """in place on packed qkv [M, 3*D]; -> (rstd_q, rstd_k).  m_dev (int32 [1] in HBM): only the first *m_dev rows exist"""
    _L.require_gpu()
    _chk(qkv, BF16, "qkv"); _chk(wq, F32, "wq"); _chk(wk, F32, "wk")
    if not qkv.is_contiguous():
        raise InternVideoHipError("qkv must be contiguous")
    M, D3 = qkv.shape
    D = D3 // 3
    rq = torch.empty((M,), dtype=F32, device=qkv.device)
    rk = torch.empty((M,), dtype=F32, device=qkv.device)
    if m_dev is not None:
        _pcall("qk_rmsnorm_fwd", M * D * 8, "B", "ivh_qk_rmsnorm_fwd_dyn", ptr(qkv), ptr(wq), ptr(wk), float(eps), M, D, ptr(rq), ptr(rk), ptr(_cnt(m_dev, "m_dev")),
               stream_ptr(), dyn=(m_dev, M))
        return rq, rk
    _pcall("qk_rmsnorm_fwd", M * D * 8, "B", "ivh_qk_rmsnorm_fwd", ptr(qkv), ptr(wq), ptr(wk), float(eps), M, D, ptr(rq), ptr(rk), stream_ptr())
    return rq, rk


def qk_rmsnorm_bwd(qkv: torch.Tensor, dqkv: torch.Tensor, wq, wk, rstd_q, rstd_k, dwq_out=None, dwk_out=None, m_dev: Optional[torch.Tensor] = None):
    """dqkv rewritten in place; -> (dwq, dwk)"""
    _L.require_gpu()
    _chk(qkv, BF16, "qkv"); _chk(dqkv, BF16, "dqkv")
    M, D3 = qkv.shape
    D = D3 // 3
    n_part = _L.load().ivh_qk_norm_bwd_parts(int(M), int(D))
    pq = torch.empty((n_part, D), dtype=F32, device=qkv.device)
    pk = torch.empty((n_part, D), dtype=F32, device=qkv.device)
    if m_dev is not None:
        _pcall("qk_rmsnorm_bwd", M * D * 12, "B", "ivh_qk_rmsnorm_bwd_dyn", ptr(qkv), ptr(dqkv), ptr(wq), ptr(wk), ptr(rstd_q), ptr(rstd_k), M, D, ptr(pq), ptr(pk),
               ptr(_cnt(m_dev, "m_dev")), stream_ptr(), dyn=(m_dev, M))
    else:
        _pcall("qk_rmsnorm_bwd", M * D * 12, "B", "ivh_qk_rmsnorm_bwd", ptr(qkv), ptr(dqkv), ptr(wq), ptr(wk), ptr(rstd_q), ptr(rstd_k), M, D, ptr(pq), ptr(pk), stream_ptr())
    return tuple(colsum_finish_multi([pq, pk], [dwq_out, dwk_out]))


def set_attn_kernel(choice: int) -> None:
    """0 = automatic (32x32x16-MFMA attention kernels when the layout allows), 1 = 16x16x32 kernels, 2 = 32x32x16 or error (tests / benchmarks)."""
    call("ivh_set_attn_kernel", int(choice))


def _kv_len(kv_len: Optional[torch.Tensor], B: int) -> Optional[torch.Tensor]:
    if kv_len is None:
        return None
    if not kv_len.is_cuda or kv_len.dtype != torch.int32 or kv_len.numel() != B or not kv_len.is_contiguous():
        raise InternVideoHipError("kv_len must be a contiguous int32 [B] tensor in HBM")
    return kv_len


def flash_attn_fwd_packed(qkv: torch.Tensor, B: int, L: int, H: int, scale: Optional[float] = None,
                          kv_len: Optional[torch.Tensor] = None, drop_p: float = 0.0, seed: int = 0, nb_dev: Optional[torch.Tensor] = None):
    """qkv: [B*L, 3*H*hd] bf16 packed (three, head, d) -> (out [B*L, H*hd] bf16, lse [B,H,L] fp32).
    kv_len int32 [B]: clip b attends to its first kv_len[b] keys (right-padded batches)."""
    _L.require_gpu()
    _chk(qkv, BF16, "qkv")
    M, D3 = qkv.shape
    D = D3 // 3
    hd = D // H
    if M != B * L or not qkv.is_contiguous():
        raise InternVideoHipError("flash_attn: qkv must be contiguous [B*L, 3*D]")
    scale = float(hd ** -0.5 if scale is None else scale)
    out = torch.empty((M, D), dtype=BF16, device=qkv.device)
    lse = torch.empty((B, H, L), dtype=F32, device=qkv.device)
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + D * 2, qkv.data_ptr() + 2 * D * 2
    if drop_p > 0:
        call("ivh_flash_attn_fwd_dropout", q, L * D3, D3, hd, k, v, L * D3, D3, hd, ptr(out), L * D, D, hd, ptr(lse),
             B, H, L, L, hd, scale, ptr(_kv_len(kv_len, B)), float(drop_p), int(seed) & 0xFFFFFFFF, stream_ptr())
        return out, lse
    if nb_dev is not None:                                  # only the first *nb_dev clips exist (DropPath skipping); the rest of out / lse is untouched
        _pcall("flash_attn_fwd", 4.0 * B * H * L * L * hd, "FLOP", "ivh_flash_attn_fwd_dyn", q, L * D3, D3, hd, k, v, L * D3, D3, hd, ptr(out), L * D, D, hd, ptr(lse),
               B, H, L, L, hd, scale, ptr(_kv_len(kv_len, B)), ptr(_cnt(nb_dev, "nb_dev")), stream_ptr(), dyn=(nb_dev, B))
        return out, lse
    _pcall("flash_attn_fwd", 4.0 * B * H * L * L * hd, "FLOP", "ivh_flash_attn_fwd", q, L * D3, D3, hd, k, v, L * D3, D3, hd, ptr(out), L * D, D, hd, ptr(lse),
           B, H, L, L, hd, scale, ptr(_kv_len(kv_len, B)), stream_ptr())
    return out, lse


def flash_attn_bwd_packed(qkv: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, lse: torch.Tensor,
                          B: int, L: int, H: int, scale: Optional[float] = None, kv_len: Optional[torch.Tensor] = None,
                          drop_p: float = 0.0, seed: int = 0, nb_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> dqkv [B*L, 3*D] bf16 (d q_hat, d k_hat, dv)"""
    _L.require_gpu()
    _chk(qkv, BF16, "qkv"); _chk(out, BF16, "out"); _chk(dout, BF16, "dout"); _chk(lse, F32, "lse")
    M, D3 = qkv.shape
    D = D3 // 3
    hd = D // H
    scale = float(hd ** -0.5 if scale is None else scale)
    if not (dout.is_contiguous() and out.is_contiguous()):
        raise InternVideoHipError("flash_attn_bwd: out / dout must be contiguous")
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, L), dtype=F32, device=qkv.device)
    q, k, v = qkv.data_ptr(), qkv.data_ptr() + D * 2, qkv.data_ptr() + 2 * D * 2
    dq, dk, dv = dqkv.data_ptr(), dqkv.data_ptr() + D * 2, dqkv.data_ptr() + 2 * D * 2
    if drop_p > 0:
        call("ivh_flash_attn_bwd_dropout", q, L * D3, D3, hd, k, v, L * D3, D3, hd, ptr(out), ptr(dout), L * D, D, hd,
             ptr(lse), ptr(delta), dq, L * D3, D3, hd, dk, dv, L * D3, D3, hd, B, H, L, L, hd, scale, ptr(_kv_len(kv_len, B)),
             float(drop_p), int(seed) & 0xFFFFFFFF, stream_ptr())
        return dqkv
    if nb_dev is not None:
        _pcall("flash_attn_bwd", 10.0 * B * H * L * L * hd, "FLOP", "ivh_flash_attn_bwd_dyn", q, L * D3, D3, hd, k, v, L * D3, D3, hd, ptr(out), ptr(dout), L * D, D, hd,
               ptr(lse), ptr(delta), dq, L * D3, D3, hd, dk, dv, L * D3, D3, hd, B, H, L, L, hd, scale, ptr(_kv_len(kv_len, B)), ptr(_cnt(nb_dev, "nb_dev")),
               stream_ptr(), dyn=(nb_dev, B))
        return dqkv
    _pcall("flash_attn_bwd", 10.0 * B * H * L * L * hd, "FLOP", "ivh_flash_attn_bwd", q, L * D3, D3, hd, k, v, L * D3, D3, hd, ptr(out), ptr(dout), L * D, D, hd,
           ptr(lse), ptr(delta), dq, L * D3, D3, hd, dk, dv, L * D3, D3, hd, B, H, L, L, hd, scale, ptr(_kv_len(kv_len, B)), stream_ptr())
    return dqkv


def _bshd(t: torch.Tensor, name: str):
    _chk(t, BF16, name)
    if t.dim() != 4:
        raise InternVideoHipError(f"{name} must be [B, L, H, hd]")
    return t


def flash_attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: Optional[float] = None, kv_len: Optional[torch.Tensor] = None,
                   drop_p: float = 0.0, seed: int = 0):
    """q [B,Lq,H,hd]; k, v [B,Lk,H,hd] bf16 (k and v sharing strides; any strides with hd contiguous) -> out [B,Lq,H,hd], lse [B,H,Lq]"""
    _L.require_gpu()
    _bshd(q, "q"); _bshd(k, "k"); _bshd(v, "v")
    B, Lq, H, hd = q.shape
    Lk = k.shape[1]
    if k.stride() != v.stride() or k.shape != v.shape:
        raise InternVideoHipError("flash_attn_fwd: k and v must share shape and strides")
    scale = float(hd ** -0.5 if scale is None else scale)
    out = torch.empty((B, Lq, H, hd), dtype=BF16, device=q.device)
    lse = torch.empty((B, H, Lq), dtype=F32, device=q.device)
    if drop_p > 0:
        call("ivh_flash_attn_fwd_dropout", ptr(q), q.stride(0), q.stride(1), q.stride(2), ptr(k), ptr(v), k.stride(0), k.stride(1), k.stride(2),
             ptr(out), out.stride(0), out.stride(1), out.stride(2), ptr(lse), B, H, Lq, Lk, hd, scale, ptr(_kv_len(kv_len, B)),
             float(drop_p), int(seed) & 0xFFFFFFFF, stream_ptr())
        return out, lse
    call("ivh_flash_attn_fwd", ptr(q), q.stride(0), q.stride(1), q.stride(2), ptr(k), ptr(v), k.stride(0), k.stride(1), k.stride(2),
         ptr(out), out.stride(0), out.stride(1), out.stride(2), ptr(lse), B, H, Lq, Lk, hd, scale, ptr(_kv_len(kv_len, B)), stream_ptr())
    return out, lse


def flash_attn_bwd(q, k, v, out, dout, lse, scale: Optional[float] = None, kv_len: Optional[torch.Tensor] = None, drop_p: float = 0.0, seed: int = 0):
    """-> (dq like q (contiguous), dkv [2,B,Lk,H,hd] contiguous: dk = dkv[0], dv = dkv[1])"""
    _L.require_gpu()
    B, Lq, H, hd = q.shape
    Lk = k.shape[1]
    scale = float(hd ** -0.5 if scale is None else scale)
    dout = dout.contiguous()
    if not out.is_contiguous():
        raise InternVideoHipError("flash_attn_bwd: out must be contiguous")
    dq = torch.empty((B, Lq, H, hd), dtype=BF16, device=q.device)
    dkv = torch.empty((2, B, Lk, H, hd), dtype=BF16, device=q.device)
    delta = torch.empty((B, H, Lq), dtype=F32, device=q.device)
    args = (ptr(q), q.stride(0), q.stride(1), q.stride(2), ptr(k), ptr(v), k.stride(0), k.stride(1), k.stride(2),
            ptr(out), ptr(dout), out.stride(0), out.stride(1), out.stride(2), ptr(lse), ptr(delta),
            ptr(dq), dq.stride(0), dq.stride(1), dq.stride(2), ptr(dkv[0]), ptr(dkv[1]), dkv.stride(1), dkv.stride(2), dkv.stride(3),
            B, H, Lq, Lk, hd, scale, ptr(_kv_len(kv_len, B)))
    if drop_p > 0:
        call("ivh_flash_attn_bwd_dropout", *args, float(drop_p), int(seed) & 0xFFFFFFFF, stream_ptr())
    else:
        call("ivh_flash_attn_bwd", *args, stream_ptr())
    return dq, dkv


# ---- attention-pool pieces ------------------------------------------------------------------------------------------
def token_mean_fwd(x: torch.Tensor, B: int, L: int) -> torch.Tensor:
    _L.require_gpu()
    _chk(x, F32, "x")
    D = x.shape[-1]
    out = torch.empty((B, D), dtype=F32, device=x.device)
    call("ivh_token_mean_fwd", ptr(x), B, L, D, ptr(out), stream_ptr())
    return out


def token_mean_bwd(dmean: torch.Tensor, dx: torch.Tensor, B: int, L: int) -> None:
    _L.require_gpu()
    _chk(dmean, F32, "dmean"); _chk(dx, F32, "dx")
    call("ivh_token_mean_bwd", ptr(dmean), B, L, dx.shape[-1], ptr(dx), stream_ptr())


def layernorm_fwd(x: torch.Tensor, w, b, eps: float, w2=None, b2=None):
    """x fp32|bf16 [M,C] -> (y bf16, y2 bf16|None, stats fp32 [M,2])"""
    _L.require_gpu()
    if x.dtype not in (F32, BF16) or not x.is_contiguous():
        raise InternVideoHipError("layernorm_fwd: x must be contiguous fp32/bf16")
    M, Cc = x.shape
    y = torch.empty((M, Cc), dtype=BF16, device=x.device)
    y2 = torch.empty((M, Cc), dtype=BF16, device=x.device) if w2 is not None else None
    stats = torch.empty((M, 2), dtype=F32, device=x.device)
    call("ivh_layernorm_fwd", ptr(x), int(x.dtype == F32), ptr(w), ptr(b), ptr(w2), ptr(b2), float(eps), M, Cc, ptr(y), ptr(y2), ptr(stats), stream_ptr())
    return y, y2, stats


def layernorm_bwd(x, w, stats, dy, w2=None, dy2=None, dx: Optional[torch.Tensor] = None, accumulate: bool = False):
    """-> (dx fp32, dw, db, dw2|None, db2|None)"""
    _L.require_gpu()
    M, Cc = x.shape
    n_part = norm_bwd_parts(M)
    if dx is None:
        dx = torch.empty((M, Cc), dtype=F32, device=x.device)
        accumulate = False
    parts = [torch.empty((n_part, Cc), dtype=F32, device=x.device) for _ in range(4 if w2 is not None else 2)]
    call("ivh_layernorm_bwd", ptr(x), int(x.dtype == F32), ptr(w), ptr(w2), ptr(stats), ptr(dy), ptr(dy2), M, Cc, ptr(dx), int(accumulate),
         ptr(parts[0]), ptr(parts[1]), ptr(parts[2]) if w2 is not None else None, ptr(parts[3]) if w2 is not None else None, stream_ptr())
    outs = colsum_finish_multi(parts)
    return (dx, outs[0], outs[1], outs[2] if w2 is not None else None, outs[3] if w2 is not None else None)


# ---- token edges --------------------------------------------------------------------------------------------------
def mask_to_indices(mask: torch.Tensor, L: int):
    """mask bool/uint8 [B, 1+N] on device, True = masked -> (vis_idx int32 [B,L], inv_idx int32 [B,1+N], count int32 [B])"""
    _L.require_gpu()
    m = mask.to(torch.uint8) if mask.dtype != torch.uint8 else mask
    m = m.contiguous()
    B, N1 = m.shape
    vis = torch.empty((B, L), dtype=torch.int32, device=m.device)
    inv = torch.empty((B, N1), dtype=torch.int32, device=m.device)
    cnt = torch.empty((B,), dtype=torch.int32, device=m.device)
    call("ivh_mask_to_indices", ptr(m), B, N1, L, ptr(vis), ptr(inv), ptr(cnt), stream_ptr())
    return vis, inv, cnt


def patch_im2col(video: torch.Tensor, vis_idx: torch.Tensor, tubelet: int, patch: int, Kp: int) -> torch.Tensor:
    _L.require_gpu()
    if video.dtype not in (F32, BF16):
        raise InternVideoHipError("video must be fp32 or bf16")
    video = video.contiguous()
    B, Cc, T, H, W = video.shape
    L = vis_idx.shape[1]
    cols = torch.empty((B * (L - 1), Kp), dtype=BF16, device=video.device)
    call("ivh_patch_im2col", ptr(video), int(video.dtype == F32), ptr(vis_idx), B, Cc, T, H, W, tubelet, patch, L, Kp, ptr(cols), stream_ptr())
    return cols


def assemble_tokens(tok: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, vis_idx: torch.Tensor) -> torch.Tensor:
    _L.require_gpu()
    _chk(tok, BF16, "tok"); _chk(cls, F32, "cls"); _chk(pos, F32, "pos")
    B, L = vis_idx.shape
    D = tok.shape[-1]
    x0 = torch.empty((B * L, D), dtype=F32, device=tok.device)
    call("ivh_assemble_tokens", ptr(tok), ptr(cls), ptr(pos), ptr(vis_idx), B, L, D, ptr(x0), stream_ptr())
    return x0


def add_pos_gather(x: torch.Tensor, pos: torch.Tensor, vis_idx: torch.Tensor, skip: int) -> torch.Tensor:
    """x: a tap of the residual stream, fp32 or bf16 [B*L, D] -> bf16 [B*(L-skip), D]"""
    _L.require_gpu()
    _chk(x, x.dtype if x.dtype in (F32, BF16) else F32, "x"); _chk(pos, F32, "pos")
    if not x.is_contiguous():
        raise InternVideoHipError("add_pos_gather: x must be contiguous")
    B, L = vis_idx.shape
    D = x.shape[-1]
    y = torch.empty((B * (L - skip), D), dtype=BF16, device=x.device)
    call("ivh_add_pos_gather" if x.dtype == F32 else "ivh_add_pos_gather_bf16", ptr(x), ptr(pos), ptr(vis_idx), B, L, D, skip, ptr(y), stream_ptr())
    return y


def rows_shift_bf16(src: torch.Tensor, B: int, L: int, skip: int) -> torch.Tensor:
    """src bf16 [B*(L-skip), D] -> bf16 [B*L, D] with dst[b, j+skip] = src[b, j] and zero rows j < skip"""
    _L.require_gpu()
    _chk(src, BF16, "src")
    if not src.is_contiguous():
        raise InternVideoHipError("rows_shift_bf16: src must be contiguous")
    D = src.shape[-1]
    dst = torch.empty((B * L, D), dtype=BF16, device=src.device)
    call("ivh_rows_shift_bf16", ptr(dst), ptr(src), B, L, D, skip, stream_ptr())
    return dst


def gather_rows(src: torch.Tensor, idx: torch.Tensor, skip: int = 0) -> torch.Tensor:
    """src [K,B,Nsrc,C] or [B,Nsrc,C] (any dtype whose rows are a multiple of 16 bytes), idx int32 [B,L] ->
    dst[..., b, j, :] = src[..., b, idx[b, j+skip]-skip, :]   (`t[~mask].reshape(K,B,-1,C)`, bit-exact copy)"""
    _L.require_gpu()
    if not src.is_cuda or not src.is_contiguous():
        raise InternVideoHipError("gather_rows: src must be a contiguous HBM tensor")
    _chk(idx, torch.int32, "idx")
    s4 = src if src.dim() == 4 else src.unsqueeze(0)
    if s4.dim() != 4 or idx.dim() != 2 or not idx.is_contiguous() or idx.shape[0] != s4.shape[1]:
        raise InternVideoHipError(f"gather_rows: src {tuple(src.shape)} / idx {tuple(idx.shape)} do not match")
    K, B, Nsrc, Cc = s4.shape
    L = idx.shape[1]
    dst = torch.empty((K, B, L - skip, Cc), dtype=src.dtype, device=src.device)
    call("ivh_gather_rows", ptr(s4), Cc * src.element_size(), K, B, Nsrc, ptr(idx), L, int(skip), ptr(dst), stream_ptr())
    return dst if src.dim() == 4 else dst[0]


def rows_to_bf16(src: torch.Tensor, B: int, L: int, skip: int) -> torch.Tensor:
    _L.require_gpu()
    _chk(src, F32, "src")
    D = src.shape[-1]
    dst = torch.empty((B * (L - skip), D), dtype=BF16, device=src.device)
    call("ivh_rows_to_bf16", ptr(src), B, L, D, skip, ptr(dst), stream_ptr())
    return dst


def accum_rows(dst: torch.Tensor, src: torch.Tensor, B: int, L: int, skip: int, accumulate: bool) -> None:
    _L.require_gpu()
    _chk(dst, F32, "dst")
    D = dst.shape[-1]
    call("ivh_accum_rows", ptr(dst), ptr(src), int(src.dtype == BF16), B, L, D, skip, int(accumulate), stream_ptr())


def pos_grad(src: torch.Tensor, K: int, B: int, Lsrc: int, inv_idx: torch.Tensor, skip: int,
             dpos: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    _L.require_gpu()
    D = src.shape[-1]
    N1 = inv_idx.shape[1]
    if dpos is None:
        dpos = torch.empty((N1 - skip, D), dtype=F32, device=src.device)
        accumulate = False
    call("ivh_pos_grad", ptr(src), int(src.dtype == BF16), K, B, Lsrc, D, ptr(inv_idx), N1, skip, ptr(dpos), int(accumulate), stream_ptr())
    return dpos


# ---- VideoMAE pixel path ------------------------------------------------------------------------------------------------
def assemble_tokens_nocls(tok: torch.Tensor, pos: torch.Tensor, vis_idx: torch.Tensor) -> torch.Tensor:
    """tok bf16 [B*(L-1), D], pos fp32 [N, D], vis_idx int32 [B, L] (leading pseudo-cls 0) -> fp32 [B*(L-1), D]"""
    _L.require_gpu()
    _chk(tok, BF16, "tok"); _chk(pos, F32, "pos"); _chk(vis_idx, torch.int32, "vis_idx")
    B, L = vis_idx.shape
    D = tok.shape[-1]
    x0 = torch.empty((B * (L - 1), D), dtype=F32, device=tok.device)
    call("ivh_assemble_tokens_nocls", ptr(tok), ptr(pos), ptr(vis_idx), B, L, D, ptr(x0), stream_ptr())
    return x0


def mae_decoder_input(xvis: torch.Tensor, mask_token: torch.Tensor, pos: torch.Tensor, vis_idx: torch.Tensor,
                      msk_idx: torch.Tensor) -> torch.Tensor:
    """xvis bf16 [B*Nvis, D]; mask_token fp32 [D]; pos fp32 [N, D]; vis_idx int32 [B, 1+Nvis]; msk_idx int32 [B, Nmask]
    -> fp32 [B*(Nvis+Nmask), D] = cat([x_vis + pos[~mask], mask_token + pos[mask]], 1)"""
    _L.require_gpu()
    _chk(xvis, BF16, "xvis"); _chk(mask_token, F32, "mask_token"); _chk(pos, F32, "pos")
    B, Nv1 = vis_idx.shape
    Nvis, Nmask = Nv1 - 1, msk_idx.shape[1]
    D = xvis.shape[-1]
    out = torch.empty((B * (Nvis + Nmask), D), dtype=F32, device=xvis.device)
    call("ivh_mae_decoder_input", ptr(xvis), ptr(mask_token), ptr(pos), ptr(vis_idx), ptr(msk_idx), B, Nvis, Nmask, D, ptr(out), stream_ptr())
    return out


def rows_window(src: torch.Tensor, B: int, L: int, start: int, count: int) -> torch.Tensor:
    """fp32 [B*L, D] -> bf16 [B*count, D]: rows [start, start+count) of every clip"""
    _L.require_gpu()
    _chk(src, F32, "src")
    D = src.shape[-1]
    dst = torch.empty((B * count, D), dtype=BF16, device=src.device)
    call("ivh_rows_window", ptr(src), B, L, D, int(start), int(count), ptr(dst), stream_ptr())
    return dst


def rows_window_bwd(src: torch.Tensor, B: int, L: int, start: int, count: int) -> torch.Tensor:
    """bf16|fp32 [B*count, D] -> fp32 [B*L, D] with the window rows filled and every other row zero"""
    _L.require_gpu()
    if src.dtype not in (BF16, F32) or not src.is_contiguous():
        raise InternVideoHipError("rows_window_bwd: src must be contiguous bf16/fp32")
    D = src.shape[-1]
    dst = torch.empty((B * L, D), dtype=F32, device=src.device)
    call("ivh_rows_window_bwd", ptr(src), int(src.dtype == BF16), B, L, D, int(start), int(count), ptr(dst), stream_ptr())
    return dst


def pixel_target(video: torch.Tensor, msk_idx: torch.Tensor, tubelet: int, patch: int, normalize: bool = True,
                 mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)) -> torch.Tensor:
    """video fp32|bf16 (B,3,T,H,W) (ImageNet-normalised), msk_idx int32 [B, Nmask] (token id + 1) -> fp32 (B, Nmask, tubelet*p*p*3)"""
    _L.require_gpu()
    if video.dtype not in (F32, BF16):
        raise InternVideoHipError("video must be fp32 or bf16")
    video = video.contiguous()
    _chk(msk_idx, torch.int32, "msk_idx")
    B, Cc, T, H, W = video.shape
    Nmask = msk_idx.shape[1]
    out = torch.empty((B, Nmask, tubelet * patch * patch * Cc), dtype=F32, device=video.device)
    m3 = (C.c_float * 3)(*[float(v) for v in mean]); s3 = (C.c_float * 3)(*[float(v) for v in std])
    call("ivh_pixel_target", ptr(video), int(video.dtype == F32), ptr(msk_idx.contiguous()), B, Cc, T, H, W, int(tubelet), int(patch), Nmask,
         int(normalize), m3, s3, ptr(out), stream_ptr())
    return out


def mse_rows(pred: torch.Tensor, target: torch.Tensor, dscale: float = 0.0, want_grad: bool = True):
    """pred bf16|fp32 [M,C], target fp32 [M,C] -> (rows fp32 [M] = sum_c (pred-target)^2, dpred bf16 [M,C] = 2*dscale*(pred-target) | None)"""
    _L.require_gpu()
    if pred.dtype not in (F32, BF16) or not pred.is_contiguous():
        raise InternVideoHipError("mse_rows: pred must be contiguous bf16/fp32")
    _chk(target, F32, "target")
    M, Cc = pred.shape
    if target.numel() != M * Cc or not target.is_contiguous():
        raise InternVideoHipError("mse_rows: target must be contiguous with pred's shape")
    rows = torch.empty((M,), dtype=F32, device=pred.device)
    dpred = torch.empty((M, Cc), dtype=BF16, device=pred.device) if want_grad else None
    call("ivh_mse_rows", ptr(pred), int(pred.dtype == F32), ptr(target), M, Cc, float(dscale), ptr(rows), ptr(dpred), stream_ptr())
    return rows, dpred


def cosine_rows(s: torch.Tensor, t: torch.Tensor, dscale: float = 0.0, want_grad: bool = True):
    """s, t bf16|fp32 [M,C] -> (rows fp32 [M] = 2 - 2 <s, t>, ds bf16 [M,C] = -2 * dscale * t | None)"""
    _L.require_gpu()
    for x, n in ((s, "s"), (t, "t")):
        if x.dtype not in (F32, BF16) or not x.is_contiguous() or not x.is_cuda:
            raise InternVideoHipError(f"cosine_rows: {n} must be a contiguous bf16/fp32 HBM tensor")
    M, Cc = s.shape
    if t.numel() != M * Cc:
        raise InternVideoHipError("cosine_rows: s and t must have the same shape")
    rows = torch.empty((M,), dtype=F32, device=s.device)
    ds = torch.empty((M, Cc), dtype=BF16, device=s.device) if want_grad else None
    call("ivh_cosine_rows", ptr(s), int(s.dtype == F32), ptr(t), int(t.dtype == F32), M, Cc, float(dscale), ptr(rows), ptr(ds), stream_ptr())
    return rows, ds


# ---- teacher tails ------------------------------------------------------------------------------------------------
def frames_merge_l2(x: torch.Tensor, B: int, T: int, L: int, l2: bool = True, out_fp32: bool = False) -> torch.Tensor:
    """x fp32|bf16 [B*T*L, C] (per-frame sequences) -> [B, 1 + T*(L-1), C]: cls rows averaged over the frames, patch rows
    concatenated frame-major, rows l2-normalised (internvl_clip_vision.py:445-456).  L = 1: mean over frames (+ l2)."""
    _L.require_gpu()
    if x.dtype not in (F32, BF16) or not x.is_contiguous() or x.numel() != B * T * L * x.shape[-1]:
        raise InternVideoHipError("frames_merge_l2: x must be contiguous fp32/bf16 with B*T*L rows")
    Cc = x.shape[-1]
    out = torch.empty((B, 1 + T * (L - 1), Cc), dtype=F32 if out_fp32 else BF16, device=x.device)
    call("ivh_frames_merge_l2", ptr(x), int(x.dtype == F32), B, T, L, Cc, int(l2), ptr(out), int(out_fp32), stream_ptr())
    return out


def pool_attn_map(q: torch.Tensor, k: torch.Tensor, scale: Optional[float] = None, skip: int = 1) -> torch.Tensor:
    """q bf16 [S, H, hd] (contiguous), k bf16 [S, L, H, hd] (hd, H contiguous) -> fp32 [S, L - skip]: the head-averaged
    probabilities of the 1-query pooling attention over the keys l >= skip (internvl_clip_vision.py:82-83,463)."""
    _L.require_gpu()
    _chk(q, BF16, "q"); _chk(k, BF16, "k")
    S, Lk, H, hd = k.shape
    if tuple(q.shape) != (S, H, hd) or not q.is_contiguous() or k.stride(3) != 1 or k.stride(2) != hd:
        raise InternVideoHipError("pool_attn_map: q must be contiguous [S,H,hd] and k [S,L,H,hd] with (H,hd) contiguous")
    scale = float(hd ** -0.5 if scale is None else scale)
    out = torch.empty((S, Lk - skip), dtype=F32, device=q.device)
    call("ivh_pool_attn_map", ptr(q), ptr(k), k.stride(0), k.stride(1), S, Lk, H, hd, scale, int(skip), ptr(out), stream_ptr())
    return out


def pool_attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: Optional[float] = None):
    """q bf16 [S,H,hd] contiguous; k, v bf16 [S,L,H,hd] ((H,hd) contiguous, same strides) -> (o bf16 [S,H,hd], lse fp32 [S,H])"""
    _L.require_gpu()
    _chk(q, BF16, "q"); _chk(k, BF16, "k"); _chk(v, BF16, "v")
    S, Lk, H, hd = k.shape
    if tuple(q.shape) != (S, H, hd) or not q.is_contiguous() or k.stride() != v.stride() or k.stride(3) != 1 or k.stride(2) != hd:
        raise InternVideoHipError("pool_attn_fwd: q must be contiguous [S,H,hd]; k, v [S,L,H,hd] with (H,hd) contiguous and equal strides")
    scale = float(hd ** -0.5 if scale is None else scale)
    o = torch.empty((S, H, hd), dtype=BF16, device=q.device)
    lse = torch.empty((S, H), dtype=F32, device=q.device)
    call("ivh_pool_attn_fwd", ptr(q), ptr(k), ptr(v), k.stride(0), k.stride(1), S, Lk, H, hd, scale, ptr(o), ptr(lse), stream_ptr())
    return o, lse


def pool_attn_bwd(q, k, v, dout, lse, scale: Optional[float] = None):
    """-> (dq bf16 [S,H,hd], dkv bf16 [2,S,L,H,hd]: dk = dkv[0], dv = dkv[1])"""
    _L.require_gpu()
    S, Lk, H, hd = k.shape
    scale = float(hd ** -0.5 if scale is None else scale)
    dout = dout.contiguous()
    _chk(dout, BF16, "dout"); _chk(lse, F32, "lse")
    dq = torch.empty((S, H, hd), dtype=BF16, device=q.device)
    dkv = torch.empty((2, S, Lk, H, hd), dtype=BF16, device=q.device)
    call("ivh_pool_attn_bwd", ptr(q), ptr(k), ptr(v), k.stride(0), k.stride(1), ptr(dout), ptr(lse), S, Lk, H, hd, scale,
         ptr(dq), ptr(dkv[0]), ptr(dkv[1]), stream_ptr())
    return dq, dkv


# ---- decoder tail ---------------------------------------------------------------------------------------------------
def ln_l2_fwd(y: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float, want_out: bool = True,
              target: Optional[torch.Tensor] = None):
    """-> (out bf16 [M,C] | None, stats fp32 [M,3], loss_rows fp32 [M] | None)"""
    _L.require_gpu()
    _chk(y, BF16, "y"); _chk(w, F32, "w"); _chk(b, F32, "b")
    M, Cc = y.shape
    out = torch.empty((M, Cc), dtype=BF16, device=y.device) if want_out else None
    stats = torch.empty((M, 3), dtype=F32, device=y.device)
    loss_rows = torch.empty((M,), dtype=F32, device=y.device) if target is not None else None
    tb = 0
    if target is not None:
        if target.dtype not in (BF16, F32) or not target.is_contiguous() or target.numel() != M * Cc:
            raise InternVideoHipError("ln_l2_fwd: target must be a contiguous bf16/fp32 [M,C] tensor")
        tb = int(target.dtype == BF16)
    call("ivh_ln_l2_fwd", ptr(y), ptr(w), ptr(b), float(eps), M, Cc, ptr(out), ptr(stats), ptr(target), tb, ptr(loss_rows), stream_ptr())
    return out, stats, loss_rows


def ln_l2_bwd(y, w, b, stats, dout: Optional[torch.Tensor], target: Optional[torch.Tensor], dscale: float,
              dscale_dev: Optional[torch.Tensor] = None, dw_out: Optional[torch.Tensor] = None, db_out: Optional[torch.Tensor] = None):
    """-> (dy bf16 [M,C], dw fp32 [C], db fp32 [C]); dw_out / db_out: fp32 [C] buffers the column sums are written to directly (a main_grad)"""
    _L.require_gpu()
    M, Cc = y.shape
    n_part = norm_bwd_parts(M)
    dy = torch.empty((M, Cc), dtype=BF16, device=y.device)
    pw = torch.empty((n_part, Cc), dtype=F32, device=y.device)
    pb = torch.empty((n_part, Cc), dtype=F32, device=y.device)
    call("ivh_ln_l2_bwd", ptr(y), ptr(w), ptr(b), ptr(stats), ptr(dout), int(dout is not None and dout.dtype == BF16),
         ptr(target), int(target is not None and target.dtype == BF16), float(dscale), ptr(dscale_dev), M, Cc, ptr(dy), ptr(pw), ptr(pb), stream_ptr())
    dw, db = colsum_finish_multi([pw, pb], [dw_out, db_out])
    return dy, dw, db


def sum_rows(x: torch.Tensor, scale: float) -> torch.Tensor:
    _L.require_gpu()
    _chk(x, F32, "x")
    out = torch.empty((1,), dtype=F32, device=x.device)
    call("ivh_sum_rows", ptr(x), x.numel(), float(scale), ptr(out), stream_ptr())
    return out


# ---- stage-2 text / fusion tower rows (bert.hip) -------------------------------------------------------------------------
def _ids32(ids: torch.Tensor, name: str = "ids") -> torch.Tensor:
    if not ids.is_cuda:
        raise InternVideoHipError(f"{name} must live in HBM")
    return ids.reshape(-1).to(torch.int32).contiguous()


def bert_embed_fwd(ids: torch.Tensor, L: int, word: torch.Tensor, pos: torch.Tensor, type_: torch.Tensor, w, b, eps: float,
                   drop_p: float = 0.0, seed: int = 0):
    """ids int [B*L] -> (y bf16 [B*L, C] = dropout(LayerNorm((word[ids] + type[0]) + pos[l])), stats fp32 [B*L, 2])"""
    _L.require_gpu()
    _chk(word, F32, "word"); _chk(pos, F32, "pos"); _chk(type_, F32, "type")
    ids = _ids32(ids)
    M, Cc = ids.numel(), word.shape[1]
    if L > pos.shape[0]:
        raise InternVideoHipError(f"sequence length {L} exceeds the position table ({pos.shape[0]})")
    y = torch.empty((M, Cc), dtype=BF16, device=word.device)
    stats = torch.empty((M, 2), dtype=F32, device=word.device)
    call("ivh_bert_embed_fwd", ptr(ids), M, int(L), ptr(word), ptr(pos), ptr(type_), ptr(w), ptr(b), float(eps), Cc, ptr(y), ptr(stats), float(drop_p),
         int(seed) & 0xFFFFFFFF, stream_ptr())
    return y, stats


def bert_embed_bwd(ids, L: int, word, pos, type_, w, stats, dy, pad_id: int, dword, dpos, dtype_, drop_p: float = 0.0, seed: int = 0):
    """adds the row gradients into dword / dpos / dtype_ (fp32, same shapes as the tables) -> (dw, db) of the LayerNorm"""
    _L.require_gpu()
    _chk(dy, BF16, "dy"); _chk(dword, F32, "dword"); _chk(dpos, F32, "dpos"); _chk(dtype_, F32, "dtype")
    ids = _ids32(ids)
    M, Cc = ids.numel(), word.shape[1]
    n_part = norm_bwd_parts(M)
    parts = [torch.empty((n_part, Cc), dtype=F32, device=word.device) for _ in range(2)]
    call("ivh_bert_embed_bwd", ptr(ids), M, int(L), ptr(word), ptr(pos), ptr(type_), ptr(w), ptr(stats), ptr(dy), Cc, int(pad_id),
         ptr(dword), ptr(dpos), ptr(dtype_), ptr(parts[0]), ptr(parts[1]), float(drop_p), int(seed) & 0xFFFFFFFF, stream_ptr())
    dw, db = colsum_finish_multi(parts)
    return dw, db


def add_layernorm_fwd(a: torch.Tensor, r: Optional[torch.Tensor], w, b, eps: float, gelu: bool = False, drop_p: float = 0.0, seed: int = 0):
    """-> (y bf16 = LayerNorm(a + r) [gelu: LayerNorm(gelu_erf(a)), r None], stats fp32 [M, 2]); a, r bf16 [M, C] contiguous"""
    _L.require_gpu()
    _chk(a, BF16, "a")
    if r is not None:
        _chk(r, BF16, "r")
        if r.shape != a.shape or not r.is_contiguous():
            raise InternVideoHipError("add_layernorm: a and r must be contiguous and share a shape")
    if not a.is_contiguous():
        raise InternVideoHipError("add_layernorm: a must be contiguous")
    M, Cc = a.shape
    y = torch.empty_like(a)
    stats = torch.empty((M, 2), dtype=F32, device=a.device)
    nbytes = M * Cc * (2 * (3 if r is not None else 2))
    _pcall("add_layernorm_fwd", nbytes, "B", "ivh_add_layernorm_fwd", ptr(a), ptr(r), int(gelu), ptr(w), ptr(b), float(eps), M, Cc, ptr(y), ptr(stats),
           float(drop_p), int(seed) & 0xFFFFFFFF, stream_ptr())
    return y, stats


def add_layernorm_bwd(a, r, w, stats, dy, dy2=None, gelu: bool = False, drop_p: float = 0.0, seed: int = 0):
    """-> (dx bf16 [M, C], dw fp32 [C], db fp32 [C]) or, with drop_p > 0, ((dx of r, dx of a), dw, db)"""
    _L.require_gpu()
    _chk(dy, BF16, "dy")
    if dy2 is not None:
        _chk(dy2, BF16, "dy2")
    M, Cc = a.shape
    n_part = norm_bwd_parts(M)
    parts = [torch.empty((n_part, Cc), dtype=F32, device=a.device) for _ in range(2)]
    dx = torch.empty_like(a)
    dx_a = torch.empty_like(a) if drop_p > 0 else None
    nbytes = M * Cc * 2 * (3 + (r is not None) + (dy2 is not None) + (drop_p > 0))
    _pcall("add_layernorm_bwd", nbytes, "B", "ivh_add_layernorm_bwd", ptr(a), ptr(r), int(gelu), ptr(w), ptr(stats), ptr(dy), ptr(dy2), M, Cc, ptr(dx),
           ptr(dx_a), ptr(parts[0]), ptr(parts[1]), float(drop_p), int(seed) & 0xFFFFFFFF, stream_ptr())
    dw, db = colsum_finish_multi(parts)
    return ((dx, dx_a) if drop_p > 0 else dx), dw, db


def ce_rows(logits: torch.Tensor, labels: torch.Tensor, V: Optional[int] = None, ignore_index: int = -100, dscale: float = 1.0,
            want_grad: bool = True, dscale_dev: Optional[torch.Tensor] = None, inplace: bool = False):
    """mean cross entropy over the rows whose label != ignore_index.  logits bf16|fp32 [M, ld] (columns >= V are padding);
    -> (loss fp32 [1], dlogits bf16 [M, ld] | None = dscale * dscale_dev * d loss / d logits; inplace: written over the bf16 logits)"""
    _L.require_gpu()
    if logits.dtype not in (BF16, F32) or logits.dim() != 2 or logits.stride(1) != 1:
        raise InternVideoHipError("ce_rows: logits must be a bf16 / fp32 matrix with contiguous rows")
    M, ld = logits.shape[0], logits.stride(0)
    V = logits.shape[1] if V is None else int(V)
    lab = _ids32(labels, "labels")
    if lab.numel() != M:
        raise InternVideoHipError("ce_rows: one label per row")
    if dscale_dev is not None:
        _chk(dscale_dev, F32, "dscale_dev")
    rows = torch.empty((M,), dtype=F32, device=logits.device)
    inv = torch.empty((1,), dtype=F32, device=logits.device)
    dl = None
    if want_grad:
        if inplace:
            if logits.dtype != BF16:
                raise InternVideoHipError("ce_rows: in-place gradients need bf16 logits")
            dl = logits
        else:
            dl = torch.empty((M, ld), dtype=BF16, device=logits.device)
    nbytes = M * V * (logits.element_size() * (2 if want_grad else 1) + (2 if want_grad else 0))
    _pcall("ce_rows", nbytes, "B", "ivh_ce_rows", ptr(logits), int(logits.dtype == F32), ld, M, V, ptr(lab), int(ignore_index), float(dscale), ptr(dscale_dev),
           ptr(inv), ptr(rows), ptr(dl), ld, stream_ptr())
    return sum_rows(rows, 1.0), dl


# ---- optimizer --------------------------------------------------------------------------------------------------------
def adamw_step(master, exp_avg, exp_avg_sq, grad, shadow, lr, beta1, beta2, eps, weight_decay, step,
               grad_scale: float = 1.0, clip_coef: Optional[torch.Tensor] = None, lr_segments=None, seg_base: int = 0) -> None:
    """lr_segments = (seg_end int64 [nseg], seg_scale fp32 [nseg]) device tensors: per-segment lr scale (layer-wise lr decay); seg_base =
    offset of this slice inside the region the table describes."""
    _L.require_gpu()
    nbytes = master.numel() * (24 + (2 if grad.dtype == BF16 else 4) + (2 if shadow is not None else 0))
    if lr_segments is not None:
        se, sc = lr_segments
        _chk(se, torch.int64, "seg_end"); _chk(sc, F32, "seg_scale")
        _pcall("adamw_step", nbytes, "B", "ivh_adamw_step_scaled", ptr(master), ptr(exp_avg), ptr(exp_avg_sq), ptr(grad), int(grad.dtype == BF16), ptr(shadow),
               master.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale),
               ptr(clip_coef), ptr(se), ptr(sc), int(se.numel()), int(seg_base), stream_ptr())
        return
    _pcall("adamw_step", nbytes, "B", "ivh_adamw_step", ptr(master), ptr(exp_avg), ptr(exp_avg_sq), ptr(grad), int(grad.dtype == BF16), ptr(shadow),
           master.numel(), float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale),
           ptr(clip_coef), stream_ptr())


def sqnorm(g: torch.Tensor, out: torch.Tensor, accumulate: bool, scratch: Optional[torch.Tensor] = None) -> None:
    _L.require_gpu()
    if scratch is None:
        scratch = torch.empty((_L.load().ivh_sqnorm_scratch_floats(),), dtype=F32, device=g.device)
    call("ivh_sqnorm", ptr(g), int(g.dtype == BF16), g.numel(), ptr(scratch), ptr(out), int(accumulate), stream_ptr())


def clip_coef(sumsq: torch.Tensor, max_norm: float):
    _L.require_gpu()
    coef = torch.empty((1,), dtype=F32, device=sumsq.device)
    nrm = torch.empty((1,), dtype=F32, device=sumsq.device)
    call("ivh_clip_coef", ptr(sumsq), float(max_norm), ptr(coef), ptr(nrm), stream_ptr())
    return coef, nrm


def shard_sum_bf16(recv: torch.Tensor, W: int, out: torch.Tensor) -> None:
    """out fp32 [chunk] = sum over the W bf16 chunks of recv [W * chunk], in rank order (fp32 accumulation of an all-to-all's pieces)"""
    _L.require_gpu()
    _chk(recv, BF16, "recv"); _chk(out, F32, "out")
    chunk = recv.numel() // W
    if recv.numel() != W * chunk or out.numel() != chunk:
        raise InternVideoHipError("shard_sum_bf16: recv must hold W chunks of out.numel() elements")
    call("ivh_shard_sum_bf16", ptr(recv), int(W), chunk, ptr(out), stream_ptr())


# ---- stage-2 contrastive ------------------------------------------------------------------------------------------------
def vtc_loss_fwd_bwd(v: torch.Tensor, t: torch.Tensor, idx: Optional[torch.Tensor], temp, want_grad: bool = True):
    """v, t fp32 [n,C] (already gathered over ranks), idx int64 [n] | None -> (loss[1], sim[n,n], dv, dt, dtemp[1]).
    temp: a Python float, or a 0-dim / 1-element fp32 tensor in HBM that the kernels read themselves (no host synchronisation)"""
    _L.require_gpu()
    _chk(v, F32, "v"); _chk(t, F32, "t")
    v = v.contiguous(); t = t.contiguous()
    n, Cc = v.shape
    if idx is not None:
        idx = idx.to(torch.int64).contiguous()
    ws = torch.empty((_L.load().ivh_vtc_workspace_floats(n, Cc),), dtype=F32, device=v.device)
    sim = torch.empty((n, n), dtype=F32, device=v.device)
    loss = torch.empty((1,), dtype=F32, device=v.device)
    dtemp = torch.empty((1,), dtype=F32, device=v.device)
    dv = torch.empty_like(v) if want_grad else None
    dt = torch.empty_like(t) if want_grad else None
    if isinstance(temp, torch.Tensor) and temp.is_cuda:
        td = temp.detach().reshape(1)
        if td.dtype != F32:
            td = td.float()
        call("ivh_vtc_loss_fwd_bwd_dev", ptr(v), ptr(t), ptr(idx), n, Cc, ptr(td.contiguous()), ptr(sim), ptr(loss), ptr(dv), ptr(dt), ptr(dtemp), ptr(ws),
             stream_ptr())
    else:
        call("ivh_vtc_loss_fwd_bwd", ptr(v), ptr(t), ptr(idx), n, Cc, float(temp), ptr(sim), ptr(loss), ptr(dv), ptr(dt), ptr(dtemp), ptr(ws), stream_ptr())
    return loss, sim, dv, dt, dtemp


def abt_f32(a: torch.Tensor, b: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """out[i, j] = alpha * <a[i], b[j]>, fp32 (ivh_vtc_abt): a [ni, K], b [nj, K] -> [ni, nj]"""
    _L.require_gpu()
    _chk(a, F32, "a"); _chk(b, F32, "b")
    a = a.contiguous(); b = b.contiguous()
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1]:
        raise InternVideoHipError(f"abt_f32: {tuple(a.shape)} x {tuple(b.shape)}^T")
    out = torch.empty((a.shape[0], b.shape[0]), dtype=F32, device=a.device)
    call("ivh_vtc_abt", ptr(a), ptr(b), a.shape[0], b.shape[0], a.shape[1], float(alpha), ptr(out), stream_ptr())
    return out


# ---- probes ---------------------------------------------------------------------------------------------------------------
def probe_tr16(inp: torch.Tensor) -> torch.Tensor:
    _L.require_gpu()
    out = torch.empty((64, 4), dtype=torch.int16, device=inp.device)
    call("ivh_probe_tr16", ptr(inp), ptr(out), stream_ptr())
    return out


def probe_mfma16(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _L.require_gpu()
    c = torch.empty((16, 16), dtype=F32, device=a.device)
    call("ivh_probe_mfma16", ptr(a), ptr(b), ptr(c), stream_ptr())
    return c


def probe_mfma32(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a, b bf16 [32, 16] -> fp32 [32, 32] = a b^T through one 32x32x16 MFMA and the C layout the attention kernels assume"""
    _L.require_gpu()
    c = torch.empty((32, 32), dtype=F32, device=a.device)
    call("ivh_probe_mfma32", ptr(a), ptr(b), ptr(c), stream_ptr())
    return c


def probe_mfma_rate(iters: int, workgroups: int = 256) -> float:
    """FLOPs of one launch of the known-rate MFMA stream (enqueued on the current stream): the calibration reference of tools/pmc_mfma.py"""
    _L.require_gpu()
    sink = torch.zeros((1,), dtype=F32, device="cuda")
    call("ivh_probe_mfma_rate", int(iters), int(workgroups), ptr(sink), stream_ptr())
    return 2.0 * 32 * 32 * 16 * 8 * iters * 4 * workgroups


def probe_mfma_rate2(shape: int, waves_per_simd: int, iters: int, workgroups: int = 256) -> float:
    """FLOPs of one launch of the MFMA stream on shape 0 (32x32x16) / 1 (16x16x32) with 1 or 2 waves per SIMD"""
    _L.require_gpu()
    sink = torch.zeros((1,), dtype=F32, device="cuda")
    call("ivh_probe_mfma_rate2", int(shape), int(waves_per_simd), int(iters), int(workgroups), ptr(sink), stream_ptr())
    return 262144.0 * iters * 4 * waves_per_simd * workgroups
