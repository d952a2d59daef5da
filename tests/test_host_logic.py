"""CPU-only checks: the C-ABI library builds/loads and exports every symbol the header declares, host-side logic
(state_dict contract, mask compaction, engine layout, 2-rank gloo gradient reduction) works without a GPU, and the product
refuses to compute on the CPU."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from internvideo_amd import lib as L  # noqa: E402
from internvideo_amd import internvideo2_pretrain as M  # noqa: E402
from oracle import internvideo2_oracle as O  # noqa: E402


def _tiny(name="tiny64", **kw):
    cfg = O.named_config(name)
    m = M.PretrainInternVideo2(
        img_size=cfg.img_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
        num_frames=cfg.num_frames, attn_pool_num_heads=cfg.attn_pool_num_heads, clip_embed_dim=cfg.clip_embed_dim,
        clip_teacher_embed_dim=cfg.clip_teacher_embed_dim, clip_teacher_final_dim=cfg.clip_teacher_final_dim,
        clip_return_layer=cfg.clip_return_layer, mae_teacher_embed_dim=cfg.mae_teacher_embed_dim,
        mae_return_layer=cfg.mae_return_layer, **kw)
    return cfg, m


def test_library_builds_loads_and_exports_every_declared_symbol():
    from internvideo_amd.csrc import build as b
    path = b.build()
    assert os.path.isfile(path)
    lib = L.load()
    header = open(os.path.join(ROOT, "include", "internvideo_hip.h")).read()
    debug_header = open(os.path.join(ROOT, "include", "internvideo_hip_debug.h")).read()
    product = set(re.findall(r"\b(ivh_[a-z0-9_]+)\s*\(", header))
    product.discard("ivh_gemm_desc")
    debug = set(re.findall(r"\b(ivh_[a-z0-9_]+)\s*\(", debug_header)) - product
    assert product and debug, "no declarations parsed"
    # the product header is the drop-in contract INTEGRATION.md binds: no measurement hook or hardware probe in it
    assert not [n for n in product if "debug" in n or "probe" in n], sorted(n for n in product if "debug" in n or "probe" in n)
    assert all("debug" in n or "probe" in n for n in debug), sorted(debug)
    declared = product | debug
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
    assert declared == set(L.SIGNATURES), (declared ^ set(L.SIGNATURES))
    assert lib.ivh_version() >= 100


def test_no_cpu_compute_path():
    cfg, m = _tiny()
    video, mask, _ = O.synthetic_batch(cfg, 1, 4, seed=0)
    with pytest.raises(L.InternVideoHipError):
        m(video, torch.from_numpy(mask))
    from internvideo_amd import ops
    with pytest.raises(L.InternVideoHipError):
        ops.gemm(torch.zeros(8, 8, dtype=torch.bfloat16), torch.zeros(8, 8, dtype=torch.bfloat16))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "internvideo_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_state_dict_contract_and_registry():
    for name in ("tiny64", "tiny88"):
        cfg, m = _tiny(name)
        shapes = O.param_shapes(cfg)                       # == the reference's keys/shapes (tests/test_oracle_golden pins them)
        sd = m.state_dict()
        assert set(sd) == set(shapes)
        assert all(tuple(sd[k].shape) == shapes[k] for k in shapes)
        m.load_state_dict(O.synthetic_params(cfg, seed=0), strict=True)
    assert "pretrain_internvideo2_1B_patch14_224" in M._registry and "pretrain_internvideo2_6B_patch14_224" in M._registry
    cfg, m = _tiny()
    assert m.get_num_layers() == cfg.depth and m.dtype == torch.float32 and m.patch_embed.patch_size == (14, 14)
    assert {"pos_embed", "cls_token", "clip_pos_embed", "mae_pos_embed"} <= m.no_weight_decay()
    # reference init statistics (P:588-603): proj / fc2 of block i scaled by 1/sqrt(2(i+1)), LayerScale 1e-5, sincos tables
    assert abs(m.blocks[2].ls1.gamma[0].item() - 1e-5) < 1e-12
    r = m.blocks[0].mlp.fc2.weight.std() / m.blocks[2].mlp.fc2.weight.std()
    assert 1.5 < r.item() < 2.0                            # sqrt(6)/sqrt(2) = 1.73
    assert np.array_equal(m.pos_embed[0].detach().numpy(), O.sincos_pos_embed_3d(cfg.embed_dim, 4, 4, True).astype(np.float32))
    with pytest.raises(AssertionError):
        _tiny(use_flash_attn=False)                        # flag-consistency assert of the reference (P:446-447)


def test_host_mask_compaction_bit_exact_and_ragged_error():
    cfg = O.named_config("S14")
    _, mask, _ = O.synthetic_batch(cfg, 3, 16, seed=5)
    vis, inv = M.build_gather_indices(torch.from_numpy(mask), "cpu")
    assert np.array_equal(vis.numpy(), O.visible_indices(mask))
    for b in range(3):
        assert np.array_equal(inv[b].numpy()[vis[b].numpy()], np.arange(vis.shape[1]))
        assert (inv[b] >= 0).sum().item() == vis.shape[1]
    # tube / random masks of the reference generators under the reference's numpy seeding
    tm = O.tube_mask((4, 8, 8), 0.75, np.random.RandomState(0)).astype(bool)
    full = np.concatenate([[False], tm])[None]
    v, _ = M.build_gather_indices(torch.from_numpy(full), "cpu")
    assert v.shape[1] == 1 + 4 * 16
    bad = mask.copy(); bad[1, 5] = not bad[1, 5]
    with pytest.raises(RuntimeError):
        M.build_gather_indices(torch.from_numpy(bad), "cpu")


def test_engine_layout_is_backward_ordered_and_views_alias_flat_buffers():
    from internvideo_amd.engine import IVTrainEngine
    cfg, m = _tiny("tiny88")
    ref = {k: v.clone() for k, v in m.state_dict().items()}
    eng = IVTrainEngine(m)
    for k, v in m.state_dict().items():                   # re-pointing must not change any value
        assert torch.equal(v, ref[k]), k
    names = [n for n, _ in eng.mat_params]
    first_block = next(i for i, n in enumerate(names) if n.startswith("blocks."))
    assert all(not n.startswith("blocks.") for n in names[:first_block])            # heads first
    blk = [int(n.split(".")[1]) for n in names if n.startswith("blocks.")]
    assert blk == sorted(blk, reverse=True)                                        # last block first
    assert names[-1].startswith("patch_embed")
    assert all(p.dim() == 1 or n.endswith(".bias") or n in m.no_weight_decay() for n, p in eng.vec_params)
    assert all(p.dim() >= 2 for _, p in eng.mat_params)
    w = m.blocks[1].attn.qkv.weight
    w.data.fill_(3.0)
    off = eng.mat_off[names.index("blocks.1.attn.qkv.weight")]
    assert torch.all(eng.master[off:off + w.numel()] == 3.0)
    assert w.main_grad.data_ptr() == eng.grad_mat[off:].data_ptr() and w._ivh_bf16.dtype == torch.bfloat16
    ends = [eng.block_end[i] for i in range(cfg.depth - 1, -1, -1)]
    assert ends == sorted(ends) and eng.head_end <= ends[0]
    # engine.close() (before dist.destroy_process_group()): captured graphs and the segment chain are released, the engine stays usable
    eng._graph, eng._segments, eng._graph_out = object(), [object()], (1, 2)
    eng.close()
    assert eng._graph is None and eng._segments is None and eng._graph_out is None
    eng.zero_grad()


def test_a_dropped_engine_goes_with_its_last_reference():
    """The model's hooks hold the engine weakly: no model -> hook -> engine -> model cycle, so a dropped engine (flat buffers, HIP graph, the
    graph's private pool) is freed by reference counting at once instead of waiting for the cyclic collector -- which might run while a
    later engine is capturing (engine._cyclic_gc_paused).  The hooks of a dead engine are no-ops."""
    import gc
    import weakref
    from internvideo_amd.engine import IVTrainEngine
    cfg, m = _tiny("tiny64")
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        eng = IVTrainEngine(m)
        hook = eng._block_hook()
        r = weakref.ref(eng)
        del eng
        assert r() is None, "the engine is kept alive by a reference cycle: " + repr(gc.get_referrers(r())[:3] if r() is not None else None)
    finally:
        if was:
            gc.enable()
    assert hook(0) is None
    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})          # the dead engine's post-hook does nothing


def test_capture_guard_pauses_the_cyclic_collector_and_restores_it():
    """engine._cyclic_gc_paused: garbage that exists is collected on entry, nothing is collected inside, the collector's previous state comes
    back on exit -- also when the body raises, also when it was off to begin with, also nested"""
    import gc
    import weakref
    from internvideo_amd.engine import _cyclic_gc_paused

    class Node:
        pass

    def cycle():
        a, b = Node(), Node()
        a.o, b.o = b, a
        return weakref.ref(a)

    assert gc.isenabled()
    old = cycle()
    with _cyclic_gc_paused():
        assert old() is None and not gc.isenabled()           # collected on entry
        young = cycle()
        for _ in range(5000):                                  # enough allocations for several automatic collections, were they allowed
            [Node() for _ in range(10)]
        assert young() is not None
        with _cyclic_gc_paused():
            assert not gc.isenabled()
        assert not gc.isenabled()                              # the inner guard restores "off"
    assert gc.isenabled()
    gc.collect()
    assert young() is None
    with pytest.raises(KeyError):
        with _cyclic_gc_paused():
            raise KeyError("x")
    assert gc.isenabled()


def test_bf16_copies_follow_checkpoint_loads():
    """a checkpoint loaded AFTER the engine / the first teacher forward must reach the bf16 matrices the GEMMs read.
    load_state_dict copies in place (data_ptr unchanged): the engine refreshes its shadow from a post-hook, the frozen teachers key
    their bf16 cache on the parameters' version counters."""
    from internvideo_amd.engine import IVTrainEngine
    cfg, m = _tiny("tiny64")
    eng = IVTrainEngine(m)
    sd = {k: torch.randn_like(v) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)                                   # the reference resume order: engine first, checkpoint second (utils.py:568-647)
    w = m.blocks[0].attn.qkv.weight
    assert torch.equal(w.detach(), sd["blocks.0.attn.qkv.weight"])
    assert torch.equal(w._ivh_bf16, w.detach().to(torch.bfloat16).reshape(w.shape[0], -1)), "shadow is stale after model.load_state_dict"
    w.data.mul_(2.0)                                        # manual edits: the caller syncs
    eng.sync_shadow()
    assert torch.equal(w._ivh_bf16, w.detach().to(torch.bfloat16).reshape(w.shape[0], -1))
    from internvideo_amd import internvl_clip_vision as T, videomae_teacher as V
    clip = T.InternVL_CLIP(img_size=28, patch_size=14, embed_dim=64, depth=1, num_heads=2, mlp_ratio=2, clip_embed_dim=32, attn_pool_num_heads=2,
                           clip_return_layer=1)
    mae = V.VisionTransformer(img_size=32, patch_size=16, embed_dim=64, depth=1, num_heads=2, mlp_ratio=2, all_frames=4, tubelet_size=2)
    for teacher in (clip, mae):
        teacher._bf16_weights()
        p = next(q for q in teacher.parameters() if q.dim() >= 2 and q.requires_grad is not None)
        old = p._ivh_bf16.clone()
        teacher.load_state_dict({k: torch.randn_like(v) for k, v in teacher.state_dict().items()})
        teacher._bf16_weights()
        assert not torch.equal(p._ivh_bf16, old) and torch.equal(p._ivh_bf16.reshape(-1), p.detach().to(torch.bfloat16).reshape(-1)), type(teacher).__name__


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from internvideo_amd import internvideo2_pretrain as M
from internvideo_amd.engine import IVTrainEngine
from oracle import internvideo2_oracle as O
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
cfg = O.named_config("tiny64")
torch.manual_seed(0)
m = M.PretrainInternVideo2(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
    num_frames=cfg.num_frames, attn_pool_num_heads=cfg.attn_pool_num_heads, clip_embed_dim=cfg.clip_embed_dim,
    clip_teacher_embed_dim=cfg.clip_teacher_embed_dim, clip_teacher_final_dim=cfg.clip_teacher_final_dim,
    clip_return_layer=cfg.clip_return_layer, mae_teacher_embed_dim=cfg.mae_teacher_embed_dim, mae_return_layer=cfg.mae_return_layer)
eng = IVTrainEngine(m, bucket_bytes=64 * 1024, reduce_dtype="bf16")            # small buckets -> several overlapped reductions; in-place bf16 wire sum
assert eng.world == 2 and m.grad_ready_hook is not None
# emulate the backward: every rank fills its gradient buffers with rank-dependent values, block hooks fire last block first
eng.zero_grad()
g = torch.Generator().manual_seed(100 + rank)
local_mat = torch.randn(eng.n_mat, generator=g).to(torch.bfloat16)
local_vec = torch.randn(eng.n_vec, generator=g)
eng.grad_mat.copy_(local_mat); eng.grad_vec.copy_(local_vec)
for i in range(cfg.depth - 1, -1, -1):
    m.grad_ready_hook(i)
eng._finish_reduce()
# reference: plain all_reduce of the same data
ref_mat = local_mat.clone(); ref_vec = local_vec.clone()
dist.all_reduce(ref_mat); dist.all_reduce(ref_vec)
assert torch.equal(eng.grad_mat, ref_mat), "bucketed reduction differs from a single all_reduce"
assert torch.equal(eng.grad_vec, ref_vec)
log = eng.reduce_log
assert len(log) >= 2 and log[0][0] == 0 and log[-1][1] == eng.n_mat
assert all(a[1] == b[0] for a, b in zip(log, log[1:])), log     # contiguous, non-overlapping, in backward order
# the no-overlap variant used after a graph-replayed step (capture_step(defer_reduce=True)): same result, bucket by bucket
eng.grad_mat.copy_(local_mat); eng.grad_vec.copy_(local_vec)
eng.reduce_all_now()
assert torch.equal(eng.grad_mat, ref_mat) and torch.equal(eng.grad_vec, ref_vec), "reduce_all_now differs from a single all_reduce"
# with the deferred mode armed, backward-time hooks and _finish_reduce issue no collective at all
eng._defer_reduce = True
eng.grad_mat.copy_(local_mat); eng.zero_grad()
eng._finish_reduce()
assert torch.equal(eng.grad_mat, local_mat) and eng.reduce_log == []
# ---- reduction precision and the ZeRO-1 path ---------------------------------------------------------------------------
def fresh():
    torch.manual_seed(0)
    return M.PretrainInternVideo2(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
        num_frames=cfg.num_frames, attn_pool_num_heads=cfg.attn_pool_num_heads, clip_embed_dim=cfg.clip_embed_dim,
        clip_teacher_embed_dim=cfg.clip_teacher_embed_dim, clip_teacher_final_dim=cfg.clip_teacher_final_dim,
        clip_return_layer=cfg.clip_return_layer, mae_teacher_embed_dim=cfg.mae_teacher_embed_dim, mae_return_layer=cfg.mae_return_layer)
exact = local_mat.double().clone(); dist.all_reduce(exact)             # the true sum of the ranks' bf16 gradients
# (1) all-reduce with fp32 accumulation: the bucket is widened first, the sum is exact to fp32 rounding
e32 = IVTrainEngine(fresh(), bucket_bytes=64 * 1024, reduce_dtype="fp32")
assert e32.n_mat == eng.n_mat and e32.grad_comm32 is not None
e32.zero_grad(); e32.grad_mat.copy_(local_mat); e32.grad_vec.copy_(local_vec)
for i in range(cfg.depth - 1, -1, -1):
    e32.model.grad_ready_hook(i)
e32._finish_reduce()
err32 = ((e32.grad_comm32.double() - exact).norm() / exact.norm()).item()
err16 = ((ref_mat.double() - exact).norm() / exact.norm()).item()
assert err32 < 1e-7, err32
assert 1e-4 < err16 < 6e-3, err16                                       # what the bf16 wire sum loses: ~2^-9 relative per element
# (2) ZeRO-1: all-to-all of bf16 shards + fp32 accumulation on the owner; every bucket splits evenly over the ranks
ez = IVTrainEngine(fresh(), bucket_bytes=64 * 1024, reduce_mode="zero1")
assert ez.zero1 and ez.n_mat % (1024 * world) == 0 and all((hi - lo) % (64 * world) == 0 for lo, hi in ez.buckets)
assert ez.buckets[0][0] == 0 and ez.buckets[-1][1] == ez.n_mat and all(a[1] == b[0] for a, b in zip(ez.buckets, ez.buckets[1:]))
gz = torch.Generator().manual_seed(200 + rank)
zl = torch.randn(ez.n_mat, generator=gz).to(torch.bfloat16)
zexact = zl.double().clone(); dist.all_reduce(zexact)
ez.zero_grad(); ez.grad_mat.copy_(zl); ez.grad_vec.copy_(local_vec)
for i in range(cfg.depth - 1, -1, -1):
    ez.model.grad_ready_hook(i)
ez._finish_reduce()
assert ez.reduce_log == ez.buckets and len(ez.buckets) >= 2
for lo, hi in ez.buckets:
    s0, c = ez._shard(lo, hi)
    got = ez.grad_shard32[lo // world:lo // world + c].double()
    assert (got - zexact[s0:s0 + c]).abs().max().item() <= 1e-6 * zexact.abs().max().item(), "zero1 shard sum is not the fp32 sum"
assert torch.equal(ez.grad_vec, ref_vec)
# the bf16 compute copy after the (emulated) sharded update: every rank writes its shard, one all-gather per bucket rebuilds the rest
for lo, hi in ez.buckets:
    s0, c = ez._shard(lo, hi)
    ez.shadow[lo:hi].zero_()
    ez.shadow[s0:s0 + c] = torch.arange(s0, s0 + c, dtype=torch.float32).remainder(251).to(torch.bfloat16)
ez._gather_buckets(ez.shadow)
assert torch.equal(ez.shadow, torch.arange(ez.n_mat, dtype=torch.float32).remainder(251).to(torch.bfloat16))
# fp32 state of the matrix region is rank-local until consolidate() (checkpointing) gathers it
for lo, hi in ez.buckets:
    s0, c = ez._shard(lo, hi)
    ez.master[lo:hi].zero_(); ez.master[s0:s0 + c] = float(rank + 1)
ez.step_count += 1                                      # as after an optimizer step: shards differ between the ranks
for read in (ez.state_dict, ez.model.state_dict):       # a rank-local read of sharded state must fail loudly, not deadlock or mix weights
    try:
        read()
        raise AssertionError("sharded zero1 state was readable without consolidate()")
    except RuntimeError as e:
        assert "consolidate" in str(e)
ez.consolidate()                                        # the collective, on every rank ...
if rank == 0:                                           # ... then rank 0 alone may read (the reference's save_on_master pattern)
    assert ez.state_dict()["step"] == ez.step_count and "pos_embed" in ez.model.state_dict()
for lo, hi in ez.buckets:
    c = (hi - lo) // world
    for r in range(world):
        assert torch.all(ez.master[lo + r * c:lo + (r + 1) * c] == float(r + 1))
# the reference's per-step NaN / Inf guard (engine_for_pretraining.py:151-161): the losses of all ranks are gathered, any bad one stops every rank
eg = IVTrainEngine(fresh(), check_finite=True)
eg._guard_finite(torch.tensor(1.0 + rank))
assert abs(eg.all_loss_mean - 1.5) < 1e-6
try:
    eg._guard_finite(torch.tensor(float("nan") if rank == 1 else 2.0))
    raise AssertionError("NaN on rank 1 did not stop rank %d" % rank)
except SystemExit as e:
    assert e.code == 1
print("RANK", rank, "OK", len(log), "buckets", "bf16-sum err %.1e fp32-sum err %.1e" % (err16, err32))
dist.destroy_process_group()
"""


def test_two_rank_gloo_bucketed_gradient_reduction(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("OK") == 2, r.stdout


_WORKER_S2 = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
from internvideo_amd import stage2
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
args = stage2._gather_args()
assert args.world_size == 2 and args.rank == rank
x = (torch.arange(12, dtype=torch.float32).reshape(3, 4) + 100 * rank).requires_grad_(True)
y = stage2.allgather_wgrad(x, args)                       # multi_modality/models/utils.py:193-212
assert y.shape == (6, 4)
for r in range(world):
    assert torch.equal(y[3 * r:3 * r + 3], torch.arange(12, dtype=torch.float32).reshape(3, 4) + 100 * r)   # rank order
w = torch.arange(24, dtype=torch.float32).reshape(6, 4)
(y * w).sum().backward()
assert torch.equal(x.grad, w[3 * rank:3 * rank + 3])       # backward = local slice, no cross-rank reduction
# the packed [v | t | idx] exchange of VTC_VTM_Loss.vtc_loss, with the HIP loss kernel replaced by a recorder (no GPU here)
seen = {{}}
class Rec:
    @staticmethod
    def apply(v, t, idx, temp):
        seen.update(v=v, t=t, idx=idx)
        return (v.sum() + t.sum()) * 0
stage2._VTCFn = Rec
C = 8
g = torch.Generator().manual_seed(rank)
v = torch.randn(3, C, generator=g); t = torch.randn(3, C, generator=g)
idx = torch.tensor([5, (1 << 40) + 7 + rank, 123456789012 + rank])
stage2.VTC_VTM_Loss(False).vtc_loss(v, t, idx, 0.07, all_gather=True)
vs = [torch.empty_like(v) for _ in range(world)]; dist.all_gather(vs, v)
ts = [torch.empty_like(t) for _ in range(world)]; dist.all_gather(ts, t)
ids = [torch.empty_like(idx) for _ in range(world)]; dist.all_gather(ids, idx)
assert torch.equal(seen["v"], torch.cat(vs)) and torch.equal(seen["t"], torch.cat(ts))
assert torch.equal(seen["idx"], torch.cat(ids)), (seen["idx"], torch.cat(ids))     # 64-bit ids survive the float packet
# ---- the training engine on the stage-2 model, two ranks: the bucketed reduction of the flat buffers (text tower first, released by the
# vision tower's per-block hook) and the ZeRO-1 shard plan -- the gradients a rank ends up with are the sum over both ranks
from internvideo_amd.engine import IVTrainEngine
{build}
for mode in ("allreduce", "zero1"):
    model, scfg = tiny_stage2()
    eng = IVTrainEngine(model, bucket_bytes=32 * 1024, reduce_mode=mode, reduce_dtype="bf16")
    assert eng.comm and eng.world == 2 and len(eng.buckets) >= 3
    gg = torch.Generator().manual_seed(50 + rank)
    lm = torch.randn(eng.n_mat, generator=gg).to(torch.bfloat16); lv = torch.randn(eng.n_vec, generator=gg)
    exact = lm.double().clone(); dist.all_reduce(exact)
    exv = lv.clone(); dist.all_reduce(exv)
    eng.zero_grad(); eng.grad_mat.copy_(lm); eng.grad_vec.copy_(lv)
    depth = len(model.vision_encoder.blocks)
    for i in range(depth - 1, -1, -1):
        eng.tower.grad_ready_hook(i)
        if i == depth - 1:
            assert eng.reduce_log and eng.reduce_log[-1][1] >= eng.head_end     # the whole text / heads region went out with the first hook
    eng._finish_reduce()
    assert eng.reduce_log == eng.buckets and torch.equal(eng.grad_vec, exv)
    if mode == "allreduce":
        assert ((eng.grad_mat.double() - exact).abs() <= 2 ** -7 * exact.abs() + 1e-6).all()     # the bf16 wire sum of two ranks
    else:
        for lo, hi in eng.buckets:
            s0, c = eng._shard(lo, hi)
            assert ((eng.grad_shard32[lo // world:lo // world + c].double() - exact[s0:s0 + c]).abs() <= 1e-6 * exact.abs().max()).all()
print("RANK", rank, "OK")
dist.destroy_process_group()
"""


def test_two_rank_gloo_stage2_allgather_and_packed_exchange(tmp_path):
    script = tmp_path / "worker_s2.py"
    script.write_text(_WORKER_S2.format(root=ROOT, build=_STAGE2_BUILD))
    port = 31500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("OK") == 2, r.stdout


def test_torch_ops_registration_meta_kernels_and_no_cpu_path():
    """torch.ops.internvideo_hip.*: schemas registered, Meta kernels give the shapes / dtypes the HIP kernels produce, and there is no
    CPU implementation to fall back to."""
    import internvideo_amd.torch_ops as T
    ns = torch.ops.internvideo_hip
    for name in T.OPERATORS:
        assert hasattr(ns, name), name
    a = torch.empty((96, 64), dtype=torch.bfloat16, device="meta")
    w = torch.empty((40, 64), dtype=torch.bfloat16, device="meta")
    y = ns.gemm(a, w, None, "gelu_erf")
    assert y.shape == (96, 40) and y.dtype == torch.bfloat16 and y.device.type == "meta"
    assert ns.gemm(a, w, None, "none", True, True, 1.0, True).dtype == torch.float32
    dw = ns.gemm(torch.empty((96, 40), dtype=torch.bfloat16, device="meta"), a, None, "none", False, False)
    assert dw.shape == (40, 64)
    qkv = torch.empty((2 * 16, 3 * 128), dtype=torch.bfloat16, device="meta")
    o, lse = ns.flash_attn_fwd(qkv, 2, 16, 2)
    assert o.shape == (32, 128) and lse.shape == (2, 2, 16) and lse.dtype == torch.float32
    assert ns.flash_attn_bwd(qkv, o, o, lse, 2, 16, 2).shape == qkv.shape
    r, n, s = ns.rmsnorm_add_fwd(torch.empty((32, 128), device="meta"), o, None, None, 16, torch.empty(128, device="meta"), 1e-6)
    assert r.dtype == torch.float32 and n.dtype == torch.bfloat16 and s.shape == (32,)
    with pytest.raises((NotImplementedError, RuntimeError)):
        ns.gemm(torch.zeros((8, 8), dtype=torch.bfloat16), torch.zeros((8, 8), dtype=torch.bfloat16))


def test_weight_gradient_group_size_fills_the_cus():
    """functional._wgrad_group_size: the queued weight-gradient GEMMs of one K are launched when their 256 x 256 tiles fill the last round
    of 256 CUs to 95 % -- 12 problems for the 1B widths, 28 for ViT-B/14, 4 for the 6B widths; never more than 32"""
    from internvideo_amd import functional as Fn
    Fn._N_CU[0] = 256

    def block(D, Hm, rows=1024):
        mk = lambda n, k: (torch.empty((rows, n), dtype=torch.bfloat16, device="meta"), torch.empty((rows, k), dtype=torch.bfloat16, device="meta"), None)  # noqa: E731
        return [mk(Hm, D), mk(D, Hm), mk(D, D), mk(3 * D, D)]            # backward order: fc2, fc1, proj, qkv -> (dy [rows, N_out], x [rows, K_in])
    for D, Hm, want in ((1408, 6144, 12), (768, 3072, 28), (3200, 12800, 4)):
        q = []
        got = 0
        for _ in range(10):
            q += block(D, Hm)
            got = Fn._wgrad_group_size(q)
            if got:
                break
        assert got == want, (D, got)
    q = [b for _ in range(12) for b in block(64, 64)]                     # tiny tiles never fill a round: flushed at the kernel's limit
    assert Fn._wgrad_group_size(q) == 32 and Fn._wgrad_group_size(q[:31]) == 0
    # forced flushes: problems of different K share one launch when that saves rounds (round 6) -- the 1B decoders' weight gradients over
    # B L rows (6 x 3200 x 1408: 468 tiles) and over B (L - 1) rows (8 x 1408 x 1408: 288 tiles) take 3 rounds together instead of 2 + 2;
    # the stage-2 text tower's K = 2048 / 6144 groups (several rounds each) stay apart
    mk = lambda rows, n, k: (torch.empty((rows, n), dtype=torch.bfloat16, device="meta"), torch.empty((rows, k), dtype=torch.bfloat16, device="meta"), None)  # noqa: E731
    dec = [mk(53376, 3200, 1408) for _ in range(6)] + [mk(53248, 1408, 1408) for _ in range(8)]
    assert Fn._mixing_pays(dec)
    text = [mk(2048, 1024, 1024) for _ in range(50)] + [mk(6144, 1024, 4096) for _ in range(14)]
    assert not Fn._mixing_pays(text[:32]) and not Fn._mixing_pays(dec[:6])
    Fn._N_CU[0] = 0


def test_grouped_weight_grads_defer_to_the_end_of_backward_and_accumulate(monkeypatch):
    """functional.grouped_weight_grads: inside it a Linear node queues its weight gradient and returns None to autograd; the autograd
    engine's end-of-pass callback launches the queue (grouped) and accumulates into .grad.  Host bookkeeping only: the three kernels the
    nodes call are emulated with torch on the CPU."""
    from internvideo_amd import functional as Fn, ops

    def gemm(a, b, a_kc=True, b_kc=True, bias=None, out=None, **kw):
        r = (a.float() if a_kc else a.float().t()) @ (b.float() if b_kc else b.float().t()).t()
        r = (r if bias is None else r + bias).to(torch.bfloat16)
        return r if out is None else out.copy_(r)
    launches = []

    def gemm_grouped(problems, a_kc=False, b_kc=False):
        launches.append(len(problems))
        for a, b, out in problems:
            gemm(a, b, a_kc=a_kc, b_kc=b_kc, out=out)
    monkeypatch.setattr(ops, "gemm", gemm); monkeypatch.setattr(ops, "gemm_grouped", gemm_grouped)
    monkeypatch.setattr(ops, "colsum_bf16", lambda x: x.float().sum(0))
    torch.manual_seed(0)
    w1, b1 = torch.nn.Parameter(torch.randn(16, 8)), torch.nn.Parameter(torch.randn(16))
    w2, b2 = torch.nn.Parameter(torch.randn(4, 16)), torch.nn.Parameter(torch.randn(4))
    x = torch.randn(24, 8).to(torch.bfloat16)
    params = (w1, b1, w2, b2)

    def run(ctx, passes=1):
        for q in params:
            q.grad = None
        with ctx:
            for _ in range(passes):
                Fn.LinearFn.apply(Fn.LinearFn.apply(x, w1, b1), w2, b2).float().pow(2).sum().backward()
        return [q.grad.clone() for q in params]
    import contextlib
    node = run(contextlib.nullcontext())
    assert launches == []
    grouped = run(Fn.grouped_weight_grads())
    assert launches == [2] and not Fn._end_pending and not Fn._wgrad_queue          # both weight gradients (same row count) in one launch
    assert all(torch.equal(a, b) for a, b in zip(node, grouped))
    twice = run(Fn.grouped_weight_grads(), passes=2)                                 # .grad accumulates across passes like autograd's
    assert torch.allclose(twice[0], 2 * node[0]) and torch.allclose(twice[2], 2 * node[2]) and Fn._END_DEFER[0] is False
    # a backward pass that raises after a node queued its gradient must not poison the next pass

    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")
    for q in params:
        q.grad = None
    with Fn.grouped_weight_grads():
        h = Boom.apply(Fn.LinearFn.apply(x, w1, b1))       # backward order: second Linear (queues), Boom (raises), first Linear (never runs)
        with pytest.raises(RuntimeError, match="boom"):
            Fn.LinearFn.apply(h, w2, b2).float().pow(2).sum().backward()
    assert w1.grad is None
    again = run(Fn.grouped_weight_grads())
    assert all(torch.equal(a, b) for a, b in zip(node, again)) and not Fn._end_pending and not Fn._wgrad_queue


def test_gemm_tail_split_plan_is_for_long_k_and_mostly_empty_last_rounds():
    """ivh_gemm_split_workspace (host logic of gemm256.hip's tail split, 256 CUs assumed without a device): only problems whose last tile
    round is at most half full AND whose K is long enough to pay for the exchange ask for a workspace -- the B = 32 shapes of the 1B block
    with a 1408-wide output and K = 4224 / 6144 (318 tiles: 62 tail tiles x 4 slices), not the K = 1408 ones, nothing at B = 128"""
    import ctypes as C
    from internvideo_amd import lib
    L = lib.load()

    def units(M, N, K, b_kc=1, fp8=False, **kw):
        d = lib.GemmDesc()
        d.M, d.N, d.K, d.a_kc, d.b_kc, d.batch = M, N, K, 1, b_kc, 1
        d.lda, d.ldb, d.ldc = K, (K if b_kc else N), N
        for k, v in kw.items():
            setattr(d, k, v)
        need = (L.ivh_gemm_fp8_split_workspace if fp8 else L.ivh_gemm_split_workspace)(C.byref(d))
        assert need == 0 or (need - 4096) % 262144 == 0
        return (need - 4096) // 262144 if need else 0
    assert units(13344, 1408, 6144) == 248 and units(13344, 1408, 4224) == 248 and units(13344, 1408, 6144, b_kc=0) == 248
    assert units(13184, 1408, 6144) == 224                                    # stage-2 vision tower: 56 tail tiles x 4
    assert units(13344, 1408, 1408) == 0 and units(13344, 4224, 1408) == 0     # short K / a last round that is more than half full
    assert units(53376, 1408, 6144) == 0 and units(53376, 6144, 1408) == 0     # B = 128: 4.9 and 19.6 rounds
    assert units(2048, 2048, 4096) == 0                                        # 64 tiles: the launch-time model prefers the 128^2 kernel ...
    L.ivh_set_gemm_kernel(2)
    try:
        assert units(2048, 2048, 4096) == 256                                  # ... on the 256^2 kernel every tile is split four ways
    finally:
        L.ivh_set_gemm_kernel(0)
    assert units(13344, 1408, 6144, batch=2) == 0 and units(13344, 1408, 6144, c_fp32=1) == 0
    assert units(1024, 768, 8192, fp8=True) == 48 and units(1024, 768, 1024, fp8=True) == 0
    L.ivh_gemm256_debug_split(0)
    try:
        assert units(13344, 1408, 6144) == 0
    finally:
        L.ivh_gemm256_debug_split(1)


def test_bench_keeps_stdout_for_its_one_json_line(tmp_path):
    """bench.py's contract is ONE JSON line on stdout; RCCL prints a version banner through C stdio when a communicator is created (it lands
    after our line).  bench._reserve_stdout() must leave only what is written to the returned descriptor on stdout."""
    import subprocess
    import sys as _sys
    code = (
        "import ctypes, os, sys\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "import bench\n"
        "fd = bench._reserve_stdout()\n"
        "ctypes.CDLL(None).printf(b'RCCL version : banner through C stdio\\n')\n"
        "print('a stray python print')\n"
        "os.write(fd, b'{\"value\": 1}\\n')\n"
    )
    r = subprocess.run([_sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout == '{"value": 1}\n', repr(r.stdout)
    assert "banner through C stdio" in r.stderr and "a stray python print" in r.stderr


def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` with no launcher around it (the driver's N > 1 form) starts its own ranks (torch.distributed.run, 127.0.0.1,
    a free port), runs the barrier / max-over-ranks timing protocol and prints exactly ONE JSON line from rank 0 -- on gloo, no GPU.  Launch
    contract of the reference: InternVideo2/multi_modality/torchrun.sh:13, single_modality/utils.py:332-373."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["dry_run"] is True and out["steps"] == 3 and out["warmup"] == 1
    # the same file under an external launcher (the documented driver form) behaves identically
    port = 29500 + (os.getpid() % 2000) + 7
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                         "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "2",
                         "--warmup", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r2.returncode == 0, r2.stderr[-2000:]
    out2 = [json.loads(ln) for ln in r2.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(out2) == 1 and out2[0]["n_gpus"] == 2
    # a mismatch between --gpus and the launcher's world size is an error, not a silent single-rank run
    r3 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                         "--master-port", str(port + 1), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run", "--backend", "gloo"],
                        capture_output=True, text=True, timeout=300, env=env)
    assert r3.returncode != 0


def test_constructor_variants_of_the_b1_contract_build_on_the_host():
    """`sep_pos_embed=True` and `norm_type='none'` (P:358-363, 479-495): constructor kwargs of the reference that no shipped recipe sets.  The
    module builds without a GPU, exposes the reference's parameter names / shapes (tests/golden/variants.npz was produced by loading exactly
    these names into the reference's own module) and initialises the separable tables like P:562-577."""
    from internvideo_amd.pos_embed import get_1d_sincos_pos_embed, get_2d_sincos_pos_embed
    cfg = O.named_config("tiny64")
    m = M.PretrainInternVideo2(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
                               mlp_ratio=cfg.mlp_ratio, num_frames=cfg.num_frames, sep_pos_embed=True, clip_norm_type="none", mae_norm_type="none")
    gold = np.load(os.path.join(ROOT, "tests", "golden", "variants.npz"))
    sd = m.state_dict()
    for k in [f[7:] for f in gold.files if f.startswith("sep:in:")]:
        assert k in sd and tuple(sd[k].shape) == gold["sep:in:" + k].shape, k
    assert not any(k in sd for k in ("pos_embed", "clip_pos_embed", "mae_pos_embed"))
    g = m.patch_embed.grid_size
    assert np.allclose(m.pos_embed_spatial[0].detach().numpy(), get_2d_sincos_pos_embed(cfg.embed_dim, g[1]), atol=1e-6)
    assert np.allclose(m.mae_pos_embed_temporal[0].detach().numpy(), get_1d_sincos_pos_embed(cfg.embed_dim, g[0]), atol=1e-6)
    assert float(m.clip_pos_embed_cls.abs().max()) == 0.0
    joint = m._pos_table("clip_")
    assert tuple(joint.shape) == (1, 1 + g[0] * g[1] * g[2], cfg.embed_dim)
    t, hw = 1, 3                                                            # token (t, hw) of the joint table = spatial[hw] + temporal[t]
    assert torch.allclose(joint[0, 1 + t * g[1] * g[2] + hw], m.clip_pos_embed_spatial[0, hw] + m.clip_pos_embed_temporal[0, t])
    assert m.clip_decoder[0].norm_type == "none" and m.mae_decoder[0].norm_type == "none"
    with pytest.raises(NotImplementedError):
        M.MLP_Decoder(norm_type="l1")


def _tiny_finetune_classifier():
    from internvideo_amd import internvideo2 as FT
    from oracle import internvideo2_oracle as O
    cfg = O.named_config("tiny88")
    m = FT.InternVideo2(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
                        num_frames=cfg.num_frames, drop_path_rate=0.0, attn_pool_num_heads=cfg.attn_pool_num_heads,
                        clip_embed_dim=cfg.clip_embed_dim, num_classes=10)
    m.load_state_dict(O.synthetic_finetune_params(cfg, 10, seed=12), strict=True)
    return m


def test_layer_wise_lr_decay_groups_match_the_reference_and_become_flat_segments():
    """single_modality/optim_factory.py:24-98 + run_finetuning.py:548-549 through tests/golden/layer_decay.npz (the reference's own
    LayerDecayValueAssigner / get_parameter_groups run on these parameter names): layer id, lr_scale and weight decay of every parameter;
    and the engine's flat-buffer tables (one (end, scale) run per region) say the same thing element by element."""
    from internvideo_amd import schedules as S
    from internvideo_amd.engine import IVTrainEngine
    g = np.load(os.path.join(ROOT, "tests", "golden", "layer_decay.npz"))
    layer_decay = float(g["meta"][0])
    m = _tiny_finetune_classifier()
    depth = m.get_num_layers()
    a = S.LayerDecayValueAssigner.for_depth(depth, layer_decay)
    named = list(m.named_parameters())
    groups = S.parameter_groups(named, float(g["meta"][2]), m.no_weight_decay(), a.get_layer_id, a.get_scale)
    where = {n: gr for gr in groups.values() for n in gr["params"]}
    for i, (n, p) in enumerate(named):
        assert str(g[f"name:{i}"]) == n
        assert a.get_layer_id(n) == int(g[f"layer:{i}"]), n
        assert where[n]["lr_scale"] == float(g[f"scale:{i}"]) and where[n]["weight_decay"] == float(g[f"wd:{i}"]), n
    assert len(groups) == 10                                      # what the reference printed for this model: layers 0..4 x {decay, no_decay}
    eng = IVTrainEngine(m, layer_decay=layer_decay)
    scale_of = {str(g[f"name:{i}"]): float(g[f"scale:{i}"]) for i in range(len(named))}
    for items, offs, tab, total in ((eng.mat_params, eng.mat_off, eng._lr_seg_mat, eng.n_mat), (eng.vec_params, eng.vec_off, eng._lr_seg_vec, eng.n_vec)):
        ends, scales = tab[0].tolist(), tab[1].tolist()
        assert ends == sorted(ends) and ends[-1] == total and all(e % 64 == 0 for e in ends) and len(ends) <= depth + 2
        for (n, p), off in zip(items, offs):
            for e in (off, off + p.numel() - 1):                  # first and last element of the parameter fall into a segment of its scale
                seg = next(k for k, end in enumerate(ends) if e < end)
                assert abs(scales[seg] - scale_of[n]) < 1e-7 * scale_of[n], (n, e)
    assert eng.lr_scale_of("head.weight") == 1.0 and abs(eng.lr_scale_of("patch_embed.proj.weight") - layer_decay ** (depth + 1)) < 1e-12


def test_gemm_half_width_tile_plan():
    """host logic of gemm256.hip's half-width tiles (256 CUs assumed without a device): which launches of the 1B step get them, how the ids
    are laid out (whole tiles, then the column halves of the leftover whole tiles, then the N-edge tiles), and when the plan declines"""
    import ctypes as C
    from internvideo_amd import lib
    L = lib.load()

    def plan(M, N, K=1408, cap=256, b_kc=1, **kw):
        d = lib.GemmDesc()
        d.M, d.N, d.K, d.a_kc, d.b_kc, d.batch = M, N, K, 1, b_kc, 1
        d.lda, d.ldb, d.ldc = K, (K if b_kc else N), N
        for k, v in kw.items():
            setattr(d, k, v)
        out = (C.c_int32 * 4)()
        return (L.ivh_gemm256_half_plan(C.byref(d), cap, out), list(out))
    # B = 128 (53376 rows).  1408 wide: 5 whole column tiles x 209 = 1045 = 4 rounds + 21 -> the 21 are cut in two: 42 + 209 edge tiles = 251 units
    assert plan(53376, 1408) == (1, [5, 1024, 42, 1275])
    assert plan(53376, 4224) == (1, [16, 3328, 32, 3569])                    # 3344 = 13 rounds + 16 -> 32 + 209 = 241 units
    assert plan(53376, 6144)[0] == 0                                          # 19.6 rounds, no edge: 152 leftovers do not fit a round of halves
    assert plan(53376, 1408, b_kc=0) == (1, [5, 1024, 42, 1275])            # dgrad layout
    # B = 32 (13344 rows): one round + 9 whole tiles + 53 edge tiles -> 18 + 53 = 71 units after one round
    assert plan(13344, 1408) == (1, [5, 256, 18, 327])
    assert plan(13344, 6144)[0] == 0
    # a small launch that does not fill the CUs is cut entirely into halves (30 + 6 whole tiles on 256 workgroups -> 66 half tiles)
    assert plan(1336, 1408) == (1, [5, 0, 60, 66])
    # an N edge wider than 128 columns is a whole (masked) tile as before
    assert plan(53376, 1408 + 64)[1][0] in (0, 6)
    # more half tiles than workgroups (the frozen CLIP teacher's 263168-row GEMMs: 1028 edge tiles): declined, one half tile per workgroup at most
    assert plan(263168, 9600, 3200)[0] == 0 and plan(263168, 3200, 3200)[0] == 0
    # flavours without a HALF kernel: rows-contiguous A (wgrad), fp32 output, batched
    assert plan(53376, 1408, a_kc=0)[0] == 0 and plan(53376, 1408, c_fp32=1)[0] == 0 and plan(53376, 1408, batch=2)[0] == 0
    L.ivh_gemm256_debug_half(0)
    try:
        assert plan(53376, 1408)[0] == 0
    finally:
        L.ivh_gemm256_debug_half(1)


_STAGE2_BUILD = r"""
def tiny_stage2():
    from types import SimpleNamespace
    from internvideo_amd import mm_internvideo2 as mm, xbert
    from internvideo_amd.stage2 import InternVideo2_Stage2_visual
    from oracle import internvideo2_oracle as O
    scfg = O.named_config("mm88"); bcfg = O.named_bert_config("bert_tiny")
    torch.manual_seed(0)
    vision = mm.PretrainInternVideo2(img_size=scfg.img_size, embed_dim=scfg.embed_dim, depth=scfg.depth, num_heads=scfg.num_heads,
                                     mlp_ratio=scfg.mlp_ratio, num_frames=scfg.num_frames, drop_path_rate=0.0,
                                     attn_pool_num_heads=scfg.attn_pool_num_heads, clip_embed_dim=scfg.clip_embed_dim,
                                     clip_teacher_embed_dim=scfg.clip_teacher_embed_dim, clip_teacher_final_dim=scfg.clip_teacher_final_dim,
                                     clip_return_layer=scfg.clip_return_layer, sep_image_video_pos_embed=scfg.sep_image_video_pos_embed)
    pc = xbert.BertConfig(vocab_size=bcfg.vocab_size, hidden_size=bcfg.hidden_size, num_hidden_layers=bcfg.num_hidden_layers,
                          num_attention_heads=bcfg.num_attention_heads, intermediate_size=bcfg.intermediate_size,
                          max_position_embeddings=bcfg.max_position_embeddings, fusion_layer=bcfg.fusion_layer, encoder_width=scfg.embed_dim)
    config = dict(model=dict(vision_encoder=dict(clip_embed_dim=scfg.clip_embed_dim, img_size=scfg.img_size, num_frames=scfg.num_frames,
                                                 tubelet_size=1, patch_size=scfg.patch_size, only_mask=True),
                             text_encoder=dict(d_model=bcfg.hidden_size), embed_dim=32, temp=0.07),
                  criterion=dict(loss_weight=dict(uta=0.0, vtc=1.0, vtm=1.0, mlm=1.0)))
    tok = SimpleNamespace(pad_token_id=bcfg.pad_token_id, cls_token_id=bcfg.cls_token_id, mask_token_id=bcfg.mask_token_id)
    return InternVideo2_Stage2_visual(config, tok, True, vision_encoder=vision, text_encoder=xbert.BertForMaskedLM(pc)), scfg
"""


def test_engine_manages_the_stage2_model_text_tower_first_vision_tower_in_backward_order():
    """IVTrainEngine on InternVideo2_Stage2_visual (host side, no GPU): the block stack it hooks is the VISION tower's; everything outside
    that stack / the patch embedding (text + fusion tower, heads, temperature, vision decoders) sits at the front of the flat buffers, is
    flagged accumulate (several autograd nodes may contribute per step) and is zeroed per step; the vision blocks follow in backward
    order and keep write-in-place gradients; tied word embeddings appear once; weight decay skips 1-D / bias / no_weight_decay() names
    (multi_modality/utils/optimizer.py:18-31)."""
    import torch  # noqa: F401
    from internvideo_amd.engine import IVTrainEngine
    ns = {"torch": torch}
    exec(_STAGE2_BUILD, ns)
    model, scfg = ns["tiny_stage2"]()
    eng = IVTrainEngine(model, bucket_bytes=64 * 1024)
    assert eng.tower is model.vision_encoder and eng.tower_prefix == "vision_encoder." and eng.accumulate_outside_tower and eng.group_text_wgrads
    names = [n for n, _ in eng.mat_params]
    first_block = next(i for i, n in enumerate(names) if n.startswith("vision_encoder.blocks."))
    assert all(not n.startswith("vision_encoder.blocks.") and not n.startswith("vision_encoder.patch_embed") for n in names[:first_block])
    assert any(n.startswith("text_encoder.bert.encoder.layer.") for n in names[:first_block]) and "vision_proj.weight" in names[:first_block]
    blk = [int(n.split(".")[2]) for n in names if n.startswith("vision_encoder.blocks.")]
    assert blk == sorted(blk, reverse=True) and names[-1].startswith("vision_encoder.patch_embed")
    assert names.count("text_encoder.bert.embeddings.word_embeddings.weight") == 1 and "text_encoder.cls.predictions.decoder.weight" not in names
    named = dict(model.named_parameters())
    for n, p in named.items():
        inside = n.startswith(("vision_encoder.blocks.", "vision_encoder.patch_embed")) or n in ("vision_encoder.cls_token", "vision_encoder.pos_embed")
        assert p._ivh_accum == (not inside), n
    vec_names = {n for n, _ in eng.vec_params}
    assert "temp" in vec_names and "text_encoder.bert.encoder.layer.0.output.LayerNorm.weight" in vec_names and "itm_head.bias" in vec_names
    assert eng.head_end == eng.mat_off[first_block] and eng.head_end > 0
    # the accumulate region is zeroed per step, the in-place region is not touched
    eng.grad_mat.fill_(1.0); eng.grad_vec.fill_(1.0)
    eng.zero_grad()
    assert float(eng.grad_mat[:eng.head_end].abs().max()) == 0.0 and float(eng.grad_mat[eng.head_end:].min()) == 1.0 and float(eng.grad_vec.abs().max()) == 0.0
    # a gradient delivered by plain autograd (.grad) is folded into the buffers (the temperature's comes that way)
    named["temp"].grad = torch.tensor(2.5)
    eng._fold_autograd_grads()
    assert named["temp"].grad is None and float(named["temp"].main_grad) == 2.5
    # bucket plan: contiguous, the text tower's buckets are released by the LAST vision block (the first to finish its backward)
    assert eng.buckets[0][0] == 0 and eng.buckets[-1][1] == eng.n_mat and all(a[1] == b[0] for a, b in zip(eng.buckets, eng.buckets[1:]))
    depth = len(model.vision_encoder.blocks)
    assert eng.bucket_trigger[depth - 1] >= 1 and eng.buckets[eng.bucket_trigger[depth - 1] - 1][1] >= eng.head_end
    # ANY managed parameter may receive its gradient through plain autograd -- the tower's own `pos_embed` does in image
    # steps of a model without sep_image_video_pos_embed (the frame-averaged table is composed with torch ops) -- and is folded like `temp`
    pe = named["vision_encoder.pos_embed"]
    pe.grad = torch.full_like(pe, 0.5)
    eng._fold_autograd_grads()
    assert pe.grad is None and float(pe.main_grad.min()) == 0.5 and not pe._ivh_accum
    # a bucket that has gone to the wire is closed -- a later contribution to one of its parameters raises instead of
    # being lost on the other ranks; zero_grad() reopens everything for the next step
    from internvideo_amd import functional as Fn
    eng._close_bucket(eng.buckets[0][0])                     # what _launch_reduce does before the collective
    first = eng.mat_params[0][1]
    assert first._ivh_closed and not eng.mat_params[-1][1].__dict__.get("_ivh_closed", False)
    with pytest.raises(RuntimeError, match="already been reduced"):
        Fn._ret_grad(first, torch.zeros_like(first.main_grad))
    first.grad = torch.zeros_like(first)
    with pytest.raises(RuntimeError, match="after its bucket was reduced"):
        eng._fold_autograd_grads()
    first.grad = None
    eng.zero_grad()
    assert not first._ivh_closed and Fn._ret_grad(first, first.main_grad) is None
    # the `layer_decay=` shorthand sees the TOWER's names (prefix stripped): blocks decay, the text tower trains at scale 1
    eng2 = IVTrainEngine(ns["tiny_stage2"]()[0], layer_decay=0.5)
    assert eng2.lr_scale_of("vision_encoder.patch_embed.proj.weight") == 0.5 ** (depth + 1)
    assert eng2.lr_scale_of("vision_encoder.blocks.0.attn.qkv.weight") == 0.5 ** depth
    assert eng2.lr_scale_of(f"vision_encoder.blocks.{depth - 1}.mlp.fc1.weight") == 0.5
    assert eng2.lr_scale_of("text_encoder.bert.encoder.layer.0.output.dense.weight") == 1.0 and eng2.lr_scale_of("temp") == 1.0


def test_scaling_model_arithmetic():
    """bench.scaling_model (the prediction an N > 1 line carries): wire bytes of a bandwidth-optimal all-reduce, time on all
    peer links vs one ring, exposed tail = last bucket + vector region; world 1 = nothing on the wire."""
    import importlib.util
    from types import SimpleNamespace
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    eng = SimpleNamespace(buckets=[(0, 50_000_000), (50_000_000, 80_000_000)], n_vec=1_000_000, reduce_mode="allreduce", reduce_dtype="bf16")
    m = bench.scaling_model(eng, 8, 0.4)
    total = 2 * 80_000_000 + 4 * 1_000_000
    assert m["grad_bytes_per_step"] == total and m["wire_bytes_out_per_gpu"] == int(2 * 7 / 8 * total) and m["matrix_buckets"] == 2
    assert abs(m["comm_ms_all_links"] - 2 * 7 / 8 * total / (7 * 153e9) * 1e3) < 0.01
    assert abs(m["comm_ms_one_ring"] - 7 * m["comm_ms_all_links"]) < 0.05
    tail = 2 * 30_000_000 + 4_000_000
    assert abs(m["exposed_tail_ms_all_links"] - 2 * 7 / 8 * tail / (7 * 153e9) * 1e3) < 0.01
    assert 0.99 < m["predicted_scaling_efficiency_all_links"] < 1.0 and m["status"].startswith("model only")
    m1 = bench.scaling_model(eng, 1, 0.4)
    assert m1["wire_bytes_out_per_gpu"] == 0 and m1["predicted_scaling_efficiency_all_links"] == 1.0
    # the N > 1 default: fp32 communication buffers (4 bytes per matrix element on the wire); zero1: one all-to-all of bf16 shards
    # ((W-1)/W of the buffer) + the all-gather of the bf16 weights after AdamW, which is exposed
    m32 = bench.scaling_model(eng, 8, 0.4, reduce_dtype="fp32")
    assert m32["reduce"] == "allreduce/fp32" and m32["grad_bytes_per_step"] == 4 * 80_000_000 + 4 * 1_000_000
    assert abs(m32["comm_ms_all_links"] - 2 * 7 / 8 * (4 * 80_000_000 + 4_000_000) / (7 * 153e9) * 1e3) < 0.01
    mz = bench.scaling_model(eng, 8, 0.4, reduce_mode="zero1")
    assert mz["wire_bytes_out_per_gpu"] == int(7 / 8 * 160_000_000 + 2 * 7 / 8 * 4_000_000 + 7 / 8 * 160_000_000)
    assert mz["exposed_tail_ms_all_links"] > m["exposed_tail_ms_all_links"]
    assert abs(m["predicted_speedup_all_links"] - 8 * m["predicted_scaling_efficiency_all_links"]) < 0.01


def test_stage2_optimizer_groups_and_schedule_match_the_reference():
    """multi_modality/utils/optimizer.py:17-84 and utils/scheduler.py:26-60 through tests/golden/mm_optim.json (the reference's own functions
    run on a toy module with stage-2-shaped parameter names): weight decay and lr of every parameter, the grouping order, the LambdaLR
    multiplier of every step; and `different_lr_scales` as the engine's per-name factor."""
    import json
    import importlib.util
    from internvideo_amd import schedules as S
    spec = importlib.util.spec_from_file_location("_mk_mm_optim", os.path.join(ROOT, "tests", "golden", "make_golden_mm_optim.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    with open(os.path.join(ROOT, "tests", "golden", "mm_optim.json")) as f:
        g = json.load(f)
    for rec in g["schedules"]:
        got = [S.cosine_warmup_factor(i, **rec["kw"]) for i in range(len(rec["factors"]))]
        assert got == rec["factors"], rec["kw"]                     # same float expression: equal to the last bit
    model = mk.Toy()
    names = {id(p): n for n, p in model.named_parameters()}
    t = S.add_different_lr(S.add_weight_decay(model, 0.05, mk.NO_DECAY, True), mk.DIFF["names"], mk.DIFF["lr"], mk.DIFF["default"])
    assert [[n, wd, lr] for n, _, wd, lr in t] == g["tuples"]
    assert "vision_encoder.frozen.weight" not in [n for n, *_ in t]                  # frozen weights are not optimised
    groups = S.create_optimizer_params_group(t)
    assert [dict(weight_decay=gr["weight_decay"], lr=gr["lr"], params=[names[id(p)] for p in gr["params"]]) for gr in groups] == g["groups"]
    torch.optim.AdamW(groups, lr=mk.DIFF["default"], betas=(0.9, 0.98))               # the groups are what torch's optimizer takes
    assert [[n, wd] for n, _, wd in S.add_weight_decay(model, 0.1, (), False)] == g["tuples_nofilter"]
    scale = S.different_lr_scales(mk.DIFF["names"], mk.DIFF["lr"], mk.DIFF["default"])
    for n, _, lr in g["tuples"]:
        assert abs(scale(n) * mk.DIFF["default"] - lr) < 1e-18, n


def test_sep_pos_embed_builds_in_the_distill_and_finetune_mirrors_on_the_host():
    """internvideo2_distill.py:481-494 / internvideo2.py:390-397: the separable tables replace the joint ones under the reference's names and
    shapes (tests/golden/sep_pos.npz holds the tensors that were loaded into the reference's own modules under exactly these names), the sincos
    initialisation is the reference's (D:551-563, F:454-465) and the composed table is spatial.repeat(T) + temporal.repeat_interleave(H W)."""
    from internvideo_amd import internvideo2 as FT
    from internvideo_amd import internvideo2_distill as D
    from internvideo_amd.pos_embed import get_1d_sincos_pos_embed, get_2d_sincos_pos_embed
    gold = np.load(os.path.join(ROOT, "tests", "golden", "sep_pos.npz"))
    cfg = O.named_config("dist64")
    md = D.DistInternVideo2(img_size=cfg.img_size, embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads, mlp_ratio=cfg.mlp_ratio,
                            num_frames=cfg.num_frames, clip_teacher_embed_dim=cfg.clip_teacher_embed_dim, clip_teacher_final_dim=cfg.clip_teacher_final_dim,
                            clip_return_layer=cfg.clip_return_layer, sep_pos_embed=True)
    cfg2 = O.named_config("tiny88")
    mf = FT.InternVideo2(img_size=cfg2.img_size, embed_dim=cfg2.embed_dim, depth=cfg2.depth, num_heads=cfg2.num_heads, mlp_ratio=cfg2.mlp_ratio,
                         num_frames=cfg2.num_frames, num_classes=10, sep_pos_embed=True)
    for m, tag, c in ((md, "dist", cfg), (mf, "ft", cfg2)):
        sd = m.state_dict()
        keys = [f[len(tag) + 4:] for f in gold.files if f.startswith(tag + ":in:")]
        assert keys and all(k in sd and tuple(sd[k].shape) == gold[f"{tag}:in:{k}"].shape for k in keys), tag
        assert "pos_embed" not in sd and "clip_pos_embed" not in sd
        g = m.patch_embed.grid_size
        assert np.allclose(m.pos_embed_spatial[0].detach().numpy(), get_2d_sincos_pos_embed(c.embed_dim, g[1]), atol=1e-6)
        assert np.allclose(m.pos_embed_temporal[0].detach().numpy(), get_1d_sincos_pos_embed(c.embed_dim, g[0]), atol=1e-6)
        assert float(m.pos_embed_cls.detach().abs().max()) == 0.0
        joint = m._pos_table("")
        assert tuple(joint.shape) == (1, 1 + g[0] * g[1] * g[2], c.embed_dim)
        assert torch.allclose(joint[0, 1 + 1 * g[1] * g[2] + 2], m.pos_embed_spatial[0, 2] + m.pos_embed_temporal[0, 1])


def test_internvideo2_teacher_mirror_has_the_reference_state_dict_and_registry_names():
    """models/internvideo2_teacher.py: the class behind teacher_internvideo2_{1B,stage2_1B,6B} (run_distill.py:27).  Keys and shapes of the tiny
    instance equal those of the reference's own module (recorded by make_golden_distill_protocol.py); the stage-2 checkpoint filter of
    :639-656 keeps the vision tower, drops the student-side heads / tables and resizes the 4-frame positional table."""
    import json
    from internvideo_amd import internvideo2_teacher as T2
    fix = json.load(open(os.path.join(ROOT, "tests", "golden", "distill_protocol.json")))
    m = T2.InternVideo2(drop_path_rate=0.0, clip_return_layer=2, **fix["teacher"])
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == fix["teacher_state_dict"]
    assert m.return_index == [2, 1] and m.T == 4
    for name in ("teacher_internvideo2_1B", "teacher_internvideo2_stage2_1B", "teacher_internvideo2_6B"):
        assert callable(getattr(T2, name))
    # T2:639-656 on a synthetic stage-2 checkpoint of the same tiny geometry but 2 frames
    D, n2 = 96, 1 + 2 * 16
    ck = {"vision_encoder.pos_embed": torch.randn(1, n2, D), "vision_encoder.cls_token": torch.ones(1, 1, D), "vision_encoder.clip_pos_embed": torch.zeros(1, n2, D),
          "vision_encoder.clip_decoder.0.head.weight": torch.zeros(3, 3), "vision_encoder.img_pos_embed": torch.zeros(1, 17, D), "text_encoder.x": torch.zeros(1),
          "vision_encoder.blocks.0.norm1.weight": torch.full((D,), 2.0)}
    src = {k: v.clone() for k, v in ck.items()}

    class Probe:                                                # what interpolate_pos_embed reads off the model
        patch_embed, pos_embed, T = m.patch_embed, m.pos_embed, m.T
    from internvideo_amd.pos_embed import interpolate_pos_embed
    ref_ck = dict(src)
    interpolate_pos_embed(ref_ck, Probe, orig_t_size=2)
    import internvideo_amd.internvideo2_teacher as mod
    orig = mod.interpolate_pos_embed
    mod.interpolate_pos_embed = lambda c, mm, orig_t_size=4: orig(c, mm, orig_t_size=2)      # the synthetic checkpoint has 2 frames, not 4
    try:
        out = T2.stage2_vision_state_dict(ck, m)
    finally:
        mod.interpolate_pos_embed = orig
    assert set(out) == {"pos_embed", "cls_token", "blocks.0.norm1.weight"}
    assert tuple(out["pos_embed"].shape) == tuple(m.pos_embed.shape) and torch.equal(out["pos_embed"], ref_ck["vision_encoder.pos_embed"])


def test_bench_prices_utilisation_on_executed_flops():
    """bench.py (round 6): with DropPath skipping the MFMA fractions are priced on the FLOPs the step executed.  droppath_account turns the kept
    (block, branch, sample) counts summed on the device into executed FLOPs per clip with the arithmetic of SURVEY.md 8(d) (the nominal 1B
    figure is reproduced when nothing is dropped); droppath_straggler is the binomial model of the slowest of N ranks."""
    import importlib.util
    from types import SimpleNamespace
    spec = importlib.util.spec_from_file_location("_bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    depth, D, F, L, B, steps = 40, 1408, 6144, 417, 128, 5
    blk = SimpleNamespace(attn=SimpleNamespace(qkv=SimpleNamespace(weight=torch.empty(3 * D, D))), mlp=SimpleNamespace(fc1=SimpleNamespace(weight=torch.empty(F, D))))
    model = SimpleNamespace(depth=depth, blocks=[blk], drop_path_rates=[0.25 * i / 39 for i in range(depth)],
                            dp_count_acc=torch.zeros((depth, 2, 2), dtype=torch.int64))
    nominal = bench.FLOP_PER_CLIP_FWD_BWD
    off = bench.droppath_account(model, steps, B, L, nominal)
    assert off["enabled"] is False and off["executed_flop_per_clip"] == nominal
    model.dp_count_acc[:, :, 0] = steps * B                                 # everything kept
    model.dp_count_acc[:, :, 1] = steps * B * L
    full = bench.droppath_account(model, steps, B, L, nominal)
    assert full["enabled"] and abs(full["executed_flop_per_clip"] - nominal) < 1e-6 * nominal and full["kept_fraction"] == 1.0
    br = full["branch_flop_per_clip_attn_mlp"]
    assert abs(depth * sum(br) - 3 * 880.9e9) < 2e-3 * 3 * 880.9e9             # 880.9 GFLOP forward in the blocks
    keep = torch.tensor([1.0 - r for r in model.drop_path_rates])
    model.dp_count_acc[:, :, 0] = (keep * steps * B).round().long().view(-1, 1)
    part = bench.droppath_account(model, steps, B, L, nominal)
    assert 0.11 < part["skipped_flop_share"] < 0.125 and abs(part["kept_fraction"] - 0.875) < 2e-3
    st = bench.droppath_straggler(model, B, L, 350.0, part)
    assert 0.8 < st["sigma_ms_per_rank"] < 1.8 and st["efficiency_factor"]["8"] < st["efficiency_factor"]["2"] < 1.0
    assert bench.droppath_straggler(model, B, L, 350.0, off) is None
    t = torch.tensor([3], dtype=torch.int32)
    assert bench._dyn_scale(None) == 1.0 and bench._dyn_scale((t, 4)) == 0.75
    assert abs(bench._dyn_scale([(100.0, (t, 4)), (100.0, None)]) - 0.875) < 1e-9


def test_bench_counter_stamp_lookup(tmp_path):
    """bench.py attaches the PMC summaries under profiles/ to its `roofline` object with a digest of the kernel sources.  The lookup takes an
    exact key, else the instantiation that shares the leading template arguments and has the MOST launches (tail split / device-side row count
    twins share them), and says whether the digest is the current one."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("_bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    path = tmp_path / "pmc.json"
    table = {"k<true, false, 0, false, false>": {"launches": 80, "x": 1}, "k<true, false, 0, true, true>": {"launches": 160, "x": 2},
             "k<true, true, 0, false>": {"launches": 500, "x": 3}, "exact<1,0>": {"launches": 7, "x": 4}}
    path.write_text(json.dumps({"kernels": table, "source_digest": bench._source_digest()}))
    assert bench._stamped(str(path), "exact<1,0>")["x"] == 4
    got = bench._stamped(str(path), "k<true, false, 0>")
    assert got["x"] == 2 and got["launches"] == 160 and got["matches_current_sources"] is True
    assert bench._stamped(str(path), "k<false>") is None and bench._stamped(str(tmp_path / "absent.json"), "k") is None
    path.write_text(json.dumps({"kernels": table, "source_digest": "0123456789abcdef"}))
    assert bench._stamped(str(path), "exact<1,0>")["matches_current_sources"] is False
