"""MI355X parity of the VideoMAE pixel-reconstruction path (internvideo_amd.videomae_pretrain) against
fixtures made by the REFERENCE's own PretrainVisionTransformer + engine labels (tests/golden/flavours.npz), and of its token-edge
kernels against plain torch.  Tolerances: index lists / row copies bit-exact; labels (fp32 arithmetic) 1e-5; predictions rel-L2
<= 1e-2; loss <= 1e-3 relative; gradients <= max(3e-2, 3 x the reference's own bf16-vs-fp32 discrepancy) (16x16 corners: 5e-2)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from internvideo_amd import functional as Fn, ops, videomae_pretrain as V  # noqa: E402
from oracle import internvideo2_oracle as O  # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "flavours.npz")
DEV = "cuda"


def rel(a, b):
    a = torch.as_tensor(a).double().cpu(); b = torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def build(cfg, params):
    m = V.PretrainVisionTransformer(img_size=cfg.img_size, patch_size=cfg.patch_size, encoder_embed_dim=cfg.enc_dim, encoder_depth=cfg.enc_depth,
                                    encoder_num_heads=cfg.enc_heads, decoder_num_classes=cfg.num_classes, decoder_embed_dim=cfg.dec_dim,
                                    decoder_depth=cfg.dec_depth, decoder_num_heads=cfg.dec_heads, mlp_ratio=cfg.mlp_ratio, qkv_bias=cfg.qkv_bias,
                                    init_values=cfg.init_values, tubelet_size=cfg.tubelet_size, num_frames=cfg.num_frames,
                                    norm_layer=lambda d: torch.nn.LayerNorm(d, eps=cfg.ln_eps))
    m.load_state_dict(params, strict=True)
    return m.to(DEV).train()


def check_grads(g, pre, model):
    sd = dict(model.named_parameters())
    worst = {}
    for key in g.files:
        if key.startswith(pre + "grad:"):
            k = key[len(pre) + 5:]
            worst[k] = rel(sd[k].grad, g[key])
        elif key.startswith(pre + "gradnorm:"):
            k = key[len(pre) + 9:]
            gr = sd[k].grad
            g2 = gr.reshape(gr.shape[0], -1)
            worst["corner:" + k] = rel(g2[:16, :16], g[pre + "gradcorner:" + k])
            worst["norm:" + k] = abs(gr.double().norm().item() - g[key][0]) / g[key][0]

    def tol(k):
        floor = 5e-2 if k.startswith("corner:") else 3e-2
        e = pre + "bf16err:" + k
        return max(floor, 3.0 * float(g[e][0])) if e in g.files else floor
    bad = {k: (v, tol(k)) for k, v in worst.items() if v > tol(k)}
    assert not bad, bad
    return len(worst)


@pytest.mark.parametrize("name,seed,B,n_mask", [("mae_tiny", 8, 2, 20), ("mae_tiny88", 9, 2, 12)])
def test_videomae_matches_reference_golden(name, seed, B, n_mask):
    g = np.load(GOLD)
    cfg = O.named_mae_config(name)
    pre = name + ":"
    params = O.synthetic_mae_params(cfg, seed=seed)
    video, mask = O.synthetic_mae_batch(cfg, B, n_mask, seed=seed)
    mm = torch.from_numpy(mask)
    model = build(cfg, params)
    # labels: engine_for_pretraining.py:66-98 (normalised and raw flavours)
    labels = model.pixel_target(video.to(DEV), mm)
    assert labels.dtype == torch.float32 and rel(labels, g[pre + "labels"]) < 1e-5
    assert rel(model.pixel_target(video.to(DEV), mm, normlize_target=False), g[pre + "labels_raw"]) < 1e-6
    # predictions, loss, gradients (drop-in forward + torch MSELoss, as the reference engine does)
    out = model(video.to(DEV), mm)
    assert out.dtype == torch.bfloat16 and tuple(out.shape) == tuple(g[pre + "out"].shape)
    assert rel(out.float(), g[pre + "out"]) < 1e-2
    loss = torch.nn.MSELoss()(input=out.float(), target=labels)
    ref = g[pre + "loss"][0]
    assert abs(loss.item() - ref) / ref < 1e-3, (loss.item(), ref)
    loss.backward()
    assert check_grads(g, pre, model) >= 20
    # fused step forward (labels + predictions + MSE kernels) gives the same loss and gradients
    model.zero_grad(set_to_none=True)
    l2 = model.forward_loss(video.to(DEV), mm)
    assert abs(l2.item() - ref) / ref < 1e-3, (l2.item(), ref)
    l2.backward()
    assert check_grads(g, pre, model) >= 20
    # a device mask takes the HIP compaction path and yields the same index lists
    v1, k1 = V.mae_gather_indices(mm, DEV)
    v2, k2 = V.mae_gather_indices(mm.to(DEV), DEV)
    assert torch.equal(v1, v2) and torch.equal(k1, k2)
    with pytest.raises(RuntimeError):
        bad = mm.clone(); bad[0, torch.nonzero(~bad[0])[0]] = True                # ragged: one more masked token in clip 0
        model(video.to(DEV), bad)


@pytest.mark.parametrize("mask_type,ratio", [("t_progressive", 0.5), ("t_center_prog", 0.5), ("t_consist", 0.75)])
def test_videomae_under_the_recipes_other_mask_types(mask_type, ratio):
    """`--mask_type t_progressive / t_center_prog / t_consist` (run_mae_pretraining.py:50-55): the frames of a clip show DIFFERENT numbers of
    patches (one of them none at all with these small grids), every clip the same total -- the encoder gather, the mask-token scatter and the
    label gather against the fp32 oracle on such masks.  Masks from internvideo_amd.videomae_masking (bit-exact mirrors, test_videomae_masks.py)."""
    from internvideo_amd import videomae_masking as VM
    cfg = O.MaeConfig(img_size=48, patch_size=8, tubelet_size=2, num_frames=8, enc_dim=64, enc_depth=2, enc_heads=2, dec_dim=32, dec_depth=1,
                      dec_heads=2, mlp_ratio=4.0, qkv_bias=True, init_values=0.1)
    gen = VM.build_mask_generator(mask_type, (cfg.num_frames // cfg.tubelet_size, 6, 6), ratio)
    np.random.seed(5)
    mask = np.stack([gen() for _ in range(3)]).astype(bool)
    per_frame = (~mask).reshape(3, 4, 36).sum(2)
    if mask_type != "t_consist":
        assert len(set(per_frame[0].tolist())) > 1                  # the point of the test: ragged over frames, equal over clips
    assert (per_frame.sum(1) == per_frame[0].sum()).all()
    params = O.synthetic_mae_params(cfg, seed=4)
    video, _ = O.synthetic_mae_batch(cfg, 3, 8, seed=4)
    model = build(cfg, params)
    mm = torch.from_numpy(mask)
    want_labels = O.videomae_pixel_target(video, mask, cfg.patch_size, cfg.tubelet_size)
    assert rel(model.pixel_target(video.to(DEV), mm), want_labels) < 1e-5
    want = O.videomae_forward({k: v.float() for k, v in params.items()}, video, mask, cfg)
    out = model(video.to(DEV), mm)
    assert tuple(out.shape) == tuple(want.shape) and rel(out.float(), want) < 1e-2
    want_loss = torch.nn.functional.mse_loss(want, want_labels).item()
    got_loss = model.forward_loss(video.to(DEV), mm.to(DEV)).item()
    assert abs(got_loss - want_loss) / want_loss < 1e-3, (got_loss, want_loss)


def test_videomae_trains_with_a_torch_optimizer_and_drop_path():
    cfg = O.named_mae_config("mae_tiny")
    params = O.synthetic_mae_params(cfg, seed=8)
    video, mask = O.synthetic_mae_batch(cfg, 4, 20, seed=3)
    model = build(cfg, params)
    for blk, r in zip(model.encoder.blocks, (0.0, 0.2)):
        blk.drop_path = r
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3, weight_decay=0.05)
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        loss = model.forward_loss(video.to(DEV), torch.from_numpy(mask))
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_mae_token_edge_kernels_vs_torch():
    gen = torch.Generator().manual_seed(0)
    B, N, D = 3, 24, 64
    mask = torch.zeros(B, N, dtype=torch.bool)
    for b in range(B):
        mask[b, torch.randperm(N, generator=gen)[:15]] = True
    vis, msk = V.mae_gather_indices(mask, DEV)
    Nvis, Nmask = 9, 15
    pos = torch.randn(N, D, generator=gen)
    tok = torch.randn(B * Nvis, D, generator=gen).bfloat16()
    # assemble (no cls)
    want = tok.float().view(B, Nvis, D) + pos.expand(B, -1, -1)[~mask].view(B, Nvis, D)
    got = ops.assemble_tokens_nocls(tok.to(DEV), pos.to(DEV), vis)
    assert torch.equal(got.cpu().view(B, Nvis, D), want)
    # decoder input = cat([x_vis + pos[~mask], mask_token + pos[mask]])
    mt = torch.randn(D, generator=gen)
    wfull = torch.cat([want, mt + pos.expand(B, -1, -1)[mask].view(B, Nmask, D)], 1)
    gfull = ops.mae_decoder_input(tok.to(DEV), mt.to(DEV), pos.to(DEV), vis, msk)
    assert torch.equal(gfull.cpu().view(B, N, D), wfull)
    # row windows and their backward
    x = torch.randn(B * N, D, generator=gen)
    w = ops.rows_window(x.to(DEV), B, N, Nvis, Nmask)
    assert torch.equal(w.cpu().view(B, Nmask, D), x.view(B, N, D)[:, Nvis:].bfloat16())
    back = ops.rows_window_bwd(w, B, N, Nvis, Nmask).cpu().view(B, N, D)
    assert torch.equal(back[:, Nvis:], w.float().cpu().view(B, Nmask, D)) and not back[:, :Nvis].any()
    head = ops.rows_window_bwd(torch.ones(B * Nvis, D, device=DEV), B, N, 0, Nvis).cpu().view(B, N, D)
    assert head[:, :Nvis].eq(1).all() and not head[:, Nvis:].any()
    # autograd seams: decoder input and MSE
    xv = tok.to(DEV).requires_grad_(True)
    mtp = torch.nn.Parameter(mt.to(DEV).view(1, 1, D))
    y = Fn.MaeDecoderInputFn.apply(xv, mtp, pos.to(DEV), vis, msk)
    cot = torch.randn(B * N, D, generator=gen).to(DEV)
    (y * cot).sum().backward()
    assert rel(xv.grad.float(), cot.view(B, N, D)[:, :Nvis].reshape(-1, D)) < 5e-3
    assert rel(mtp.grad.view(-1), cot.view(B, N, D)[:, Nvis:].bfloat16().float().sum((0, 1))) < 1e-5
    pred = torch.randn(7, 50, generator=gen).bfloat16().to(DEV).requires_grad_(True)
    tgt = torch.randn(7, 50, generator=gen).to(DEV)
    l = Fn.MseLossFn.apply(pred, tgt)
    lr = torch.nn.functional.mse_loss(pred.detach().float(), tgt)
    assert abs(l.item() - lr.item()) / lr.item() < 1e-6
    (l * 3.0).backward()
    assert rel(pred.grad.float(), 3.0 * 2 * (pred.detach().float() - tgt) / pred.numel()) < 5e-3


def test_ln_block_stack_inference_matches_training_forward():
    """functional.ln_block_stack_infer (teacher / evaluation loop) == LNBlockStackFn.forward, bit for bit up to the GELU flavour's
    identical arithmetic (both erf): same kernels, no saved activations."""
    cfg = O.named_mae_config("mae_tiny")
    model = build(cfg, O.synthetic_mae_params(cfg, seed=8)).eval()
    gen = torch.Generator().manual_seed(1)
    B, L, D = 2, 12, cfg.enc_dim
    x0 = torch.randn(B * L, D, generator=gen).to(DEV)
    blocks = model.encoder.blocks
    with torch.no_grad():
        a = V._run_blocks(blocks, x0, B, L, cfg.enc_heads, cfg.ln_eps, False)
        outs = Fn.ln_block_stack_infer(x0, [blk.flat_params() for blk in blocks], B, L, cfg.enc_heads, cfg.ln_eps, taps=(0,))
    assert rel(outs[len(blocks) - 1], a) < 2e-3 and 0 in outs


def test_videomae_teacher_matches_reference_golden():
    """the frozen VideoMAE teacher (internvideo_amd.videomae_teacher): attention layout as coded in the reference (heads as the
    sequence axis) through the strided attention kernel, resized positional table, final norm on the last tap, l2 -- vs the reference
    module (flash_attn_func stand-in following flash_attn's documented contract)."""
    from internvideo_amd import videomae_teacher as T
    g = np.load(GOLD)
    cfg = O.named_mae_config("mae_teach")
    p = O.mae_teacher_params(cfg, seed=10)
    p["pos_embed"] = torch.from_numpy(g["mteach:pos_embed"])
    video, mask = O.synthetic_mae_batch(cfg, 2, 96, seed=10)
    m = T.VisionTransformer(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.enc_dim, depth=cfg.enc_depth, num_heads=cfg.enc_heads,
                            mlp_ratio=cfg.mlp_ratio, qkv_bias=True, all_frames=cfg.num_frames, tubelet_size=cfg.tubelet_size, mae_return_layer=2,
                            norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6))
    m.load_state_dict(p, strict=True)
    m = m.to(DEV).eval()
    zf = m(video.to(DEV))
    zm = m(video.to(DEV), torch.from_numpy(mask))
    assert zf.dtype == torch.bfloat16 and tuple(zf.shape) == tuple(g["mteach:z_full"].shape) and tuple(zm.shape) == tuple(g["mteach:z_masked"].shape)
    tol = max(1e-2, 2.0 * float(g["mteach:bf16err:z_full"][0]))
    assert rel(zf.float(), g["mteach:z_full"]) < tol and rel(zm.float(), g["mteach:z_masked"]) < tol
    # token-to-token attention (the semantics the checkpoint was trained with) vs the oracle's standard flavour
    ms = T.VisionTransformer(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.enc_dim, depth=cfg.enc_depth, num_heads=cfg.enc_heads,
                             mlp_ratio=cfg.mlp_ratio, qkv_bias=True, all_frames=cfg.num_frames, tubelet_size=cfg.tubelet_size, mae_return_layer=2,
                             norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6), attn_semantics="standard")
    ms.load_state_dict(p, strict=True)
    with torch.no_grad():
        want = O.videomae_teacher_forward(p, video, None, cfg.enc_heads, cfg.enc_depth, [2, 1], cfg.tubelet_size, cfg.patch_size, as_coded=False)
    assert rel(ms.to(DEV).eval()(video.to(DEV)).float(), want) < tol


def test_videomae_teacher_at_real_geometry_matches_the_reference_digest():
    """The frozen VideoMAE teacher at VideoMAE-g's width and sequence geometry (1408 wide, 16 heads of 88, 16 frames of 224^2 -> 2048 tokens on
    the 8 x 16 x 16 sinusoid table; depth 2) against a digest of the REFERENCE's own module at that size
    (tests/golden/mae_teacher_fullwidth_digest.npz, make_golden_mae_teacher_fullwidth.py; attention as coded in videomae.py:91-96, i.e. through
    the strided attention entry point with 2048 "heads" of sequence length 16): 1e-2 on the l2-normalised targets."""
    from internvideo_amd import videomae_teacher as T
    from tests.test_flavours_oracle import _mae_teacher_fullwidth
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mae_teacher_fullwidth_digest.npz"))
    cfg, p, video = _mae_teacher_fullwidth()
    m = T.VisionTransformer(img_size=cfg.img_size, patch_size=cfg.patch_size, embed_dim=cfg.enc_dim, depth=cfg.enc_depth, num_heads=cfg.enc_heads,
                            mlp_ratio=cfg.mlp_ratio, qkv_bias=True, all_frames=cfg.num_frames, tubelet_size=cfg.tubelet_size, mae_return_layer=2,
                            norm_layer=lambda d: torch.nn.LayerNorm(d, eps=1e-6))
    m.load_state_dict(p, strict=True)
    m = m.to(DEV).eval()
    z = m(video.to(DEV))
    assert tuple(z.shape) == tuple(int(i) for i in g["z:shape"])
    rows = z.detach().float().cpu().double().numpy().reshape(-1, z.shape[-1])
    C = rows.shape[1]
    proj = np.random.Generator(np.random.PCG64(777 + C)).standard_normal((C, 16)).astype(np.float32) / np.sqrt(C).astype(np.float32)
    e_rows = np.linalg.norm(rows[:3] - g["z:rows"]) / np.linalg.norm(g["z:rows"])
    e_proj = np.linalg.norm(rows @ proj.astype(np.float64) - g["z:proj"]) / np.linalg.norm(g["z:proj"])
    assert e_rows < 1e-2 and e_proj < 1e-2, (e_rows, e_proj)


def test_videomae_pixel_path_at_base_geometry_matches_the_reference_digest():
    """The VideoMAE pixel path at pretrain_mae_base_patch16_224's geometry (ViT-B/16 encoder, 4 x 384 decoder, 16 frames of 224^2 -> 1568
    tokens, 157 visible, 1411 reconstructed) against a digest of the REFERENCE's own module + the engine's labels and MSE at that size
    (tests/golden/videomae_base_digest.npz, make_golden_videomae_base.py): predictions 1e-2, loss 1e-3, norms of sampled gradients 3e-2,
    their 16 x 16 corners 5e-2 (the bars of the fixture-sized test)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "videomae_base_digest.npz"))
    cfg = O.named_mae_config("mae_base")
    B, n_mask, seed = (int(x) for x in g["meta"])
    params = O.synthetic_mae_params(cfg, seed=seed)
    video, mask = O.synthetic_mae_batch(cfg, B, n_mask, seed=seed)
    mm = torch.from_numpy(mask)
    model = build(cfg, params)
    labels = model.pixel_target(video.to(DEV), mm)
    out = model(video.to(DEV), mm)
    assert tuple(out.shape) == tuple(int(i) for i in g["out:shape"])
    rows = out.detach().float().cpu().double().numpy().reshape(-1, out.shape[-1])
    C = rows.shape[1]
    proj = np.random.Generator(np.random.PCG64(777 + C)).standard_normal((C, 16)).astype(np.float32) / np.sqrt(C).astype(np.float32)
    e_rows = np.linalg.norm(rows[:3] - g["out:rows"]) / np.linalg.norm(g["out:rows"])
    e_proj = np.linalg.norm(rows @ proj.astype(np.float64) - g["out:proj"]) / np.linalg.norm(g["out:proj"])
    assert e_rows < 1e-2 and e_proj < 1e-2, (e_rows, e_proj)
    loss = torch.nn.MSELoss()(input=out.float(), target=labels)
    ref = float(g["loss"][0])
    assert abs(loss.item() - ref) / ref < 1e-3, (loss.item(), ref)
    loss.backward()
    sd = dict(model.named_parameters())
    worst = {}
    for key in g.files:
        if key.startswith("grad:") and key.endswith(":norm"):
            k = key[5:-5]
            gr = sd[k].grad
            g2 = gr.reshape(gr.shape[0], -1)
            worst["norm:" + k] = abs(gr.double().norm().item() - float(g[key][0])) / float(g[key][0])
            worst["corner:" + k] = rel(g2[:16, :16], g["grad:" + k + ":corner"])
    bad = {k: v for k, v in worst.items() if v > (5e-2 if k.startswith("corner:") else 3e-2)}
    assert len(worst) >= 12 and not bad, bad
