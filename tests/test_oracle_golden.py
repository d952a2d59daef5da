"""Pins the CPU oracle (oracle/internvideo2_oracle.py) against outputs of the REFERENCE's own code
(tests/golden/*.npz, produced by tests/golden/make_golden.py from /root/reference)."""
import os

import numpy as np
import pytest
import torch

from oracle import internvideo2_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _rel(a, b):
    a = torch.as_tensor(a, dtype=torch.float64); b = torch.as_tensor(b, dtype=torch.float64)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("name", ["tiny64", "tiny88"])
def test_student_forward_matches_reference(name):
    g = np.load(os.path.join(GOLD, f"student_{name}.npz"))
    B, n_vis, seed = (int(v) for v in g["meta"])
    cfg = O.named_config(name)
    p = O.synthetic_params(cfg, seed=seed)
    video, mask, targets = O.synthetic_batch(cfg, B, n_vis, seed=seed)
    # gather indices: integer work, bit-exact
    assert np.array_equal(O.visible_indices(mask), g["vis_idx"])
    (oc, of, om), blocks = O.student_forward(p, video, mask, cfg, return_blocks=True)
    # fp32 oracle vs fp32 reference: only summation-order noise is allowed
    assert _rel(torch.stack(blocks), g["blocks"]) < 2e-6
    assert _rel(oc, g["x_clip_align"]) < 5e-6
    assert _rel(of, g["x_align"]) < 5e-6
    assert _rel(om, g["x_mae_align"]) < 5e-6
    total, parts = O.distill_losses((oc, of, om), targets)
    ref = g["losses"]
    assert abs(total.item() - ref[0]) / abs(ref[0]) < 2e-6
    for a, b in zip(parts, ref[1:]):
        assert abs(a.item() - b) / abs(b) < 2e-6


@pytest.mark.parametrize("name", ["tiny64", "tiny88"])
def test_student_backward_matches_reference(name):
    """autograd through the oracle reproduces the reference's parameter gradients."""
    g = np.load(os.path.join(GOLD, f"student_{name}.npz"))
    B, n_vis, seed = (int(v) for v in g["meta"])
    cfg = O.named_config(name)
    p = {k: v.clone().requires_grad_(True) for k, v in O.synthetic_params(cfg, seed=seed).items()}
    video, mask, targets = O.synthetic_batch(cfg, B, n_vis, seed=seed)
    out = O.student_forward(p, video, mask, cfg)
    total, _ = O.distill_losses(out, targets)
    total.backward()
    n = 0
    for key in g.files:
        if key.startswith("grad:"):
            k = key[5:]
            assert _rel(p[k].grad, g[key]) < 2e-5, k
            n += 1
        elif key.startswith("gradnorm:"):
            k = key[9:]
            gr = p[k].grad
            assert abs(gr.double().norm().item() - g[key][0]) / g[key][0] < 2e-5, k
            g2 = gr.reshape(gr.shape[0], -1) if gr.ndim == 5 else (gr.reshape(-1, gr.shape[-1]) if gr.ndim != 2 else gr)
            assert _rel(g2[:16, :16], g["gradcorner:" + k]) < 5e-5, k
            n += 1
    assert n >= 25


def test_sincos_tables_match_reference():
    g = np.load(os.path.join(GOLD, "tables.npz"))
    for (D, gs, t) in [(128, 4, 4), (176, 4, 4), (384, 8, 4)]:
        tab = O.sincos_pos_embed_3d(D, gs, t, cls_token=True)
        assert np.array_equal(tab, g[f"sincos3d_{D}_{gs}_{t}"])     # same float64 numpy arithmetic: bit-exact
    tab = O.sincos_pos_embed_3d(1408, 16, 8, cls_token=True)[::97, ::13]
    assert np.array_equal(tab, g["sincos3d_1408_16_8"])


def test_mask_generators_match_reference():
    g = np.load(os.path.join(GOLD, "tables.npz"))
    for seed in (0, 7):
        assert np.array_equal(O.tube_mask((4, 8, 8), 0.75, np.random.RandomState(seed)), g[f"tube_{seed}"])
        assert np.array_equal(O.tube_mask((8, 16, 16), 0.8, np.random.RandomState(seed)), g[f"tube16_{seed}"])
        assert np.array_equal(O.random_mask((4, 16, 16), 0.8, np.random.RandomState(seed)), g[f"random_{seed}"])


def test_vtc_loss_matches_reference():
    g = np.load(os.path.join(GOLD, "tables.npz"))
    rng = np.random.Generator(np.random.PCG64(5))
    v = torch.from_numpy(rng.standard_normal((24, 512)).astype(np.float32)).requires_grad_(True)
    t = torch.from_numpy(rng.standard_normal((24, 512)).astype(np.float32)).requires_grad_(True)
    idx = torch.from_numpy(g["vtc_idx"])
    s, _ = O.contrastive_sim(v, t, torch.tensor(0.07))
    assert _rel(s, g["vtc_sim_v2t"]) < 1e-6
    loss = O.vtc_loss(v, t, idx, torch.tensor(0.07))
    assert abs(loss.item() - g["vtc_loss"][0]) / g["vtc_loss"][0] < 1e-6
    loss.backward()
    assert _rel(v.grad, g["vtc_grad_v"]) < 1e-5 and _rel(t.grad, g["vtc_grad_t"]) < 1e-5
    l2 = O.vtc_loss(v.detach(), t.detach(), None, torch.tensor(0.07))
    assert abs(l2.item() - g["vtc_loss"][1]) / g["vtc_loss"][1] < 1e-6


def test_attention_mask_from_importance_and_ragged_rejected():
    imp = np.stack([np.random.RandomState(i).permutation(16) for i in range(6)])     # B=2, T=3
    m = O.attention_mask_from_importance(imp, B=2, mask_ratio=0.8)
    assert m.shape == (2, 49) and not m[:, 0].any()
    assert ((~m).sum(1) == 1 + 3 * (16 - int(16 * 0.8))).all()
    idx = O.visible_indices(m)
    assert (np.diff(idx, axis=1) > 0).all()
    bad = m.copy(); bad[0, 1] = not bad[0, 1]
    with pytest.raises(ValueError):
        O.visible_indices(bad)


@pytest.mark.parametrize("name", ["1B", "B14", "S14", "6Bshape"])
def test_oracle_matches_the_reference_at_the_full_1B_size(name):
    """The oracle pinned at the sizes the benchmark runs at: tests/golden/student_{1B,B14}_digest.npz hold digests of the REFERENCE's own fp32
    CPU forward + backward of pretrain_internvideo2_1B_patch14_224 (40 x 1408, 8 x 224^2, L = 417: BASELINE configs[2]), of the B/14 model (configs[1]),
    the S/14 model (configs[0], the reference's CPU-runnable case) and configs[4]'s width at depth 2 (make_golden_fullsize.py) on the synthetic parameters / batch the parity tests of those
    sizes use.  Outputs (first rows in full, 16 random projections of every token row), the four losses and sampled parameter gradients of
    the oracle's run of the same inputs: 2e-5 / 1e-6 / 2e-4 relative (fp32 summation order only)."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"student_{name}_digest.npz")
    g = np.load(path)
    torch.set_num_threads(min(os.cpu_count() or 8, 32))
    if name == "6Bshape":     # configs[4]'s width / heads (3200, 25 x 128, pool heads 200 wide) at depth 2: the geometry of test_6B_shaped_student_matches_oracle
        cfg = O.StudentConfig(img_size=56, embed_dim=3200, depth=2, num_heads=25, mlp_ratio=4.0, num_frames=4, attn_pool_num_heads=16,
                              clip_embed_dim=768, clip_teacher_embed_dim=3200, clip_teacher_final_dim=768, clip_return_layer=1,
                              mae_teacher_embed_dim=1408, mae_return_layer=1)
    else:
        cfg = O.named_config(name)
    B, n_vis, seed = (int(x) for x in g["meta"])
    params = O.synthetic_params(cfg, seed=seed)
    video, mask, targets = O.synthetic_batch(cfg, B, n_vis, seed=seed)
    want_grad = [k[5:-7] for k in g.files if k.startswith("grad:") and k.endswith(":corner")] + \
                [k[5:] for k in g.files if k.startswith("grad:") and not k.endswith(":corner") and not k.endswith(":norm")]
    p = {k: (v.clone().requires_grad_(True) if k in want_grad else v) for k, v in params.items()}
    out = O.student_forward(p, video, mask, cfg)
    total, parts = O.distill_losses(out, targets)
    total.backward()
    got_losses = np.array([total.item()] + [float(x.detach()) for x in parts], dtype=np.float64)
    assert np.allclose(got_losses, g["losses"], rtol=1e-6, atol=0), (got_losses, g["losses"])

    def rel(a, b):
        a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

    for name, t in zip(("x_clip_align", "x_align", "x_mae_align"), out):
        a = t.detach().double().numpy()
        C = a.shape[-1]
        rows = a.reshape(-1, C)
        proj = np.random.Generator(np.random.PCG64(777 + C)).standard_normal((C, 16)).astype(np.float32) / np.sqrt(C).astype(np.float32)
        assert rows.shape[0] == g[name + ":proj"].shape[0]
        assert rel(rows[:3], g[name + ":rows"]) < 2e-5, (name, rel(rows[:3], g[name + ":rows"]))
        assert rel(rows @ proj.astype(np.float64), g[name + ":proj"]) < 2e-5, (name, rel(rows @ proj.astype(np.float64), g[name + ":proj"]))
    for k in want_grad:
        gr = p[k].grad.detach()
        if ("grad:" + k + ":corner") in g.files:
            g2 = gr.reshape(gr.shape[0], -1)
            assert rel(g2[:16, :16].numpy(), g["grad:" + k + ":corner"]) < 2e-4, k
        else:
            assert rel(gr.reshape(-1)[:64].numpy(), g["grad:" + k]) < 2e-4, k
        assert abs(gr.double().norm().item() - float(g["grad:" + k + ":norm"][0])) < 2e-4 * float(g["grad:" + k + ":norm"][0]), k
