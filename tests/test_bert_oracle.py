"""Pins the CPU oracle of the stage-2 text / fusion tower (oracle.internvideo2_oracle: bert_*, mlm_*, vtm_*) to outputs of the
reference's own BertForMaskedLM / MLMLoss / VTC_VTM_Loss.vtm_loss (tests/golden/bert_tiny.npz, made by tests/golden/make_golden_bert.py).
Floating point: fp32 CPU against fp32 CPU, tolerance 2e-5 relative (different association order of the same sums); token / label /
index work: bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import internvideo2_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden", "bert_tiny.npz")
TOL = 2e-5


def rel(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(G))


@pytest.fixture(scope="module")
def tower():
    cfg = O.named_bert_config("bert_tiny")
    return cfg, O.synthetic_bert_params(cfg, seed=0)


def test_forward_modes(gold, tower):
    cfg, p = tower
    ids, mask = torch.from_numpy(gold["in:ids"]), torch.from_numpy(gold["in:mask"])
    vision = torch.from_numpy(gold["in:vision"])
    text = O.bert_model(p, cfg, input_ids=ids, attention_mask=mask, mode="text")
    assert rel(text.numpy(), gold["text"]) < TOL
    fused = O.bert_model(p, cfg, encoder_embeds=text, attention_mask=mask, encoder_hidden_states=vision, mode="fusion")
    assert rel(fused.numpy(), gold["fused"]) < TOL
    multi = O.bert_model(p, cfg, input_ids=ids, attention_mask=mask, encoder_hidden_states=vision,
                         encoder_attention_mask=torch.ones(vision.shape[:2]), mode="multi_modal")
    assert rel(multi.numpy(), gold["multi"]) < TOL
    assert rel(multi.numpy(), fused.numpy()) < TOL                      # text then fusion == the full stack
    assert rel(O.bert_mlm_head(fused, p, cfg).numpy(), gold["mlm_logits"]) < TOL


def test_mlm_masking_is_bit_exact(gold, tower):
    cfg, _ = tower
    ids, labels = O.mlm_mask_tokens(gold["in:ids"], gold["in:draw_mask"], gold["in:draw_replace"], gold["in:draw_random"],
                                    gold["in:random_words"], cfg)
    assert np.array_equal(ids, gold["mlm_ids"]) and np.array_equal(labels, gold["mlm_labels"])
    kept = labels != -100
    assert kept.any() and not kept[gold["in:ids"] == cfg.pad_token_id].any() and not kept[:, 0].any()
    assert (ids[kept] == cfg.mask_token_id).any()


def test_mlm_loss_and_gradients(gold, tower):
    cfg, p0 = tower
    p = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
    loss = O.mlm_loss(p, cfg, torch.from_numpy(gold["mlm_ids"]), torch.from_numpy(gold["mlm_labels"]), torch.from_numpy(gold["in:mask"]),
                      torch.from_numpy(gold["in:vision"]))
    assert abs(loss.item() - gold["mlm_loss"][0]) < 1e-5 * abs(gold["mlm_loss"][0])
    loss.backward()
    for k in [k for k in gold if k.startswith("mlm_grad:bert") or k.startswith("mlm_grad:cls")]:
        name = k.split(":", 1)[1]
        assert rel(p[name].grad.numpy(), gold[k]) < 5e-5, name
    gw = p["bert.embeddings.word_embeddings.weight"].grad
    assert abs(gw.double().norm().item() - gold["mlm_gradnorm:word"][0]) < 1e-4 * gold["mlm_gradnorm:word"][0]
    # nn.Embedding(padding_idx) drops the lookup gradient of the pad row; padded positions reach the loss neither as queries (labels
    # -100) nor as keys (masked), so that gradient is exactly zero anyway and the rows agree including row 0 (tied decoder part only)
    assert rel(gw[:16].numpy(), gold["mlm_grad:word_rows"]) < 5e-5


def test_vtm_weights_negatives_loss_and_gradients(gold, tower):
    cfg, p0 = tower
    vp, tp = torch.from_numpy(gold["in:vision_proj"]), torch.from_numpy(gold["in:text_proj"])
    idx = torch.from_numpy(gold["in:idx"])
    w_v2t, w_t2v = O.vtm_negative_weights(vp, tp, idx, float(gold["in:temp"][0]))
    assert rel(w_v2t.numpy(), gold["vtm_weights_v2t"]) < TOL and rel(w_t2v.numpy(), gold["vtm_weights_t2v"]) < TOL
    same = gold["in:idx"][:, None] == gold["in:idx"][None, :]
    assert (w_v2t.numpy()[same] == 0).all() and (w_t2v.numpy()[same] == 0).all()
    vneg, tneg = w_t2v.argmax(1), w_v2t.argmax(1)
    assert np.array_equal(vneg.numpy(), gold["vtm_vision_neg"]) and np.array_equal(tneg.numpy(), gold["vtm_text_neg"])
    p = {k: v.clone().requires_grad_(True) for k, v in p0.items()}
    vision = torch.from_numpy(gold["in:vision"]).requires_grad_(True)
    text = torch.from_numpy(gold["text"]).requires_grad_(True)
    itm_w = torch.from_numpy(gold["in:itm_w"]).requires_grad_(True)
    itm_b = torch.from_numpy(gold["in:itm_b"]).requires_grad_(True)
    loss = O.vtm_loss_given_negatives(p, cfg, itm_w, itm_b, vision, text, torch.from_numpy(gold["in:mask"]), vneg, tneg)
    assert abs(loss.item() - gold["vtm_loss"][0]) < 1e-5
    loss.backward()
    assert rel(vision.grad.numpy(), gold["vtm_grad_vision"]) < 5e-5
    assert rel(text.grad.numpy(), gold["vtm_grad_text"]) < 5e-5
    assert rel(itm_w.grad.numpy(), gold["vtm_grad_itm_w"]) < 5e-5 and rel(itm_b.grad.numpy(), gold["vtm_grad_itm_b"]) < 5e-5
    for k in [k for k in gold if k.startswith("vtm_grad:")]:
        name = k.split(":", 1)[1]
        assert rel(p[name].grad.numpy(), gold[k]) < 5e-5, name


def test_attention_mask_excludes_padding(tower):
    """the additive -10000 mask (xbert.py:1118-1120) removes padded keys: states of the valid tokens do not depend on what the
    padded positions hold"""
    cfg, p = tower
    ids, mask = O.synthetic_text_batch(cfg, 3, 10, seed=1)
    a = O.bert_model(p, cfg, input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), mode="text")
    ids2 = ids.copy()
    ids2[mask == 0] = cfg.vocab_size - 1
    b = O.bert_model(p, cfg, input_ids=torch.from_numpy(ids2), attention_mask=torch.from_numpy(mask), mode="text")
    keep = torch.from_numpy(mask).bool()
    assert rel(a[keep].numpy(), b[keep].numpy()) < 1e-6


def test_param_shapes_match_reference_bert_large():
    cfg = O.named_bert_config("bert_large_1B")
    s = O.bert_param_shapes(cfg)
    n = sum(int(np.prod(v)) for v in s.values())
    # bert-large (335.1 M with the MLM head and no pooler) + 5 fusion layers' cross-attention (2 x 1024^2 + 2 x 1024 x 1408 + biases + LN)
    assert s["bert.encoder.layer.19.crossattention.self.key.weight"] == (1024, 1408)
    assert "bert.encoder.layer.18.crossattention.self.key.weight" not in s
    assert 355e6 < n < 365e6, n


def test_oracle_matches_the_reference_at_bert_large_config_size():
    """The text / fusion tower pinned at BASELINE configs[3]'s size: tests/golden/bert_large_digest.npz is a digest of the REFERENCE's own
    `BertForMaskedLM` (BERT-large, fusion_layer 19, 1408-wide cross-attention; make_golden_bert_large.py) on the inputs of the config-size GPU
    test -- text-mode and fusion-mode states (first rows + 16 random projections of every row), the MLM loss under recorded draws, corners /
    norms of sampled gradients.  The oracle's run of the same inputs: 2e-5 (states), 1e-6 (loss), 2e-4 (gradients)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bert_large_digest.npz")
    g = np.load(path)
    torch.set_num_threads(min(os.cpu_count() or 8, 32))
    cfg = O.named_bert_config("bert_large_1B")
    B, L, LV = (int(x) for x in g["meta"])
    p = O.synthetic_bert_params(cfg, seed=0, std=0.02)
    ids, mask = O.synthetic_text_batch(cfg, B, L, seed=1)
    gen = torch.Generator().manual_seed(2)
    vision = torch.randn(B, LV, cfg.encoder_width, generator=gen)
    rng = np.random.RandomState(3)
    draws = (rng.rand(B, L) < 0.5, rng.rand(B, L) < 0.8, rng.rand(B, L) < 0.5, rng.randint(0, cfg.vocab_size, size=(B, L)).astype(np.int64))
    m_ids, m_labels = O.mlm_mask_tokens(ids, *draws, cfg)
    keys = [k[5:-7] for k in g.files if k.startswith("grad:") and k.endswith(":corner")]
    pr = {k: (v.clone().requires_grad_(True) if k in keys else v) for k, v in p.items()}
    t_ref = O.bert_model(pr, cfg, input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), mode="text")
    f_ref = O.bert_model(pr, cfg, encoder_embeds=t_ref, attention_mask=torch.from_numpy(mask), encoder_hidden_states=vision, mode="fusion")
    loss = O.mlm_loss(pr, cfg, torch.from_numpy(m_ids), torch.from_numpy(m_labels), torch.from_numpy(mask), vision)
    loss.backward()

    def rel(a, b):
        a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
        return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))

    for name, t in (("text", t_ref), ("fused", f_ref)):
        rows = t.detach().double().numpy().reshape(-1, t.shape[-1])
        C = rows.shape[1]
        proj = np.random.Generator(np.random.PCG64(777 + C)).standard_normal((C, 16)).astype(np.float32) / np.sqrt(C).astype(np.float32)
        assert rel(rows[:3], g[name + ":rows"]) < 2e-5 and rel(rows @ proj.astype(np.float64), g[name + ":proj"]) < 2e-5, name
    assert abs(loss.item() - float(g["mlm_loss"][0])) < 1e-6 * float(g["mlm_loss"][0]), (loss.item(), float(g["mlm_loss"][0]))
    for k in keys:
        gr = pr[k].grad.detach()
        g2 = gr.reshape(gr.shape[0], -1) if gr.dim() > 1 else gr.reshape(1, -1)
        assert rel(g2[:16, :16].numpy(), g["grad:" + k + ":corner"]) < 2e-4, k
        assert abs(gr.double().norm().item() - float(g["grad:" + k + ":norm"][0])) < 2e-4 * float(g["grad:" + k + ":norm"][0]), k
