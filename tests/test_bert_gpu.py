"""Stage-2 text / fusion tower on the GPU against the reference's own outputs (tests/golden/bert_tiny.npz, made by running the reference's
BertForMaskedLM / MLMLoss / vtm_loss) and against the CPU oracle; row kernels against plain fp32 torch.

Tolerances: the kernels compute in bf16 with fp32 accumulation / statistics -- activations within 1.5e-2 relative l2 of the fp32
reference, losses within 5e-3 relative, parameter gradients within 4e-2 relative l2 (the same bound the vision tower's tests use);
integer work (token masking, labels, negative indices) is bit-exact."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import internvideo2_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"
G = os.path.join(os.path.dirname(__file__), "golden", "bert_tiny.npz")


def rel(a, b):
    a = a.detach().double().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(G))


def ln_ref(x, w, b, eps):
    return torch.nn.functional.layer_norm(x, (x.shape[-1],), w, b, eps)


# ---- row kernels -----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,C", [(48, 128), (203, 1024), (37, 1408)])
@pytest.mark.parametrize("flavour", ["sum", "single", "gelu"])
def test_add_layernorm_fwd_bwd(M, C, flavour):
    from internvideo_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + C)
    a = torch.randn(M, C, generator=g).to(DEV).bfloat16()
    r = torch.randn(M, C, generator=g).to(DEV).bfloat16() if flavour == "sum" else None
    w = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV)
    b = (0.1 * torch.randn(C, generator=g)).to(DEV)
    dy = torch.randn(M, C, generator=g).to(DEV).bfloat16()
    dy2 = torch.randn(M, C, generator=g).to(DEV).bfloat16() if flavour == "sum" else None
    gelu = flavour == "gelu"
    y, stats = ops.add_layernorm_fwd(a, r, w, b, 1e-12, gelu=gelu)
    dx, dw, db = ops.add_layernorm_bwd(a, r, w, stats, dy, dy2, gelu=gelu)
    af = a.float().requires_grad_(True)
    wf, bf = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    s = torch.nn.functional.gelu(af) if gelu else af
    if r is not None:
        s = s + r.float()
    yr = ln_ref(s, wf, bf, 1e-12)
    gy = dy.float() + (dy2.float() if dy2 is not None else 0)
    yr.backward(gy)
    assert rel(y, yr) < 4e-3                                   # bf16 rounding of the output only
    assert rel(stats[:, 0], s.mean(1)) < 1e-5 + 1e-4 and rel(stats[:, 1], (s.var(1, unbiased=False) + 1e-12).rsqrt()) < 1e-4
    assert rel(dx, af.grad) < 4e-3
    assert rel(dw, wf.grad) < 1e-4 and rel(db, bf.grad) < 1e-4


@pytest.mark.parametrize("B,L,C,V", [(4, 12, 128, 210), (8, 32, 1024, 3000)])
def test_bert_embed_fwd_bwd(B, L, C, V):
    from internvideo_amd import ops
    g = torch.Generator(device="cpu").manual_seed(B * L)
    word = (0.05 * torch.randn(V, C, generator=g)).to(DEV)
    pos = (0.05 * torch.randn(40, C, generator=g)).to(DEV)
    typ = (0.05 * torch.randn(2, C, generator=g)).to(DEV)
    w = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV)
    b = (0.1 * torch.randn(C, generator=g)).to(DEV)
    ids = torch.randint(1, V, (B, L), generator=g)
    ids[:, -3:] = 0                                            # padding id: its word row gets no gradient
    ids[0, :] = 7                                              # a heavily repeated token: many atomics on one row
    ids = ids.to(DEV)
    dy = torch.randn(B * L, C, generator=g).to(DEV).bfloat16()
    y, stats = ops.bert_embed_fwd(ids, L, word, pos, typ, w, b, 1e-12)
    dword, dpos, dtyp = torch.zeros_like(word), torch.zeros_like(pos), torch.zeros_like(typ)
    dw, db = ops.bert_embed_bwd(ids, L, word, pos, typ, w, stats, dy, 0, dword, dpos, dtyp)
    wr, pr, tr = word.clone().requires_grad_(True), pos.clone().requires_grad_(True), typ.clone().requires_grad_(True)
    wf, bf = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    e = torch.nn.functional.embedding(ids, wr, padding_idx=0) + tr[0] + pr[:L][None]
    yr = ln_ref(e, wf, bf, 1e-12).reshape(B * L, C)
    yr.backward(dy.float())
    assert rel(y, yr) < 4e-3
    assert rel(dword, wr.grad) < 1e-4 and float(dword[0].abs().max()) == 0.0
    assert rel(dpos, pr.grad) < 1e-4 and rel(dtyp, tr.grad) < 1e-4
    assert rel(dw, wf.grad) < 1e-4 and rel(db, bf.grad) < 1e-4


@pytest.mark.parametrize("M,V,dtype", [(48, 210, torch.bfloat16), (33, 30522, torch.bfloat16), (12, 2, torch.bfloat16), (40, 1000, torch.float32)])
def test_ce_rows(M, V, dtype):
    from internvideo_amd import ops
    g = torch.Generator(device="cpu").manual_seed(V)
    ld = (V + 7) // 8 * 8
    buf = torch.full((M, ld), 50.0)                             # padding columns hold junk that must be ignored
    buf[:, :V] = 3 * torch.randn(M, V, generator=g)
    logits = buf.to(DEV).to(dtype)
    labels = torch.randint(0, V, (M,), generator=g)
    labels[::3] = -100
    labels = labels.to(DEV)
    up = torch.tensor([0.37], device=DEV)
    loss, _ = ops.ce_rows(logits, labels, V=V, want_grad=False)
    _, dl = ops.ce_rows(logits, labels, V=V, want_grad=True, dscale=2.0, dscale_dev=up)
    x = logits[:, :V].float().requires_grad_(True)
    lr = torch.nn.functional.cross_entropy(x, labels, ignore_index=-100)
    (lr * 2.0 * 0.37).backward()
    assert abs(loss.item() - lr.item()) < 2e-5 * abs(lr.item()) + 1e-6
    assert rel(dl[:, :V], x.grad) < 5e-3
    assert float(dl[:, V:].float().abs().max()) == 0.0 if ld > V else True
    assert float(dl[::3].float().abs().max()) == 0.0
    if dtype == torch.bfloat16:                                 # in place over the logits: same numbers
        keep = logits.clone()
        _, dl2 = ops.ce_rows(keep, labels, V=V, want_grad=True, dscale=2.0, dscale_dev=up, inplace=True)
        assert dl2.data_ptr() == keep.data_ptr() and torch.equal(dl2, dl)


# ---- the tower against the reference's outputs -----------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def tower():
    from internvideo_amd import xbert
    cfg = O.named_bert_config("bert_tiny")
    pc = xbert.BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                          num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                          max_position_embeddings=cfg.max_position_embeddings, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                          fusion_layer=cfg.fusion_layer, encoder_width=cfg.encoder_width, pad_token_id=cfg.pad_token_id)
    model = xbert.BertForMaskedLM(pc)
    p = O.synthetic_bert_params(cfg, seed=0)
    missing, unexpected = model.load_state_dict(p, strict=False)
    assert not unexpected and set(missing) <= {"bert.embeddings.position_ids", "cls.predictions.decoder.weight", "cls.predictions.decoder.bias"}
    return cfg, p, model.to(DEV).train()


def test_forward_modes_match_the_reference(gold, tower):
    cfg, p, model = tower
    ids, mask = torch.from_numpy(gold["in:ids"]).to(DEV), torch.from_numpy(gold["in:mask"]).to(DEV)
    vision = torch.from_numpy(gold["in:vision"]).to(DEV)
    with torch.no_grad():
        text = model.bert(ids, attention_mask=mask, return_dict=True, mode="text").last_hidden_state
        fused = model.bert(encoder_embeds=text, attention_mask=mask, encoder_hidden_states=vision, encoder_attention_mask=None,
                           return_dict=True, mode="fusion").last_hidden_state
        multi = model.bert(ids, attention_mask=mask, encoder_hidden_states=vision,
                           encoder_attention_mask=torch.ones(vision.shape[:2], dtype=torch.long, device=DEV), return_dict=True,
                           mode="multi_modal").last_hidden_state
    keep = torch.from_numpy(gold["in:mask"]).bool()
    assert rel(text.float().cpu()[keep], gold["text"][keep.numpy()]) < 1.5e-2
    assert rel(fused.float().cpu()[keep], gold["fused"][keep.numpy()]) < 1.5e-2
    assert rel(multi.float().cpu()[keep], gold["multi"][keep.numpy()]) < 1.5e-2
    # padded positions too: their queries still attend to the valid keys (the reference computes them as well)
    assert rel(text.float().cpu(), gold["text"]) < 1.5e-2


def test_attention_mask_with_holes_is_rejected(tower):
    from internvideo_amd.lib import InternVideoHipError
    cfg, p, model = tower
    ids = torch.randint(8, cfg.vocab_size, (2, 8), device=DEV)
    mask = torch.ones(2, 8, dtype=torch.long, device=DEV)
    mask[1, 3] = 0
    with pytest.raises(InternVideoHipError):
        model.bert(ids, attention_mask=mask, mode="text")
    with pytest.raises(InternVideoHipError):
        model.bert(ids.cpu(), attention_mask=None, mode="text")


def _grads(model):
    return {k: v.grad for k, v in model.named_parameters() if v.grad is not None}


def test_mlm_loss_and_gradients_match_the_reference(gold, tower):
    from internvideo_amd.stage2 import MLMLoss
    cfg, p, model = tower
    tok = SimpleNamespace(pad_token_id=cfg.pad_token_id, cls_token_id=cfg.cls_token_id, mask_token_id=cfg.mask_token_id)
    crit = MLMLoss(0.5, tok)
    ids, mask = torch.from_numpy(gold["in:ids"]).to(DEV), torch.from_numpy(gold["in:mask"]).to(DEV)
    draws = tuple(torch.from_numpy(gold["in:" + k]) for k in ("draw_mask", "draw_replace", "draw_random", "random_words"))
    m_ids, m_labels = crit.mask(ids.clone(), cfg.vocab_size, ids.device, targets=ids.clone(), draws=draws)
    assert np.array_equal(m_ids.cpu().numpy(), gold["mlm_ids"]) and np.array_equal(m_labels.cpu().numpy(), gold["mlm_labels"])
    model.zero_grad()
    text = SimpleNamespace(input_ids=ids, attention_mask=mask)
    vision = torch.from_numpy(gold["in:vision"]).to(DEV)
    loss = crit.mlm_loss(model, text, vision, None, draws=draws)
    assert abs(loss.item() - gold["mlm_loss"][0]) < 5e-3 * gold["mlm_loss"][0]
    (loss * 1.0).backward()
    g = _grads(model)
    worst = {}
    for k in [k for k in gold if k.startswith("mlm_grad:bert") or k.startswith("mlm_grad:cls")]:
        name = k.split(":", 1)[1]
        worst[name] = rel(g[name], gold[k])
    assert max(worst.values()) < 4e-2, worst
    gw = g["bert.embeddings.word_embeddings.weight"]
    assert abs(gw.double().norm().item() - gold["mlm_gradnorm:word"][0]) < 2e-2 * gold["mlm_gradnorm:word"][0]
    assert rel(gw[:16], gold["mlm_grad:word_rows"]) < 4e-2


def test_vtm_loss_and_gradients_match_the_reference(gold, tower):
    from internvideo_amd.stage2 import VTC_VTM_Loss
    cfg, p, model = tower
    crit = VTC_VTM_Loss(True)
    head = torch.nn.Linear(cfg.hidden_size, 2).to(DEV)
    with torch.no_grad():
        head.weight.copy_(torch.from_numpy(gold["in:itm_w"])); head.bias.copy_(torch.from_numpy(gold["in:itm_b"]))
    vp, tp = torch.from_numpy(gold["in:vision_proj"]).to(DEV), torch.from_numpy(gold["in:text_proj"]).to(DEV)
    idx = torch.from_numpy(gold["in:idx"]).to(DEV)
    temp = torch.tensor(float(gold["in:temp"][0]), device=DEV)
    w_v2t, w_t2v, _ = crit.vtm_negative_weights(vp, tp, temp, idx)
    assert rel(w_v2t, gold["vtm_weights_v2t"]) < 1e-4 and rel(w_t2v, gold["vtm_weights_t2v"]) < 1e-4
    vneg, tneg = w_t2v.argmax(1), w_v2t.argmax(1)
    assert np.array_equal(vneg.cpu().numpy(), gold["vtm_vision_neg"]) and np.array_equal(tneg.cpu().numpy(), gold["vtm_text_neg"])
    model.zero_grad()
    vision = torch.from_numpy(gold["in:vision"]).to(DEV).bfloat16().requires_grad_(True)
    text = torch.from_numpy(gold["text"]).to(DEV).bfloat16().requires_grad_(True)
    mask = torch.from_numpy(gold["in:mask"]).to(DEV)
    loss = crit.vtm_loss(model.bert, head, temp, vision, text, vp, tp, mask, idx, neg_indices=(vneg, tneg))
    assert abs(loss.item() - gold["vtm_loss"][0]) < 5e-3 * gold["vtm_loss"][0]
    loss.backward()
    assert rel(vision.grad, gold["vtm_grad_vision"]) < 4e-2
    assert rel(text.grad, gold["vtm_grad_text"]) < 4e-2
    assert rel(head.weight.grad, gold["vtm_grad_itm_w"]) < 4e-2 and rel(head.bias.grad, gold["vtm_grad_itm_b"]) < 4e-2
    g = _grads(model)
    for k in [k for k in gold if k.startswith("vtm_grad:")]:
        assert rel(g[k.split(":", 1)[1]], gold[k]) < 4e-2, k
    # the multinomial path draws valid negatives (never the same example)
    loss2 = crit.vtm_loss(model.bert, head, temp, vision.detach(), text.detach(), vp, tp, mask, idx)
    assert torch.isfinite(loss2)


def test_materialised_logits_and_dropout_modes(gold, tower):
    from internvideo_amd import xbert
    from internvideo_amd.lib import InternVideoHipError
    cfg, p, model = tower
    ids, mask = torch.from_numpy(gold["in:ids"]).to(DEV), torch.from_numpy(gold["in:mask"]).to(DEV)
    with pytest.raises(InternVideoHipError):                    # 210 logits per row: only the fused loss handles a ragged vocabulary
        model(ids, attention_mask=mask, mode="text", return_logits=True)
    pc = xbert.BertConfig(vocab_size=208, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                          max_position_embeddings=40, fusion_layer=2, encoder_width=176)          # BERT's default dropout 0.1 / 0.1
    torch.manual_seed(3)
    m2 = xbert.BertForMaskedLM(pc).to(DEV)
    idc = ids.clamp(max=207)
    ev = m2.eval()(idc, attention_mask=mask, mode="text", return_logits=True)
    assert ev.shape == (ids.shape[0], ids.shape[1], 208) and torch.isfinite(ev.float()).all()
    assert torch.equal(ev, m2(idc, attention_mask=mask, mode="text", return_logits=True))        # eval: no dropout, deterministic
    m2.train()

    def run(seed):
        torch.manual_seed(seed)
        xbert._DROP_CALLS = 0
        return m2(idc, attention_mask=mask, mode="text", return_logits=True)
    a, b, c = run(5), run(5), run(6)
    assert torch.equal(a, b) and not torch.equal(a, c)           # reproducible under torch.manual_seed, different under another seed
    assert 0.02 < rel(a, ev) < 0.6                               # dropout 0.1 perturbs, does not destroy
    labels = idc.clone(); labels[:, ::2] = -100
    torch.manual_seed(5); xbert._DROP_CALLS = 0
    loss = m2(idc, attention_mask=mask, mode="text", labels=labels).loss
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(q.grad.float()).all() for q in m2.parameters() if q.grad is not None)


# ---- dropout inside the kernels: the counter-based mask restated on the host ------------------------------------------------------------
def _hash32(x):
    x = x.astype(np.uint64) & 0xFFFFFFFF
    x ^= x >> 16; x = (x * 0x7feb352d) & 0xFFFFFFFF; x ^= x >> 15; x = (x * 0x846ca68b) & 0xFFFFFFFF; x ^= x >> 16
    return x


def dropout_scale(seed, idx, p):
    """common.h drop_scale: 1 / (1 - p) where hash(seed, idx) >= p * 2^32, else 0"""
    idx = np.asarray(idx, dtype=np.uint64)
    h = _hash32(_hash32((idx & 0xFFFFFFFF) ^ np.uint64(seed)) ^ (idx >> np.uint64(32)) ^ np.uint64(0x9e3779b9))
    thresh = np.uint64(min(int(p * 4294967296.0), 4294967295))
    return np.where(h >= thresh, np.float32(1.0 / (1.0 - p)), np.float32(0.0)).astype(np.float32)


@pytest.mark.parametrize("M,C,p", [(203, 1024, 0.1), (48, 128, 0.5)])
def test_hidden_dropout_in_add_layernorm_and_embedding(M, C, p):
    from internvideo_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M)
    seed = 0xC0FFEE + M
    a = torch.randn(M, C, generator=g).to(DEV).bfloat16(); r = torch.randn(M, C, generator=g).to(DEV).bfloat16()
    w = (1 + 0.1 * torch.randn(C, generator=g)).to(DEV); b = (0.1 * torch.randn(C, generator=g)).to(DEV)
    dy = torch.randn(M, C, generator=g).to(DEV).bfloat16()
    mk = torch.from_numpy(dropout_scale(seed, np.arange(M * C, dtype=np.uint64), p).reshape(M, C)).to(DEV)
    assert abs(float((mk > 0).float().mean()) - (1 - p)) < 4.5 * (p * (1 - p) / (M * C)) ** 0.5        # 4.5 sigma of a fair Bernoulli sample
    y, stats = ops.add_layernorm_fwd(a, r, w, b, 1e-12, drop_p=p, seed=seed)
    (dx_r, dx_a), dw, db = ops.add_layernorm_bwd(a, r, w, stats, dy, drop_p=p, seed=seed)
    af, rf = a.float().requires_grad_(True), r.float().requires_grad_(True)
    wf, bf = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = ln_ref(af * mk + rf, wf, bf, 1e-12)
    yr.backward(dy.float())
    assert rel(y, yr) < 4e-3 and rel(dx_r, rf.grad) < 4e-3 and rel(dx_a, af.grad) < 4e-3
    assert float(dx_a[mk == 0].float().abs().max()) == 0.0
    assert rel(dw, wf.grad) < 1e-4 and rel(db, bf.grad) < 1e-4
    # embedding: dropout on the LayerNorm output
    V = 300
    L = 29 if M == 203 else 12
    B = M // L
    word = (0.05 * torch.randn(V, C, generator=g)).to(DEV); pos = (0.05 * torch.randn(40, C, generator=g)).to(DEV)
    typ = (0.05 * torch.randn(2, C, generator=g)).to(DEV)
    ids = torch.randint(1, V, (B, L), generator=g).to(DEV)
    ye, st = ops.bert_embed_fwd(ids, L, word, pos, typ, w, b, 1e-12, p, seed)
    dword, dpos, dtyp = torch.zeros_like(word), torch.zeros_like(pos), torch.zeros_like(typ)
    dwe, dbe = ops.bert_embed_bwd(ids, L, word, pos, typ, w, st, dy[:B * L].contiguous(), 0, dword, dpos, dtyp, p, seed)
    wr = word.clone().requires_grad_(True); wf2 = w.clone().requires_grad_(True)
    e = torch.nn.functional.embedding(ids, wr) + typ[0] + pos[:L][None]
    mke = mk[:B * L]
    yre = ln_ref(e, wf2, b, 1e-12).reshape(B * L, C) * mke
    yre.backward(dy[:B * L].float())
    assert rel(ye, yre) < 4e-3 and rel(dword, wr.grad) < 1e-4 and rel(dwe, wf2.grad) < 1e-4


@pytest.mark.parametrize("B,Lq,Lk,H,hd,p,cross", [(3, 32, 32, 2, 64, 0.1, False), (2, 32, 206, 4, 64, 0.1, True), (2, 70, 70, 1, 64, 0.5, False)])
def test_attention_probability_dropout_forward_and_backward(B, Lq, Lk, H, hd, p, cross):
    """O = (softmax(S) o M) V with the counter-based mask M restated on the host; gradients of q, k, v against torch autograd with that mask"""
    from internvideo_amd import ops
    g = torch.Generator(device="cpu").manual_seed(Lq + Lk)
    seed = 12345 + Lk
    q = torch.randn(B, Lq, H, hd, generator=g).to(DEV).bfloat16()
    k = torch.randn(B, Lk, H, hd, generator=g).to(DEV).bfloat16()
    v = torch.randn(B, Lk, H, hd, generator=g).to(DEV).bfloat16()
    do = torch.randn(B, Lq, H, hd, generator=g).to(DEV).bfloat16()
    kv_len = torch.tensor([Lk, max(1, Lk - 7), Lk][:B], dtype=torch.int32, device=DEV)
    idx = np.arange(B * H * Lq * Lk, dtype=np.uint64)
    mk = torch.from_numpy(dropout_scale(seed, idx, p).reshape(B, H, Lq, Lk)).to(DEV)
    if cross:
        out, lse = ops.flash_attn_fwd(q, k, v, kv_len=kv_len, drop_p=p, seed=seed)
        dq, dkv = ops.flash_attn_bwd(q, k, v, out, do, lse, kv_len=kv_len, drop_p=p, seed=seed)
        dk, dv = dkv[0], dkv[1]
    else:
        qkv = torch.stack([q, k, v], dim=2).reshape(B * Lq, 3 * H * hd).contiguous()
        out, lse = ops.flash_attn_fwd_packed(qkv, B, Lq, H, kv_len=kv_len, drop_p=p, seed=seed)
        dqkv = ops.flash_attn_bwd_packed(qkv, out, do.reshape(B * Lq, H * hd).contiguous(), lse, B, Lq, H, kv_len=kv_len, drop_p=p, seed=seed)
        out = out.view(B, Lq, H, hd)
        d5 = dqkv.view(B, Lq, 3, H, hd)
        dq, dk, dv = d5[:, :, 0], d5[:, :, 1], d5[:, :, 2]
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    s = torch.einsum("bihd,bjhd->bhij", qf, kf) * hd ** -0.5
    valid = torch.arange(Lk, device=DEV)[None, None, None, :] < kv_len[:, None, None, None]
    pr = torch.softmax(s.masked_fill(~valid, float("-inf")), dim=-1)
    ref = torch.einsum("bhij,bjhd->bihd", pr * mk, vf)
    ref.backward(do.float())
    assert rel(out, ref) < 1e-2
    assert rel(dq, qf.grad) < 2e-2 and rel(dk, kf.grad) < 2e-2 and rel(dv, vf.grad) < 2e-2


def _tiny_stage2(dropout=0.0, batch_text=False, seed=0, static=False, vis="mm88"):
    """the assembled stage-2 model on the fixture-sized configs (mm88 vision tower, bert_tiny text tower), weights from the oracle's generators;
    vis="mm64": the vision flavour WITHOUT separate image tables (image steps average the video tables over the frames)"""
    import dataclasses
    from internvideo_amd import mm_internvideo2 as mm, xbert
    from internvideo_amd.stage2 import InternVideo2_Stage2_visual
    scfg = O.named_config(vis)
    bcfg = dataclasses.replace(O.named_bert_config("bert_tiny"), encoder_width=scfg.embed_dim)
    torch.manual_seed(seed)
    vision = mm.PretrainInternVideo2(img_size=scfg.img_size, embed_dim=scfg.embed_dim, depth=scfg.depth, num_heads=scfg.num_heads,
                                     mlp_ratio=scfg.mlp_ratio, num_frames=scfg.num_frames, drop_path_rate=0.0,
                                     attn_pool_num_heads=scfg.attn_pool_num_heads, clip_embed_dim=scfg.clip_embed_dim,
                                     clip_teacher_embed_dim=scfg.clip_teacher_embed_dim, clip_teacher_final_dim=scfg.clip_teacher_final_dim,
                                     clip_return_layer=scfg.clip_return_layer, sep_image_video_pos_embed=scfg.sep_image_video_pos_embed)
    vision.load_state_dict(O.synthetic_params(scfg, seed=1), strict=True)
    pc = xbert.BertConfig(vocab_size=bcfg.vocab_size, hidden_size=bcfg.hidden_size, num_hidden_layers=bcfg.num_hidden_layers,
                          num_attention_heads=bcfg.num_attention_heads, intermediate_size=bcfg.intermediate_size,
                          max_position_embeddings=bcfg.max_position_embeddings, hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout,
                          fusion_layer=bcfg.fusion_layer, encoder_width=scfg.embed_dim)
    text_enc = xbert.BertForMaskedLM(pc)
    text_enc.load_state_dict(O.synthetic_bert_params(bcfg, seed=0), strict=False)
    config = dict(model=dict(vision_encoder=dict(clip_embed_dim=scfg.clip_embed_dim, img_size=scfg.img_size, num_frames=scfg.num_frames,
                                                 tubelet_size=1, patch_size=scfg.patch_size, video_mask_type="random", video_mask_ratio=0.5,
                                                 image_mask_type="random", image_mask_ratio=0.5, only_mask=True),
                             text_encoder=dict(d_model=bcfg.hidden_size), embed_dim=32, temp=0.07),
                  criterion=dict(loss_weight=dict(uta=0.0, vtc=1.0, vtm=1.0, mlm=1.0), vtm_hard_neg=True, mlm_masking_prob=0.5))
    tok = SimpleNamespace(pad_token_id=bcfg.pad_token_id, cls_token_id=bcfg.cls_token_id, mask_token_id=bcfg.mask_token_id)
    class _Static(InternVideo2_Stage2_visual):             # graph capture: the vision mask is a static input (the caller refreshes it between replays)
        static = None

        def encode_teacher(self, image):
            return self.static if self.static is not None else super().encode_teacher(image)
    model = (_Static if static else InternVideo2_Stage2_visual)(config, tok, True, vision_encoder=vision, text_encoder=text_enc).to(DEV).train()
    model.batch_text_passes = batch_text
    B, L = 8, 16
    ids, mask = O.synthetic_text_batch(bcfg, B, L, seed=2)
    text = SimpleNamespace(input_ids=torch.from_numpy(ids).to(DEV), attention_mask=torch.from_numpy(mask).to(DEV))
    g = torch.Generator().manual_seed(4)
    image = torch.randn(B, scfg.num_frames, 3, scfg.img_size, scfg.img_size, generator=g).to(DEV)
    if static:
        np.random.seed(0)
        m0, t_mid, t_fin = model.encode_teacher(image.permute(0, 2, 1, 3, 4))
        model.static = (m0, t_mid, t_fin)
        model.vision_encoder.static_visible_tokens = int((~m0[0]).sum())
        text.attention_mask._ivh_kv_len = text.attention_mask.sum(1, dtype=torch.int32).contiguous()     # right-padded by construction
    return model, image, text, torch.arange(B, device=DEV)


@pytest.mark.parametrize("batch_text,grouped", [(False, False), (True, True)])
def test_stage2_engine_step_matches_plain_autograd_and_torch_adamw(batch_text, grouped):
    """BASELINE configs[3] as a TRAINING STEP (VERDICT r3 missing 2 / next 5; multi_modality/tasks/pretrain.py:207-213 under the stage-2
    optimizer config scripts/pretraining/stage2/1B/config.py:97-102: AdamW betas (0.9, 0.98), weight decay 0.05 except 1-D / bias / `temp`,
    max_grad_norm 3).  The engine's flat-buffer step on the stage-2 model -- text / fusion tower gradients ACCUMULATED into zeroed bf16
    buffers (tied word embeddings, several passes through the same layers), the vision tower's written in place, one fused AdamW per
    region -- against the same model in plain autograd + torch.optim.AdamW on the groups multi_modality/utils/optimizer.py:18-31 builds.
    batch_text + grouped = the fast configuration of tools/bench_stage2.py (one text pass, one fusion pass, grouped weight gradients)."""
    import contextlib
    from internvideo_amd import functional as Fn, xbert
    from internvideo_amd.engine import IVTrainEngine
    lr, wd, clip = 1e-3, 0.05, 3.0
    gw = Fn.grouped_weight_grads if grouped else contextlib.nullcontext

    def run(model, image, text, idx):
        torch.manual_seed(11); np.random.seed(0); xbert._DROP_CALLS = 0      # same MLM draws, same hard negatives, same vision masks
        out = model(image, text, idx, media_type="video")
        return sum(out.values()), {k: v.detach().float().item() for k, v in out.items()}

    ref, image, text, idx = _tiny_stage2(batch_text=batch_text)
    w0 = {k: v.detach().clone() for k, v in ref.named_parameters()}
    loss_r, parts_r = run(ref, image, text, idx)
    with gw():
        loss_r.backward()
    g_ref = {k: p.grad.detach().float().clone() for k, p in ref.named_parameters() if p.grad is not None}
    skip = ref.no_weight_decay()
    groups = [dict(params=[p for n, p in ref.named_parameters() if not (p.dim() == 1 or n.endswith(".bias") or n in skip)], weight_decay=wd),
              dict(params=[p for n, p in ref.named_parameters() if (p.dim() == 1 or n.endswith(".bias") or n in skip)], weight_decay=0.0)]
    opt = torch.optim.AdamW(groups, lr=lr, betas=(0.9, 0.98), eps=1e-6)
    torch.nn.utils.clip_grad_norm_(ref.parameters(), clip)
    gn_ref = torch.sqrt(sum((g.double() ** 2).sum() for g in g_ref.values())).item()
    opt.step()

    mod, image, text, idx = _tiny_stage2(batch_text=batch_text)
    eng = IVTrainEngine(mod, lr=lr, betas=(0.9, 0.98), eps=1e-6, weight_decay=wd, max_grad_norm=clip)
    eng.group_text_wgrads = grouped
    named = dict(mod.named_parameters())
    assert eng.tower is mod.vision_encoder and eng.accumulate_outside_tower and eng.dropout_epoch is not None
    assert named["text_encoder.bert.embeddings.word_embeddings.weight"]._ivh_accum and named["temp"]._ivh_accum and named["vision_proj.weight"]._ivh_accum
    assert not named["vision_encoder.blocks.0.attn.qkv.weight"]._ivh_accum and not named["vision_encoder.patch_embed.proj.weight"]._ivh_accum
    try:
        for rep in range(2):                                # the second pass proves that the accumulating regions are zeroed per step
            eng.zero_grad()
            loss_e, parts_e = run(mod, image, text, idx)
            eng.backward(loss_e)
            eng._finish_reduce()
        assert all(abs(parts_e[k] - parts_r[k]) <= 1e-6 * max(1.0, abs(parts_r[k])) for k in parts_r), (parts_e, parts_r)
        assert all(p.grad is None for p in mod.parameters())
        worst = {}
        for k, gr in g_ref.items():
            got = named[k].main_grad.float().reshape(gr.shape)
            scale = gr.norm().item()
            if scale < 1e-12:
                assert got.norm().item() < 1e-6, k
                continue
            worst[k] = ((got - gr).norm() / gr.norm()).item()
        bad = {k: v for k, v in worst.items() if v > (2e-2 if named[k].main_grad.dtype == torch.bfloat16 else 1e-2)}
        assert not bad, dict(sorted(bad.items(), key=lambda kv: -kv[1])[:8])
        assert len(worst) > 150
        eng.optimizer_step()
        torch.cuda.synchronize()
        assert abs(eng.grad_norm.item() - gn_ref) < 1e-2 * gn_ref, (eng.grad_norm.item(), gn_ref)
        rp = dict(ref.named_parameters())
        num = den = 0.0
        for k, p in mod.named_parameters():
            num += float(((p.detach().float() - w0[k].float()) - (rp[k].detach().float() - w0[k].float())).double().pow(2).sum())
            den += float((rp[k].detach().float() - w0[k].float()).double().pow(2).sum())
        assert (num / den) ** 0.5 < 5e-2, (num / den) ** 0.5            # the UPDATE (not the weights) within 5 %: bf16 gradients in, AdamW's sign-like step
        assert torch.equal(eng.shadow, eng.master[:eng.n_mat].to(torch.bfloat16))
        # one more whole step through the generic entry point: finite loss, weights move
        before = eng.master.clone()
        out = eng.train_step_fn(lambda: run(mod, image, text, idx))
        assert torch.isfinite(out[0]).item() and not torch.equal(before, eng.master) and int(eng.dropout_epoch.item()) == 1
    finally:
        xbert.set_dropout_epoch(None)


def test_stage2_engine_image_step_folds_the_tower_pos_embed_gradient():
    """an IMAGE step of a stage-2 model without sep_image_video_pos_embed (the reference's stage2 / 6B config) builds its
    positional tables from the tower's own `pos_embed` / `clip_pos_embed` with torch ops (frame average, V:592-607), so their gradients arrive
    through plain autograd as `.grad` -- after the tower's backward node.  The engine must fold them into its buffers (they used to be dropped:
    the optimizer saw 0 and the stale .grad kept accumulating).  Checked against the same model in plain autograd."""
    from internvideo_amd import xbert
    from internvideo_amd.engine import IVTrainEngine

    def run(model, image, text, idx):
        torch.manual_seed(11); np.random.seed(0); xbert._DROP_CALLS = 0
        out = model(image[:, :1].contiguous(), text, idx, media_type="image")
        return sum(out.values())

    ref, image, text, idx = _tiny_stage2(vis="mm64")
    assert not ref.vision_encoder.sep_image_video_pos_embed
    run(ref, image, text, idx).backward()
    g_ref = {k: p.grad.detach().float().clone() for k, p in ref.named_parameters() if p.grad is not None}
    assert g_ref["vision_encoder.pos_embed"].norm().item() > 0

    mod, image, text, idx = _tiny_stage2(vis="mm64")
    eng = IVTrainEngine(mod, lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=0.05, max_grad_norm=3.0)
    named = dict(mod.named_parameters())
    try:
        for rep in range(2):                                # twice: a stale .grad would double the second pass
            eng.zero_grad()
            eng.backward(run(mod, image, text, idx))
            eng._finish_reduce()
            assert all(p.grad is None for p in mod.parameters()), [k for k, p in mod.named_parameters() if p.grad is not None][:5]
            for k in ("vision_encoder.pos_embed", "vision_encoder.clip_pos_embed", "vision_encoder.cls_token", "vision_encoder.patch_embed.proj.weight"):
                if k not in g_ref:
                    continue
                got, want = named[k].main_grad.float().reshape(g_ref[k].shape), g_ref[k]
                assert want.norm().item() > 0 and rel(got, want) < 2e-2, (k, rep, rel(got, want))
        before = eng.master.clone()
        eng.optimizer_step()
        torch.cuda.synchronize()
        pe = named["vision_encoder.pos_embed"]                    # a view of the fp32 master buffer
        o0 = (pe.data_ptr() - eng.master.data_ptr()) // 4
        assert not torch.equal(before[o0:o0 + pe.numel()], eng.master[o0:o0 + pe.numel()])       # the table moves: its gradient reached AdamW
    finally:
        xbert.set_dropout_epoch(None)


def test_dropout_epoch_gives_a_captured_graph_fresh_masks_on_every_replay():
    """VERDICT r3 missing 4: the dropout seed is a launch argument, frozen into a captured graph.  With a registered device-side epoch the
    mask of a call site is hash(seed + epoch * 0x9E3779B1, element) (csrc/common.h DropCfg): a graph that advances the epoch and runs a
    dropout kernel produces a different, host-predictable mask on every replay -- and BERT's configured dropout (0.1) no longer raises
    under capture."""
    from internvideo_amd import ops, xbert
    M, C, p, seed = 64, 256, 0.5, 0x1234567
    g = torch.Generator(device="cpu").manual_seed(3)
    a = torch.randn(M, C, generator=g).to(DEV).bfloat16()
    w = torch.ones(C, device=DEV); b = torch.zeros(C, device=DEV)
    epoch = torch.zeros(1, dtype=torch.int32, device=DEV)
    xbert.set_dropout_epoch(epoch)
    try:
        def body():
            epoch.add_(1)
            return ops.add_layernorm_fwd(a, None, w, b, 1e-12, drop_p=p, seed=seed)[0]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        epoch.zero_()
        gr = torch.cuda.CUDAGraph()
        from internvideo_amd.engine import _cyclic_gc_paused    # dead engines of earlier tests must not be collected mid-capture
        with _cyclic_gc_paused(), torch.cuda.graph(gr):
            y = body()
        outs = []
        for rep in range(3):
            gr.replay()
            torch.cuda.synchronize()
            e = int(epoch.item())
            assert e == rep + 1
            mk = torch.from_numpy(dropout_scale((seed + e * 0x9E3779B1) & 0xFFFFFFFF, np.arange(M * C, dtype=np.uint64), p).reshape(M, C)).to(DEV)
            want = ln_ref(a.float() * mk, w, b, 1e-12)
            assert rel(y, want) < 4e-3, rep
            outs.append(y.clone())
        assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])
        # the text tower with its configured dropout inside a captured step: allowed now, loss finite
        model, image, text, idx = _tiny_stage2(dropout=0.1, batch_text=True, static=True)
        from internvideo_amd import functional as Fn
        from internvideo_amd.engine import IVTrainEngine
        eng = IVTrainEngine(model, lr=1e-4)
        assert eng.dropout_epoch is not None

        def loss_fn():
            return sum(model(image, text, idx, media_type="video").values())
        torch.manual_seed(5); np.random.seed(0)
        eng.capture_fn(loss_fn)
        l1 = eng.train_step_graphed()[0].clone()
        l2 = eng.train_step_graphed()[0].clone()
        torch.cuda.synchronize()
        assert torch.isfinite(l1).item() and torch.isfinite(l2).item() and int(eng.dropout_epoch.item()) >= 2
    finally:
        xbert.set_dropout_epoch(None)


def test_stage2_model_forward_backward_all_four_losses():
    """the assembled stage-2 model (vision tower + text tower + heads) on a fixture-sized config: finite losses, gradients reach both
    towers and every head; VTM / MLM agree with the oracle evaluated on the same intermediate features and the same draws."""
    from internvideo_amd import mm_internvideo2 as mm, xbert
    from internvideo_amd.stage2 import InternVideo2_Stage2_visual
    scfg = O.named_config("mm88")
    bcfg = O.named_bert_config("bert_tiny")
    torch.manual_seed(0)
    vision = mm.PretrainInternVideo2(img_size=scfg.img_size, embed_dim=scfg.embed_dim, depth=scfg.depth, num_heads=scfg.num_heads,
                                     mlp_ratio=scfg.mlp_ratio, num_frames=scfg.num_frames, drop_path_rate=0.0,
                                     attn_pool_num_heads=scfg.attn_pool_num_heads, clip_embed_dim=scfg.clip_embed_dim,
                                     clip_teacher_embed_dim=scfg.clip_teacher_embed_dim, clip_teacher_final_dim=scfg.clip_teacher_final_dim,
                                     clip_return_layer=scfg.clip_return_layer, sep_image_video_pos_embed=scfg.sep_image_video_pos_embed)
    vision.load_state_dict(O.synthetic_params(scfg, seed=1), strict=True)
    pc = xbert.BertConfig(vocab_size=bcfg.vocab_size, hidden_size=bcfg.hidden_size, num_hidden_layers=bcfg.num_hidden_layers,
                          num_attention_heads=bcfg.num_attention_heads, intermediate_size=bcfg.intermediate_size,
                          max_position_embeddings=bcfg.max_position_embeddings, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                          fusion_layer=bcfg.fusion_layer, encoder_width=scfg.embed_dim)
    text_enc = xbert.BertForMaskedLM(pc)
    text_enc.load_state_dict(O.synthetic_bert_params(bcfg, seed=0), strict=False)
    config = dict(model=dict(vision_encoder=dict(clip_embed_dim=scfg.clip_embed_dim, img_size=scfg.img_size, num_frames=scfg.num_frames,
                                                 tubelet_size=1, patch_size=scfg.patch_size, video_mask_type="random", video_mask_ratio=0.5,
                                                 image_mask_type="random", image_mask_ratio=0.5, only_mask=True),
                             text_encoder=dict(d_model=bcfg.hidden_size), embed_dim=32, temp=0.07),
                  criterion=dict(loss_weight=dict(uta=0.0, vtc=1.0, vtm=1.0, mlm=1.0), vtm_hard_neg=True, mlm_masking_prob=0.5))
    tok = SimpleNamespace(pad_token_id=bcfg.pad_token_id, cls_token_id=bcfg.cls_token_id, mask_token_id=bcfg.mask_token_id)
    model = InternVideo2_Stage2_visual(config, tok, True, vision_encoder=vision, text_encoder=text_enc).to(DEV).train()
    B, L = 8, 16
    ids, mask = O.synthetic_text_batch(bcfg, B, L, seed=2)
    text = SimpleNamespace(input_ids=torch.from_numpy(ids).to(DEV), attention_mask=torch.from_numpy(mask).to(DEV))
    g = torch.Generator().manual_seed(4)
    image = torch.randn(B, scfg.num_frames, 3, scfg.img_size, scfg.img_size, generator=g).to(DEV)
    idx = torch.arange(B, device=DEV)
    np.random.seed(0)
    out = model(image, text, idx, media_type="video")
    assert set(out) == {"loss_uta", "loss_vtc", "loss_vtm", "loss_mlm"}
    for k in ("loss_vtc", "loss_vtm", "loss_mlm"):
        assert torch.isfinite(out[k]) and out[k].item() > 0, k
    assert out["loss_uta"].item() == 0.0
    sum(out.values()).backward()
    named = dict(model.named_parameters())
    for k in ("vision_encoder.blocks.0.attn.qkv.weight", "vision_encoder.patch_embed.proj.weight", "text_encoder.bert.embeddings.word_embeddings.weight",
              "text_encoder.bert.encoder.layer.0.attention.self.query.weight", "text_encoder.bert.encoder.layer.3.crossattention.self.key.weight",
              "text_encoder.cls.predictions.transform.dense.weight", "text_encoder.cls.predictions.bias", "vision_proj.weight",
              "text_proj.weight", "itm_head.weight", "itm_head.bias", "temp"):
        gr = named[k].grad
        assert gr is not None and torch.isfinite(gr.float()).all() and float(gr.float().abs().max()) > 0, k
    # MLM head against the oracle on the model's own vision tokens, with fixed draws
    rng = np.random.RandomState(1)
    draws = (rng.rand(B, L) < 0.5, rng.rand(B, L) < 0.8, rng.rand(B, L) < 0.5, rng.randint(0, bcfg.vocab_size, size=(B, L)).astype(np.int64))
    with torch.no_grad():
        vis = model.vision_encoder(image.permute(0, 2, 1, 3, 4), None, False, x_vis_only=True)
        got = model.criterion_mlm.mlm_loss(model.text_encoder, text, vis, None, draws=tuple(torch.from_numpy(np.asarray(d)) for d in draws))
    m_ids, m_labels = O.mlm_mask_tokens(ids, *draws, bcfg)
    pt = {k: v.detach().float().cpu() for k, v in model.text_encoder.state_dict().items()}
    want = O.mlm_loss(pt, bcfg, torch.from_numpy(m_ids), torch.from_numpy(m_labels), torch.from_numpy(mask), vis.float().cpu())
    assert abs(got.item() - want.item()) < 5e-3 * want.item()
    # batched text passes (one text-mode pass over [ids | masked ids], one fusion pass over VTM pairs + MLM rows) == the reference's call
    # structure, bit for bit, given the same draws and negatives
    from internvideo_amd import functional as Fn
    draws_t = tuple(torch.from_numpy(np.asarray(d)) for d in draws)
    negs = (torch.roll(torch.arange(B, device=DEV), 1), torch.roll(torch.arange(B, device=DEV), 3))
    from internvideo_amd import lib
    lib.load().ivh_gemm256_debug_split(0)       # a K split of a GEMM's last tile round depends on the row count: not the same bits for M and 2M rows
    with torch.no_grad():
        np.random.seed(7)
        model.clip_contrastive_temperature()
        ve, pooled, _, _, _, _ = model.encode_vision(image)
        te, pt = model.encode_text(text)
        vp = Fn.LinearFn.apply(pooled, model.vision_proj.weight, model.vision_proj.bias)
        tp = Fn.LinearFn.apply(pt, model.text_proj.weight, model.text_proj.bias)
        l_vtc = model.criterion_vtc_vtm.vtc_loss(vp, tp, idx, model.temp, all_gather=True)
        l_vtm = model.criterion_vtc_vtm.vtm_loss(model.get_text_encoder(), model.itm_head, model.temp, ve, te, vp, tp, text.attention_mask, idx,
                                                 neg_indices=negs)
        l_mlm = model.criterion_mlm.mlm_loss(model.text_encoder, text, ve, None, draws=draws_t)
        np.random.seed(7)
        ob = model._forward_batched_text(image, text, idx, mlm_draws=draws_t, neg_indices=negs)
    lib.load().ivh_gemm256_debug_split(1)
    assert torch.equal(ob["loss_vtc"], l_vtc) and torch.equal(ob["loss_vtm"], l_vtm) and torch.equal(ob["loss_mlm"], l_mlm)
    # weight gradients of the separate Linear nodes grouped at the end of the backward pass (functional.grouped_weight_grads) == the
    # node-by-node gradients up to the kernels' summation order
    import contextlib

    def grads_with(ctx):
        model.zero_grad(set_to_none=True)
        np.random.seed(7)
        o = model._forward_batched_text(image, text, idx, mlm_draws=draws_t, neg_indices=negs)
        with ctx:
            sum(o.values()).backward()
        return {k: q.grad.detach().clone() for k, q in named.items() if q.grad is not None}
    g_node, g_grouped = grads_with(contextlib.nullcontext()), grads_with(Fn.grouped_weight_grads())
    assert not Fn._end_pending and not Fn._wgrad_queue and set(g_node) == set(g_grouped)
    for k, g in g_node.items():
        assert rel(g_grouped[k].float(), g.float()) < 6e-3, k
    model.batch_text_passes = True
    model.zero_grad(set_to_none=True)
    np.random.seed(0)
    out2 = model(image, text, idx, media_type="video")
    sum(out2.values()).backward()
    assert all(torch.isfinite(v) for v in out2.values()) and named["text_encoder.bert.encoder.layer.0.attention.self.query.weight"].grad is not None


def test_gradient_checkpointing_of_the_text_tower_is_bit_identical_with_dropout_on():
    """`gradient_checkpointing = True  # for text encoder` of the shipped stage-2 configs (scripts/pretraining/stage2/1B/config.py:121 ->
    builder.py:23 -> xbert.py:743-765): every layer keeps only its input and runs again inside backward.  The kernels' dropout masks are
    functions of (seed, element index) and the re-run rewinds the host seed counter, so loss and EVERY gradient are bit-identical to the
    run that stores the activations -- with BERT's default dropout 0.1 on -- in text, fusion and multi_modal mode."""
    from internvideo_amd import xbert
    cfg = O.named_bert_config("bert_tiny")
    kw = dict(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
              num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size, max_position_embeddings=cfg.max_position_embeddings,
              hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, fusion_layer=cfg.fusion_layer, encoder_width=cfg.encoder_width,
              pad_token_id=cfg.pad_token_id)
    p = O.synthetic_bert_params(cfg, seed=0)
    ids, mask = O.synthetic_text_batch(cfg, 4, 12, seed=3)
    d_ids, d_mask = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    vision = torch.randn(4, 9, cfg.encoder_width, generator=torch.Generator().manual_seed(1)).to(DEV)
    res = {}
    for tag, cp in (("stored", False), ("recomputed", True)):
        model = xbert.BertForMaskedLM(xbert.BertConfig(gradient_checkpointing=cp, **kw))
        model.load_state_dict(p, strict=False)
        model = model.to(DEV).train()
        xbert._DROP_CALLS = 1000                                # same seed sequence in both runs
        text = model.bert(d_ids, attention_mask=d_mask, mode="text").last_hidden_state
        fused = model.bert(encoder_embeds=text, attention_mask=d_mask, encoder_hidden_states=vision, mode="fusion").last_hidden_state
        multi = model.bert(d_ids, attention_mask=d_mask, encoder_hidden_states=vision, mode="multi_modal").last_hidden_state
        loss = (fused.float() ** 2).mean() + (multi.float() * text.float()).mean()
        n_before = xbert._DROP_CALLS
        loss.backward()
        assert xbert._DROP_CALLS == n_before                    # the re-runs leave the seed counter where the forward left it
        res[tag] = (loss.detach().clone(), fused.detach().clone(), {k: v.grad.clone() for k, v in model.named_parameters() if v.grad is not None})
    assert torch.equal(res["stored"][0], res["recomputed"][0]) and torch.equal(res["stored"][1], res["recomputed"][1])
    assert res["stored"][2].keys() == res["recomputed"][2].keys() and len(res["stored"][2]) > 40
    for k, g in res["stored"][2].items():
        if "embeddings.weight" in k:                            # scatter-adds with fp32 atomics (word / position / type tables): the non-deterministic sums of the tower
            assert rel(res["recomputed"][2][k], g) < 1e-5, k
        else:
            assert torch.equal(g, res["recomputed"][2][k]), k
    model.eval()                                                # xbert.py:743: only while training
    with torch.no_grad():
        assert torch.isfinite(model.bert(d_ids, attention_mask=d_mask, mode="text").last_hidden_state.float()).all()


def test_frozen_text_tower_gets_no_weight_gradients_inside_grouped_weight_grads():
    """inside `grouped_weight_grads()` the Linear-like nodes bypass autograd for their weights, so THEY must honour
    requires_grad: with the text tower frozen (InternVideo2_Stage2_visual.freeze_text) no frozen parameter receives a .grad -- an optimizer
    built over all parameters must not move it -- while the trainable head still does, with the same values as without the context."""
    from internvideo_amd import functional as Fn, xbert
    cfg = O.named_bert_config("bert_tiny")
    pc = xbert.BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                          num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                          max_position_embeddings=cfg.max_position_embeddings, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                          fusion_layer=cfg.fusion_layer, encoder_width=cfg.encoder_width, pad_token_id=cfg.pad_token_id)
    model = xbert.BertForMaskedLM(pc)
    model.load_state_dict(O.synthetic_bert_params(cfg, seed=0), strict=False)
    model = model.to(DEV).train()
    head = torch.nn.Linear(cfg.hidden_size, 8).to(DEV)
    ids, mask = O.synthetic_text_batch(cfg, 8, 16, seed=3)
    d_ids, d_mask = torch.from_numpy(ids).to(DEV), torch.from_numpy(mask).to(DEV)
    vision = torch.randn(8, 9, cfg.encoder_width, generator=torch.Generator().manual_seed(1)).to(DEV).requires_grad_(True)

    def run(grouped):
        model.zero_grad(set_to_none=True); head.zero_grad(set_to_none=True); vision.grad = None
        h = model.bert(d_ids, attention_mask=d_mask, encoder_hidden_states=vision, mode="multi_modal").last_hidden_state
        loss = (Fn.LinearFn.apply(h, head.weight, head.bias).float() ** 2).mean()
        if grouped:
            with Fn.grouped_weight_grads():
                loss.backward()
        else:
            loss.backward()
        return head.weight.grad.clone(), vision.grad.clone()

    want_head, want_vis = run(False)
    for p in model.parameters():
        p.requires_grad = False
    got_head, got_vis = run(True)
    assert all(p.grad is None for p in model.parameters()), [k for k, p in model.named_parameters() if p.grad is not None][:5]
    assert not Fn._wgrad_queue and not Fn._end_pending
    assert rel(got_head.float(), want_head.float()) < 1e-6 and rel(got_vis.float(), want_vis.float()) < 1e-6   # gradients still flow THROUGH the frozen tower
    for p in model.parameters():
        p.requires_grad = True
