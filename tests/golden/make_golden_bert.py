"""Golden fixture for the stage-2 text / fusion tower, made by RUNNING THE REFERENCE'S OWN CODE on CPU:

    python tests/golden/make_golden_bert.py          (authoring container only: needs /root/reference and `transformers`)

  * `BertForMaskedLM` of multi_modality/models/backbones/bert/xbert.py (what builder.py:31-45 instantiates for stage 2), fixture-sized
    (oracle.named_bert_config("bert_tiny")), dropout 0: text-mode states, fusion-mode states, multi_modal states, MLM logits;
  * `MLMLoss.mask` and `MLMLoss.mlm_loss` of multi_modality/models/criterions.py:227-342 with the three Bernoulli draws and the
    random-word table fixed (torch.bernoulli / torch.randint are replaced by recorded draws while the reference method runs): masked
    ids, labels, the loss and its parameter gradients;
  * `VTC_VTM_Loss.vtm_loss` (criterions.py:105-182) with torch.multinomial replaced by arg-max (the hard negatives become a pure
    function of the weights): the sampling weights, the negative indices, the loss and its gradients.
Parameters / inputs are the deterministic synthetic ones of oracle.internvideo2_oracle; only draws and OUTPUTS are stored
(tests/golden/bert_tiny.npz).  The reference files are imported unmodified (tests/golden/ref_loader.py documents the shims for the
installed transformers version).
"""
from __future__ import annotations

import contextlib
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_loader  # noqa: E402
from oracle import internvideo2_oracle as O  # noqa: E402

B, L, LV = 4, 12, 9
GRADS = ["bert.embeddings.position_embeddings.weight", "bert.embeddings.token_type_embeddings.weight", "bert.embeddings.LayerNorm.weight",
         "bert.encoder.layer.0.attention.self.query.weight", "bert.encoder.layer.1.attention.self.value.bias",
         "bert.encoder.layer.1.output.dense.weight", "bert.encoder.layer.2.crossattention.self.key.weight",
         "bert.encoder.layer.3.crossattention.output.LayerNorm.bias", "bert.encoder.layer.3.intermediate.dense.bias",
         "cls.predictions.transform.dense.weight", "cls.predictions.transform.LayerNorm.weight", "cls.predictions.bias"]


def reference_state_dict(p, cfg):
    sd = dict(p)
    sd["bert.embeddings.position_ids"] = torch.arange(cfg.max_position_embeddings).expand((1, -1))
    sd["cls.predictions.decoder.weight"] = p["bert.embeddings.word_embeddings.weight"]
    sd["cls.predictions.decoder.bias"] = p["cls.predictions.bias"]
    return sd


def fixture_inputs(cfg):
    """everything random that the fixture needs, from fixed seeds (shared with tests/test_bert_oracle.py through the .npz)"""
    g = torch.Generator().manual_seed(11)
    ids, mask = O.synthetic_text_batch(cfg, B, L, seed=3)
    rng = np.random.RandomState(5)
    d = dict(ids=ids, mask=mask,
             vision=(0.5 * torch.randn(B, LV, cfg.encoder_width, generator=g)).numpy(),
             draw_mask=(rng.rand(B, L) < 0.5), draw_replace=(rng.rand(B, L) < 0.8), draw_random=(rng.rand(B, L) < 0.5),
             random_words=rng.randint(0, cfg.vocab_size, size=(B, L)).astype(np.int64),
             vision_proj=torch.nn.functional.normalize(torch.randn(B, 32, generator=g), dim=-1).numpy(),
             text_proj=torch.nn.functional.normalize(torch.randn(B, 32, generator=g), dim=-1).numpy(),
             idx=np.array([7, 3, 7, 1], dtype=np.int64), temp=np.array([0.07], dtype=np.float32),
             itm_w=(0.1 * torch.randn(2, cfg.hidden_size, generator=g)).numpy(), itm_b=np.array([0.03, -0.02], dtype=np.float32))
    return d


@contextlib.contextmanager
def recorded_draws(bernoulli_seq=None, randint_value=None, multinomial_argmax=False, log=None):
    """replace the random draws of the reference's loss code by recorded values while it runs"""
    orig = (torch.bernoulli, torch.randint, torch.multinomial)
    seq = list(bernoulli_seq or [])

    def bern(pm, *a, **k):
        return torch.from_numpy(seq.pop(0).astype(np.float32))

    def rint(*a, **k):
        return torch.from_numpy(randint_value)

    def multi(w, n, *a, **k):
        if log is not None:
            log.append(w.detach().clone())
        return w.argmax(dim=1, keepdim=True)
    if bernoulli_seq is not None:
        torch.bernoulli = bern
    if randint_value is not None:
        torch.randint = rint
    if multinomial_argmax:
        torch.multinomial = multi
    try:
        yield
    finally:
        torch.bernoulli, torch.randint, torch.multinomial = orig


def main():
    assert ref_loader.available(), "needs /root/reference"
    cfg = O.named_bert_config("bert_tiny")
    p = O.synthetic_bert_params(cfg, seed=0)
    model = ref_loader.build_reference_bert(cfg)
    missing = model.load_state_dict(reference_state_dict(p, cfg), strict=True)
    assert model.cls.predictions.decoder.weight is model.bert.embeddings.word_embeddings.weight
    model.train()                                                            # dropout probabilities are 0
    d = fixture_inputs(cfg)
    ids, mask = torch.from_numpy(d["ids"]), torch.from_numpy(d["mask"])
    vision = torch.from_numpy(d["vision"])
    out = {}

    # ---- forward modes (encode_text S2:271-289; fusion as vtm / mlm call it) ----
    with torch.no_grad():
        text = model.bert(ids, attention_mask=mask, return_dict=True, mode="text").last_hidden_state
        fused = model.bert(encoder_embeds=text, attention_mask=mask, encoder_hidden_states=vision, encoder_attention_mask=None,
                           return_dict=True, mode="fusion").last_hidden_state
        multi = model.bert(ids, attention_mask=mask, encoder_hidden_states=vision,
                           encoder_attention_mask=torch.ones(B, LV, dtype=torch.long), return_dict=True, mode="multi_modal").last_hidden_state
        logits = model(encoder_embeds=text, attention_mask=mask, encoder_hidden_states=vision, encoder_attention_mask=None,
                       return_dict=True, mode="fusion", return_logits=True)
    out.update(text=text.numpy(), fused=fused.numpy(), multi=multi.numpy(), mlm_logits=logits.numpy())

    # ---- MLM (criterions.py:227-342) ----
    crit = ref_loader.load_mm_criterions()
    tok = SimpleNamespace(pad_token_id=cfg.pad_token_id, cls_token_id=cfg.cls_token_id, mask_token_id=cfg.mask_token_id)
    mlm = crit.MLMLoss(0.5, tok)
    with recorded_draws([d["draw_mask"], d["draw_replace"], d["draw_random"]], d["random_words"]):
        m_ids, m_labels = mlm.mask(ids.clone(), cfg.vocab_size, ids.device, targets=ids.clone(),
                                   probability_matrix=torch.full(ids.shape, 0.5))
    out.update(mlm_ids=m_ids.numpy(), mlm_labels=m_labels.numpy())
    model.zero_grad()
    text_in = SimpleNamespace(input_ids=ids, attention_mask=mask)
    with recorded_draws([d["draw_mask"], d["draw_replace"], d["draw_random"]], d["random_words"]):
        loss = mlm.mlm_loss(model, text_in, vision, None)
    loss.backward()
    out["mlm_loss"] = np.array([loss.item()])
    params = dict(model.named_parameters())
    for k in GRADS:
        out["mlm_grad:" + k] = params[k].grad.detach().numpy().copy()
    gw = params["bert.embeddings.word_embeddings.weight"].grad.detach()
    out["mlm_gradnorm:word"] = np.array([gw.double().norm().item()])
    out["mlm_grad:word_rows"] = gw[:16].numpy().copy()                       # pad row (0) has no gradient from the lookup, only from the tied decoder

    # ---- VTM (criterions.py:105-182) ----
    vtm = crit.VTC_VTM_Loss(True)
    head = nn.Linear(cfg.hidden_size, 2)
    with torch.no_grad():
        head.weight.copy_(torch.from_numpy(d["itm_w"])); head.bias.copy_(torch.from_numpy(d["itm_b"]))
    model.zero_grad()
    v_emb = vision.clone().requires_grad_(True)
    t_emb = text.clone().requires_grad_(True)
    log = []
    with recorded_draws(multinomial_argmax=True, log=log):
        lv = vtm.vtm_loss(model.bert, head, torch.tensor(float(d["temp"][0])), v_emb, t_emb, torch.from_numpy(d["vision_proj"]),
                          torch.from_numpy(d["text_proj"]), mask, torch.from_numpy(d["idx"]))
    lv.backward()
    out.update(vtm_loss=np.array([lv.item()]), vtm_weights_t2v=log[0].numpy(), vtm_weights_v2t=log[1].numpy(),
               vtm_vision_neg=log[0].argmax(1).numpy(), vtm_text_neg=log[1].argmax(1).numpy(),
               vtm_grad_vision=v_emb.grad.numpy(), vtm_grad_text=t_emb.grad.numpy(), vtm_grad_itm_w=head.weight.grad.numpy(),
               vtm_grad_itm_b=head.bias.grad.numpy())
    for k in ("bert.encoder.layer.2.crossattention.self.key.weight", "bert.encoder.layer.3.output.dense.weight"):
        out["vtm_grad:" + k] = params[k].grad.detach().numpy().copy()

    for k, v in d.items():
        out["in:" + k] = np.asarray(v)
    path = os.path.join(HERE, "bert_tiny.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB;  mlm loss", float(loss), " vtm loss", float(lv),
          " masked tokens", int((out["mlm_labels"] != -100).sum()), " negs", out["vtm_vision_neg"], out["vtm_text_neg"])


if __name__ == "__main__":
    main()
