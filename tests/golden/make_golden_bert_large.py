"""Config-size golden digest for the stage-2 text / fusion tower: the REFERENCE's own `BertForMaskedLM`
(multi_modality/models/backbones/bert/xbert.py) at BERT-large with fusion_layer 19 and 1408-wide cross-attention (BASELINE configs[3],
scripts/pretraining/stage2/1B/config.py), fp32 CPU, dropout 0:

    python tests/golden/make_golden_bert_large.py      (authoring container only: needs /root/reference and `transformers`)

Inputs = what tests/test_fullsize_gpu.py::test_bert_large_text_and_fusion_tower_match_oracle_at_config_size feeds the oracle and the HIP
path (synthetic_bert_params(seed 0, std 0.02); B = 2, 32 text tokens, 206 vision tokens; the MLM draws of RandomState(3)).  Stored
(tests/golden/bert_large_digest.npz): text-mode and fusion-mode states (first three rows in full + 16 fixed random projections of every
row), the MLM loss of criterions.py:227-342 under the recorded draws, and corners / norms of sampled parameter gradients.
"""
from __future__ import annotations

import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_loader  # noqa: E402
from make_golden_bert import recorded_draws, reference_state_dict  # noqa: E402
from oracle import internvideo2_oracle as O  # noqa: E402

B, L, LV = 2, 32, 206
KEYS = ["bert.embeddings.position_embeddings.weight", "bert.encoder.layer.0.attention.self.query.weight", "bert.encoder.layer.11.output.dense.weight",
        "bert.encoder.layer.19.crossattention.self.key.weight", "bert.encoder.layer.23.crossattention.output.dense.weight",
        "bert.encoder.layer.23.output.LayerNorm.weight", "cls.predictions.transform.dense.weight", "cls.predictions.bias",
        "bert.embeddings.word_embeddings.weight"]


def projection(C: int) -> np.ndarray:
    return np.random.Generator(np.random.PCG64(777 + C)).standard_normal((C, 16)).astype(np.float32) / np.sqrt(C).astype(np.float32)


def inputs(cfg):
    ids, mask = O.synthetic_text_batch(cfg, B, L, seed=1)
    g = torch.Generator().manual_seed(2)
    vision = torch.randn(B, LV, cfg.encoder_width, generator=g)
    rng = np.random.RandomState(3)
    draws = (rng.rand(B, L) < 0.5, rng.rand(B, L) < 0.8, rng.rand(B, L) < 0.5, rng.randint(0, cfg.vocab_size, size=(B, L)).astype(np.int64))
    return ids, mask, vision, draws


def main():
    assert ref_loader.available(), "needs /root/reference"
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = O.named_bert_config("bert_large_1B")
    p = O.synthetic_bert_params(cfg, seed=0, std=0.02)
    model = ref_loader.build_reference_bert(cfg)
    model.load_state_dict(reference_state_dict(p, cfg), strict=True)
    model.train()                                                            # dropout probabilities are 0
    ids, mask, vision, draws = inputs(cfg)
    t_ids, t_mask = torch.from_numpy(ids), torch.from_numpy(mask)
    out = {"meta": np.array([B, L, LV], dtype=np.int64)}
    with torch.no_grad():
        text = model.bert(t_ids, attention_mask=t_mask, return_dict=True, mode="text").last_hidden_state
        fused = model.bert(encoder_embeds=text, attention_mask=t_mask, encoder_hidden_states=vision, encoder_attention_mask=None,
                           return_dict=True, mode="fusion").last_hidden_state
    for name, t in (("text", text), ("fused", fused)):
        rows = t.double().numpy().reshape(-1, t.shape[-1])
        out[name + ":rows"] = rows[:3].astype(np.float32)
        out[name + ":proj"] = (rows @ projection(rows.shape[1]).astype(np.float64)).astype(np.float32)
    crit = ref_loader.load_mm_criterions()
    tok = SimpleNamespace(pad_token_id=cfg.pad_token_id, cls_token_id=cfg.cls_token_id, mask_token_id=cfg.mask_token_id)
    mlm = crit.MLMLoss(0.5, tok)
    model.zero_grad()
    with recorded_draws([draws[0], draws[1], draws[2]], draws[3]):
        loss = mlm.mlm_loss(model, SimpleNamespace(input_ids=t_ids, attention_mask=t_mask), vision, None)
    loss.backward()
    out["mlm_loss"] = np.array([loss.item()], dtype=np.float64)
    named = dict(model.named_parameters())
    for k in KEYS:
        g = named[k].grad.detach()
        g2 = g.reshape(g.shape[0], -1) if g.dim() > 1 else g.reshape(1, -1)
        out["grad:" + k + ":corner"] = g2[:16, :16].numpy().copy()
        out["grad:" + k + ":norm"] = np.array([g.double().norm().item()], dtype=np.float64)
    path = os.path.join(HERE, "bert_large_digest.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; mlm loss", loss.item())


if __name__ == "__main__":
    main()
