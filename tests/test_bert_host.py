"""Host-side logic of the stage-2 text / fusion tower (no GPU): parameter names against the reference's state_dict, the token masking of
MLMLoss against the reference's recorded output, attention-mask handling, and the no-CPU-path rule."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import internvideo2_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden", "bert_tiny.npz")


def tiny():
    from internvideo_amd import xbert
    cfg = O.named_bert_config("bert_tiny")
    pc = xbert.BertConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                          num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                          max_position_embeddings=cfg.max_position_embeddings, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                          fusion_layer=cfg.fusion_layer, encoder_width=cfg.encoder_width)
    return cfg, pc, xbert.BertForMaskedLM(pc)


def test_state_dict_names_and_shapes_are_the_references():
    cfg, pc, model = tiny()
    sd = model.state_dict()
    want = dict(O.bert_param_shapes(cfg))                       # pinned to the reference's BertForMaskedLM by tests/golden/make_golden_bert.py (strict load)
    want["bert.embeddings.position_ids"] = (1, cfg.max_position_embeddings)
    want["cls.predictions.decoder.weight"] = (cfg.vocab_size, cfg.hidden_size)
    want["cls.predictions.decoder.bias"] = (cfg.vocab_size,)
    assert set(sd) == set(want)
    for k, shp in want.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    assert model.cls.predictions.decoder.weight is model.bert.embeddings.word_embeddings.weight
    assert model.cls.predictions.decoder.bias is model.cls.predictions.bias
    missing, unexpected = model.load_state_dict(O.synthetic_bert_params(cfg), strict=False)
    assert not unexpected
    n_large = sum(int(np.prod(s)) for s in O.bert_param_shapes(O.named_bert_config("bert_large_1B")).values())
    assert 355e6 < n_large < 365e6


def test_mlm_token_masking_matches_the_reference_bit_for_bit():
    from internvideo_amd.stage2 import MLMLoss
    gold = np.load(G)
    cfg = O.named_bert_config("bert_tiny")
    tok = SimpleNamespace(pad_token_id=cfg.pad_token_id, cls_token_id=cfg.cls_token_id, mask_token_id=cfg.mask_token_id)
    crit = MLMLoss(0.5, tok)
    ids = torch.from_numpy(gold["in:ids"])
    draws = tuple(torch.from_numpy(gold["in:" + k]) for k in ("draw_mask", "draw_replace", "draw_random", "random_words"))
    m_ids, m_labels = crit.mask(ids.clone(), cfg.vocab_size, ids.device, targets=ids.clone(), draws=draws)
    assert np.array_equal(m_ids.numpy(), gold["mlm_ids"]) and np.array_equal(m_labels.numpy(), gold["mlm_labels"])
    # random draws: statistics of the 80 / 10 / 10 rule on a large batch, never [PAD] / [CLS]
    torch.manual_seed(0)
    big, bmask = O.synthetic_text_batch(cfg, 512, 32, seed=9)
    big = torch.from_numpy(big)
    out, lab = crit.mask(big.clone(), cfg.vocab_size, big.device, targets=big.clone(), probability_matrix=torch.full(big.shape, 0.5))
    kept = lab != -100
    valid = (big != cfg.pad_token_id) & (big != cfg.cls_token_id)
    assert not kept[~valid].any()
    frac = kept[valid].float().mean().item()
    assert 0.47 < frac < 0.53
    assert 0.77 < (out[kept] == cfg.mask_token_id).float().mean().item() < 0.83
    assert torch.equal(out[~kept], big[~kept]) and torch.equal(lab[kept], big[kept])


def test_right_padded_lengths():
    from internvideo_amd.lib import InternVideoHipError
    from internvideo_amd.xbert import right_padded_lengths
    assert right_padded_lengths(None, "m") is None
    assert right_padded_lengths(torch.ones(3, 5, dtype=torch.long), "m") is None
    m = torch.tensor([[1, 1, 1, 0], [1, 0, 0, 0], [1, 1, 1, 1]])
    n = right_padded_lengths(m, "m")
    assert n.dtype == torch.int32 and n.tolist() == [3, 1, 4]
    assert right_padded_lengths(m, "m") is n                    # cached on the mask tensor: one host read per batch
    with pytest.raises(InternVideoHipError):
        right_padded_lengths(torch.tensor([[1, 0, 1, 0]]), "m")
    with pytest.raises(InternVideoHipError):
        right_padded_lengths(torch.tensor([[0, 0, 0, 0], [1, 1, 0, 0]]), "m")


def test_no_cpu_path_and_unsupported_features_raise():
    from internvideo_amd import xbert
    from internvideo_amd.lib import InternVideoHipError
    cfg, pc, model = tiny()
    ids, mask = O.synthetic_text_batch(cfg, 2, 8, seed=0)
    with pytest.raises(InternVideoHipError):
        model.bert(torch.from_numpy(ids), attention_mask=torch.from_numpy(mask), mode="text")
    with pytest.raises(InternVideoHipError):
        model.bert(torch.from_numpy(ids), head_mask=torch.ones(4), mode="text")
    with pytest.raises(InternVideoHipError):
        xbert.BertConfig(hidden_act="relu")
    with pytest.raises(InternVideoHipError):
        xbert.BertConfig(position_embedding_type="relative_key")
    with pytest.raises(InternVideoHipError):
        xbert.BertModel(pc, add_pooling_layer=True)


def test_build_bert_follows_the_model_config():
    from internvideo_amd import xbert
    tiny_json = dict(vocab_size=64, hidden_size=64, num_hidden_layers=3, num_attention_heads=1, intermediate_size=128,
                     max_position_embeddings=16, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    mc = dict(text_encoder=dict(name="bert_large", config=dict(tiny_json), fusion_layer=2), vision_encoder=dict(d_model=48),
              multimodal=dict(enable=True))
    m = xbert.build_bert(mc, pretrain=True)
    assert isinstance(m, xbert.BertForMaskedLM) and m.config.encoder_width == 48 and m.config.fusion_layer == 2
    assert "bert.encoder.layer.2.crossattention.self.key.weight" in m.state_dict()
    assert m.state_dict()["bert.encoder.layer.2.crossattention.self.key.weight"].shape == (64, 48)
    assert "bert.encoder.layer.1.crossattention.self.key.weight" not in m.state_dict()
    mc["multimodal"]["enable"] = False
    mc["text_encoder"]["config"] = dict(tiny_json)
    m2 = xbert.build_bert(mc, pretrain=False)
    assert isinstance(m2, xbert.BertModel) and m2.config.fusion_layer == 3
    assert not any("crossattention" in k for k in m2.state_dict())


def test_stage2_model_parameter_names():
    from internvideo_amd import xbert
    from internvideo_amd.stage2 import InternVideo2_Stage2_visual
    cfg, pc, text_enc = tiny()
    vision = torch.nn.Module()
    vision.patch_embed = torch.nn.Module()
    config = dict(model=dict(vision_encoder=dict(clip_embed_dim=24), text_encoder=dict(d_model=cfg.hidden_size), embed_dim=16, temp=0.9),
                  criterion=dict(loss_weight=dict(uta=0.0, vtc=1.0, vtm=1.0, mlm=1.0)))
    tok = SimpleNamespace(pad_token_id=0, cls_token_id=5, mask_token_id=7)
    m = InternVideo2_Stage2_visual(config, tok, True, vision_encoder=vision, text_encoder=text_enc)
    names = {k for k, _ in m.named_parameters()}
    assert {"vision_proj.weight", "vision_proj.bias", "text_proj.weight", "text_proj.bias", "temp", "itm_head.weight", "itm_head.bias"} <= names
    assert "text_encoder.bert.encoder.layer.0.attention.self.query.weight" in names
    assert m.itm_head.weight.shape == (2, cfg.hidden_size) and m.vision_proj.weight.shape == (16, 24)
    assert m.get_text_encoder() is text_enc.bert
    m.clip_contrastive_temperature()
    assert abs(m.temp.item() - 0.5) < 1e-7


# the `model` / `criterion` / `gradient_checkpointing` keys of InternVideo2/multi_modality/scripts/pretraining/stage2/1B/config.py (:41-101, :121)
# with configs/model.py's TextEncoders["bert_large"] substituted for "${TextEncoders[${text_enc}]}" and configs/config_bert_large.json's values
BERT_LARGE_JSON = dict(architectures=["BertForMaskedLM"], attention_probs_dropout_prob=0.1, gradient_checkpointing=False, hidden_act="gelu",
                       hidden_dropout_prob=0.1, hidden_size=1024, initializer_range=0.02, intermediate_size=4096, layer_norm_eps=1e-12,
                       max_position_embeddings=512, model_type="bert", num_attention_heads=16, num_hidden_layers=24, pad_token_id=0,
                       position_embedding_type="absolute", type_vocab_size=2, use_cache=True, vocab_size=30522, fusion_layer=19, encoder_width=768,
                       cross_module="ca")


def shipped_stage2_1B_config(bert_json_path: str, num_frames: int = 4):
    return dict(
        model=dict(
            model_cls="InternVideo2_Stage2",
            vision_encoder=dict(name="pretrain_internvideo2_1b_patch14_224", img_size=224, num_frames=num_frames, tubelet_size=1, patch_size=14,
                                d_model=1408, clip_embed_dim=768, clip_teacher_embed_dim=3200, clip_teacher_final_dim=768, clip_norm_type="l2",
                                clip_return_layer=6, clip_student_return_interval=1, pretrained=None, use_checkpoint=False, checkpoint_num=40,
                                use_flash_attn=True, use_fused_rmsnorm=True, use_fused_mlp=True, clip_teacher=None, clip_input_resolution=224,
                                clip_teacher_return_interval=1, video_mask_type="random", video_mask_ratio=0.8, image_mask_type="random",
                                image_mask_ratio=0.5, sep_image_video_pos_embed=True, keep_temporal=False, only_mask=True),
            text_encoder=dict(name="bert_large", pretrained="bert-large-uncased", config=bert_json_path, d_model=1024, fusion_layer=19),
            multimodal=dict(enable=True), embed_dim=512, temp=0.07, find_unused_parameters=False),
        criterion=dict(loss_weight=dict(vtc=1.0, mlm=1.0, vtm=1.0, mvm=0.0, uta=0.0), vtm_hard_neg=True, mlm_masking_prob=0.5,
                       distill_final_features=True, clip_loss_ratio=[1., 1.]),
        gradient_checkpointing=True)


def test_shipped_stage2_1B_config_constructs_unmodified(tmp_path):
    """BASELINE configs[3] through the reference's own config keys -- including `gradient_checkpointing = True  # for text encoder`
    (config.py:121 -> internvideo2_stage2_visual.py:351 -> builder.py:23): the model builds (meta device: 1.41 G parameters, no storage),
    the text tower honours the flag, and the parameter names are the reference checkpoint's."""
    import json
    from internvideo_amd.stage2 import InternVideo2_Stage2_visual
    path = tmp_path / "config_bert_large.json"
    path.write_text(json.dumps(BERT_LARGE_JSON))
    tok = SimpleNamespace(pad_token_id=0, cls_token_id=101, mask_token_id=103)
    with torch.device("meta"):
        m = InternVideo2_Stage2_visual(shipped_stage2_1B_config(str(path)), tok, True)
    bc = m.text_encoder.config
    assert bc.gradient_checkpointing is True and bc.encoder_width == 1408 and bc.fusion_layer == 19 and bc.num_hidden_layers == 24
    n_vis = sum(p.numel() for p in m.vision_encoder.parameters())
    n_txt = sum(p.numel() for p in m.text_encoder.parameters())
    assert 1.03e9 < n_vis < 1.07e9 and 3.5e8 < n_txt < 3.8e8
    names = set(dict(m.named_parameters()))
    for k in ("temp", "vision_proj.weight", "text_proj.bias", "itm_head.weight", "vision_encoder.blocks.39.mlp.fc2.weight",
              "vision_encoder.img_pos_embed", "text_encoder.bert.encoder.layer.19.crossattention.self.key.weight",
              "text_encoder.cls.predictions.transform.LayerNorm.weight"):
        assert k in names, k
    assert "text_encoder.bert.encoder.layer.18.crossattention.self.key.weight" not in names
    assert (m.video_mask_type, m.video_mask_ratio, m.video_window_size) == ("random", 0.8, (4, 16, 16))
    # multimodal.enable = False turns every layer into a text layer (builder.py:26-27); finetuning builds the bare BertModel
    cfg2 = shipped_stage2_1B_config(str(path))
    cfg2["model"]["multimodal"] = dict(enable=False)
    from internvideo_amd.xbert import build_bert, BertModel
    with torch.device("meta"):
        t2 = build_bert(cfg2["model"], False, True)
    assert isinstance(t2, BertModel) and t2.config.fusion_layer == 24 and t2.config.gradient_checkpointing
