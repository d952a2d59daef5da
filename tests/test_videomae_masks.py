"""VideoMAE mask generators against what the reference's own classes return under the same numpy seeds
(tests/golden/make_golden_videomae_masks.py; InternVideo1/Pretrain/VideoMAE/masking_generator.py).  Integer work: bit-exact, dtypes too."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from internvideo_amd import videomae_masking as VM

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "videomae_masks.npz")


def _cases():
    spec = importlib.util.spec_from_file_location("_mk_vm_masks", os.path.join(HERE, "golden", "make_golden_videomae_masks.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    return mk.CASES, mk.SEEDS


CASES, SEEDS = _cases()


@pytest.mark.parametrize("case", sorted(CASES))
def test_generator_reproduces_the_reference_under_the_same_seed(case):
    g = np.load(GOLD)
    cls, ctor, call = CASES[case]
    gen = getattr(VM, cls)(*ctor)
    assert repr(gen).encode() == bytes(g[f"{case}:repr"])    # the log line the reference prints for this generator
    if f"{case}:keep" in g:
        assert np.array_equal(np.asarray(gen.keep_patches_list), g[f"{case}:keep"])
    if f"{case}:maps" in g:
        assert np.array_equal(gen.all_mask_maps, g[f"{case}:maps"])
    for seed in SEEDS:
        np.random.seed(seed)
        for n in range(2):                                   # the second call checks that the RNG was consumed identically by the first
            out = gen(*call)
            want = g[f"{case}:{seed}:{n}"]
            got = out.numpy() if isinstance(out, torch.Tensor) else np.asarray(out)
            assert got.dtype == want.dtype and got.shape == want.shape, (case, got.dtype, want.dtype)
            assert np.array_equal(got, want), (case, seed, n)


def test_progressive_masks_have_the_structure_the_encoder_needs():
    """size-independent properties at the recipe's geometry (8 x 14 x 14, ratio 0.9): every sample shows the same number of patches (the
    encoder gathers equal counts per row), frame i shows keep_i - 1 (the threshold element itself is masked), and a frame that keeps fewer
    shows a subset of one that keeps more."""
    for cls in (VM.TemporalProgressiveMaskingGenerator, VM.TemporalCenteringProgressiveMaskingGenerator):
        gen = cls((8, 14, 14), 0.9)
        np.random.seed(3)
        masks = np.stack([gen() for _ in range(4)]).reshape(4, 8, 196)
        vis = (masks == 0)
        assert (vis.sum(2) == np.asarray(gen.keep_patches_list)[None] - 1).all()
        assert len(set(vis.reshape(4, -1).sum(1).tolist())) == 1
        order = np.argsort(-np.asarray(gen.keep_patches_list), kind="stable")
        for a, b in zip(order[:-1], order[1:]):
            assert not (vis[:, b] & ~vis[:, a]).any()
    with pytest.raises(IndexError):                          # 4 x 4 grid: int(0.05 * 16) = 0 kept -- the reference's top-k fails the same way
        VM.TemporalProgressiveMaskingGenerator((2, 4, 4), 0.5)()


def test_mask_type_switch():
    assert isinstance(VM.build_mask_generator("t_consist", [8, 14, 14], 0.9), VM.TemporalConsistencyMaskingGenerator)
    g = VM.build_mask_generator("random", (8, 14, 14), 0.9)
    np.random.seed(0)
    m = g()
    assert m.shape == (1568,) and int(m.sum()) == int(0.9 * 1568)
    with pytest.raises(ValueError):
        VM.build_mask_generator("tube", (8, 14, 14), 0.9)
    with pytest.raises(AssertionError):
        VM.CellRunningMaskingGenerator((2, 2, 2), 0.2)       # int(4 * 0.2) = 0 masked positions per cell


def test_single_modality_generators_log_like_the_reference():
    """InternVideo2/single_modality/datasets/masking_generator.py:12-16, 38-41 (run_pretraining.py prints the generator)"""
    from internvideo_amd import masking
    g = np.load(GOLD)
    assert repr(masking.TubeMaskingGenerator((8, 16, 16), 0.8)).encode() == bytes(g["sm_tube:repr"])
    assert repr(masking.RandomMaskingGenerator((8, 16, 16), 0.8)).encode() == bytes(g["sm_random:repr"])
