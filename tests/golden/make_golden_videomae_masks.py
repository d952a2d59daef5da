"""Golden fixture for the VideoMAE mask generators.  Authoring container only.

Loads the REFERENCE's InternVideo1/Pretrain/VideoMAE/masking_generator.py by file path and records what its generators return under fixed
numpy seeds.  The two progressive generators spell their mask dtype `np.bool`, an alias numpy removed in 1.24 (this container has 2.2):
the alias is put back for the duration of this script -- nothing in the repo depends on it.

    python tests/golden/make_golden_videomae_masks.py      ->  tests/golden/videomae_masks.npz
       <case>:<seed>:<call>     the array a call returned (two consecutive calls per seed: the second one checks RNG consumption)
       <case>:keep              keep_patches_list of the progressive generators
"""
import importlib.util
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("IV_REFERENCE_ROOT", "/root/reference")

# case: (class name, ctor args, call args)
CASES = {
    "random": ("RandomMaskingGenerator", ((4, 6, 6), 0.75), ()),
    "random_int": ("RandomMaskingGenerator", (5, 0.9), ()),
    "t_consist": ("TemporalConsistencyMaskingGenerator", ((4, 6, 6), 0.75), ()),
    "t_progressive": ("TemporalProgressiveMaskingGenerator", ((8, 14, 14), 0.75), ()),
    "t_progressive_small": ("TemporalProgressiveMaskingGenerator", ((4, 6, 6), 0.5), ()),
    "t_center_prog": ("TemporalCenteringProgressiveMaskingGenerator", ((8, 14, 14), 0.9), ()),
    "t_center_prog_odd": ("TemporalCenteringProgressiveMaskingGenerator", ((6, 8, 8), 0.8), ()),
    "cell": ("CellRunningMaskingGenerator", ((4, 4, 6), 0.5), (5,)),
    "cell75": ("CellRunningMaskingGenerator", ((3, 2, 2), 0.75, False), (3,)),
    "decode": ("RandomDecodeMaskingGenerator", ((4, 4, 4), 0.5), (3,)),
}
SEEDS = (0, 7)


def load_reference():
    if not hasattr(np, "bool"):
        np.bool = bool                                   # see the header
    path = os.path.join(REF, "InternVideo1", "Pretrain", "VideoMAE", "masking_generator.py")
    spec = importlib.util.spec_from_file_location("_iv_ref_videomae_masks", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    d = {}
    for case, (cls, ctor, call) in CASES.items():
        g = getattr(ref, cls)(*ctor)
        if hasattr(g, "keep_patches_list"):
            d[f"{case}:keep"] = np.asarray(g.keep_patches_list)
        if hasattr(g, "all_mask_maps"):
            d[f"{case}:maps"] = np.asarray(g.all_mask_maps)
        d[f"{case}:repr"] = np.frombuffer(repr(g).encode(), dtype=np.uint8)
        for seed in SEEDS:
            np.random.seed(seed)
            for n in range(2):
                out = g(*call)
                d[f"{case}:{seed}:{n}"] = out.numpy() if hasattr(out, "numpy") else np.asarray(out)
    # the log line of the InternVideo2 single_modality generators (datasets/masking_generator.py:12-16, 38-41; printed by run_pretraining.py)
    spec = importlib.util.spec_from_file_location("_iv_ref_sm_masks", os.path.join(REF, "InternVideo2", "single_modality", "datasets", "masking_generator.py"))
    sm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sm)
    d["sm_tube:repr"] = np.frombuffer(repr(sm.TubeMaskingGenerator((8, 16, 16), 0.8)).encode(), dtype=np.uint8)
    d["sm_random:repr"] = np.frombuffer(repr(sm.RandomMaskingGenerator((8, 16, 16), 0.8)).encode(), dtype=np.uint8)
    path = os.path.join(HERE, "videomae_masks.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, len(d), "arrays")


if __name__ == "__main__":
    main()
