"""Golden fixture for the checkpoint-time positional-table resizers.  Authoring container only.

Loads the REFERENCE's `multi_modality/models/backbones/internvideo2/pos_embed.py` by file path and runs its three loaders on seeded tables:
  * `interpolate_pos_embed` (:137-182; one table named by `pos_name`, frames from `model.T`)            4 frames x 4x4 -> 8 frames x 4x4 (time only)
  * `interpolate_pos_embed_internvideo2_new` (:239-298; every '*pos_embed*' key but 'img_pos_embed')     8 frames x 4x4 -> 4 frames x 6x6
  * `interpolate_pos_embed_internvideo2` (:185-236) on a table without extra (cls) rows                   8 frames x 4x4 -> 8 frames x 5x5 (grid only)

    python tests/golden/make_golden_pos_interp.py      ->  tests/golden/pos_interp.npz     (in:<case>:<key>, out:<case>:<key>)
"""
import importlib.util
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("IV_REFERENCE_ROOT", "/root/reference")


def load_reference():
    path = os.path.join(REF, "InternVideo2", "multi_modality", "models", "backbones", "internvideo2", "pos_embed.py")
    spec = importlib.util.spec_from_file_location("_iv_ref_pos_embed", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def model(frames, grid, extra, D, tubelet=1):
    class _M:
        class patch_embed:
            num_patches = frames * grid * grid
        pos_embed = torch.zeros(1, frames * grid * grid + extra, D)
        T = frames
        num_frames, tubelet_size = frames * tubelet, tubelet
    return _M


CASES = {
    # case: (function, kwargs, model geometry (frames, grid, extra, D[, tubelet]), {key: checkpoint rows})
    "single": ("interpolate_pos_embed", dict(orig_t_size=4, pos_name="vision_encoder.pos_embed"), (8, 4, 1, 24),
               {"vision_encoder.pos_embed": 4 * 16 + 1, "vision_encoder.clip_pos_embed": 4 * 16 + 1}),
    "new": ("interpolate_pos_embed_internvideo2_new", dict(orig_t_size=8), (4, 6, 1, 16, 2),
            {"vision_encoder.pos_embed": 8 * 16 + 1, "clip_pos_embed": 8 * 16 + 1, "vision_encoder.img_pos_embed": 16 + 1}),
    "nocls": ("interpolate_pos_embed_internvideo2", dict(orig_t_size=8), (8, 5, 0, 16), {"pos_embed": 8 * 16}),
}


def main():
    ref = load_reference()
    rng = np.random.Generator(np.random.PCG64(29))
    d = {}
    for case, (fn, kw, geo, keys) in CASES.items():
        D = geo[3]
        ck = {k: torch.from_numpy(rng.standard_normal((1, rows, D)).astype(np.float32)) for k, rows in keys.items()}
        for k, v in ck.items():
            d[f"in:{case}:{k}"] = v.numpy().copy()
        getattr(ref, fn)(ck, model(*geo), **kw)
        for k, v in ck.items():
            d[f"out:{case}:{k}"] = v.numpy().copy()
    path = os.path.join(HERE, "pos_interp.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, {k: v.shape for k, v in d.items() if k.startswith("out:")})


if __name__ == "__main__":
    main()
