"""Mask generators of the VideoMAE pixel-reconstruction recipe: the `--mask_type` choices of
InternVideo1/Pretrain/VideoMAE/run_mae_pretraining.py:50-55 (`random`, `t_consist`, `t_progressive`, `t_center_prog`, built in
datasets.py:54-66) and the two batched generators of the same file (cell-running encoder masks, random decoder masks).

Host-side index work; same constructor / call signatures, same return dtypes and the same consumption of numpy's GLOBAL RNG as
InternVideo1/Pretrain/VideoMAE/masking_generator.py, so `np.random.seed(s)` reproduces the reference's masks bit for bit
(tests/golden/videomae_masks.npz).  1 = masked.  The progressive generators keep a different number of patches in every FRAME (the same
numbers for every sample, so a batch still has equal visible counts per row, which the encoder's gather needs)."""
from __future__ import annotations

import numpy as np
import torch


def _kth_largest(values: np.ndarray, k: int) -> float:
    """value of the k-th largest element (k >= 1).  The reference gets it from its own top-k helper (masking_generator.py:16-31) and
    fails with an IndexError for k == 0 (a 4x4 grid at the 5 % floor): kept."""
    if k <= 0:
        raise IndexError("keep count 0: the reference's top-k of zero elements has no last element")
    n = values.shape[0]
    return np.partition(values, n - k)[n - k]


class MaskingGenerator:
    def update_state(self, epoch):
        pass

    def __repr__(self):
        """the one-line summary the reference's scripts log for a generator (masking_generator.py:51-54, 113-115, 131-134, 154-157, 182-185,
        220-223): the cell-running generator names its ratio, every other one its patch totals (held to the reference's own strings:
        tests/golden/videomae_masks.npz `<case>:repr`)"""
        if hasattr(self, "all_mask_maps"):
            return f"Cell Running Mask with mask ratio {self.mask_ratio}"
        total = getattr(self, "total_patches", None)
        if total is None:
            total = self.num_patches
        masked = getattr(self, "total_masks", None)
        if masked is None:
            masked = self.num_mask
        return f"Mask: total patches {total}, mask patches {masked}"


class RandomMaskingGenerator(MaskingGenerator):
    """masking_generator.py:40-62: int(ratio * T*H*W) ones shuffled over the whole clip -> float64 (T*H*W,)."""

    def __init__(self, input_size, mask_ratio):
        if not isinstance(input_size, tuple):
            input_size = (input_size,) * 3
        self.frames, self.height, self.width = input_size
        self.num_patches = self.frames * self.height * self.width
        self.num_mask = int(mask_ratio * self.num_patches)

    def __call__(self):
        mask = np.hstack([np.zeros(self.num_patches - self.num_mask), np.ones(self.num_mask)])
        np.random.shuffle(mask)
        return mask


class TemporalConsistencyMaskingGenerator(MaskingGenerator):
    """masking_generator.py:145-166 (`t_consist`): one shuffled frame pattern repeated over the frames (a tube) -> float64 (T*H*W,)."""

    def __init__(self, input_size, mask_ratio):
        self.frames, self.height, self.width = input_size
        self.num_patches_per_frame = self.height * self.width
        self.total_patches = self.frames * self.num_patches_per_frame
        self.num_masks_per_frame = int(mask_ratio * self.num_patches_per_frame)
        self.total_masks = self.frames * self.num_masks_per_frame

    def __call__(self):
        frame = np.hstack([np.zeros(self.num_patches_per_frame - self.num_masks_per_frame), np.ones(self.num_masks_per_frame)])
        np.random.shuffle(frame)
        return np.tile(frame, (self.frames, 1)).flatten()


class _ThresholdedNoiseMask(MaskingGenerator):
    """Shared body of the two progressive generators: ONE normal draw per patch position (np.random.randn(1, H*W)); frame i masks every
    position whose draw is <= the keep_i-th largest draw -- the keep_i-th largest itself is masked too, so frame i shows keep_i - 1 patches
    (masking_generator.py:187-197, 222-232).  Frames that keep fewer patches show a subset of those that keep more.  -> int64 (T*H*W,)."""
    frames: int
    num_patches_per_frame: int
    keep_patches_list = None

    def __call__(self):
        rand = np.random.randn(1, self.num_patches_per_frame)
        mask = np.zeros((self.frames, self.num_patches_per_frame), dtype=bool)
        for i in range(self.frames):
            mask[i] = rand[0] <= _kth_largest(rand[0], int(self.keep_patches_list[i]))
        return mask.flatten().astype(int)


class TemporalProgressiveMaskingGenerator(_ThresholdedNoiseMask):
    """masking_generator.py:169-197 (`t_progressive`): keep counts fall linearly from int((1 - ratio) * H*W) in the first frame to
    int(0.05 * H*W) in the last."""

    def __init__(self, input_size, mask_ratio):
        self.frames, self.height, self.width = input_size
        self.num_patches_per_frame = self.height * self.width
        self.total_patches = self.frames * self.num_patches_per_frame
        hi = int((1 - mask_ratio) * self.num_patches_per_frame)
        lo = int(0.05 * self.num_patches_per_frame)
        self.keep_patches_list = np.linspace(hi, lo, self.frames).astype(int)
        self.total_masks = self.total_patches - self.keep_patches_list.sum()



class TemporalCenteringProgressiveMaskingGenerator(_ThresholdedNoiseMask):
    """masking_generator.py:200-232 (`t_center_prog`): keep counts rise linearly from int(0.05 * H*W) at both ends of the clip to
    int((1 - ratio) * H*W) in its two middle frames (T // 2 values, mirrored)."""

    def __init__(self, input_size, mask_ratio):
        self.num_frames, self.height, self.width = input_size
        self.frames = self.num_frames
        self.num_patches_per_frame = self.height * self.width
        self.total_patches = self.num_frames * self.num_patches_per_frame
        hi = int((1 - mask_ratio) * self.num_patches_per_frame)
        lo = int((1 - 0.95) * self.num_patches_per_frame)
        falling = np.linspace(hi, lo, self.num_frames // 2).astype(int).tolist()
        self.keep_patches_list = falling[::-1] + falling
        self.total_masks = self.total_patches - sum(self.keep_patches_list)



class CellRunningMaskingGenerator(MaskingGenerator):
    """masking_generator.py:65-119: every 2x2 cell of a frame holds int(4 * ratio) masked positions; the pattern inside the cell rotates by one
    position per frame.  The four phases are tabulated once; a call draws one phase per sample (np.random.randint) -> float64 tensor
    (batch, T*H*W)."""

    def __init__(self, input_size, mask_ratio=0.5, is_train=True):
        self.frames, self.height, self.width = input_size
        self.mask_ratio = mask_ratio
        self.ptr_pos = -1 if is_train else 0
        n_masked = int(4 * self.mask_ratio)
        assert 0 < n_masked < 4
        self.cell_size = 4
        queue = np.hstack([np.ones(n_masked), np.zeros(4 - n_masked)])
        maps = []
        for phase in range(self.cell_size):
            frames = []
            for f in range(self.frames):
                unit = queue[(np.arange(4) + phase + f + 1) % 4].reshape(2, 2)          # the pointer advances BEFORE the cell is read
                frames.append(np.tile(unit, [self.height // 2, self.width // 2]))
            maps.append(np.stack(frames, axis=0).flatten())
        self.all_mask_maps = np.stack(maps, axis=0)

    def __call__(self, batch_size):
        phase = np.random.randint(self.cell_size, size=(batch_size))
        return torch.as_tensor(self.all_mask_maps[phase])


class RandomDecodeMaskingGenerator(MaskingGenerator):
    """masking_generator.py:122-142: per sample the int(ratio * T*H*W) positions with the largest of T*H*W normal draws -> float32 tensor
    (batch, T*H*W)."""

    def __init__(self, input_size, mask_ratio=0.5):
        self.frame, self.height, self.width = input_size
        self.mask_ratio = mask_ratio
        self.num_patches = self.frame * self.height * self.width
        self.num_mask = int(mask_ratio * self.num_patches)

    def __call__(self, batch_size):
        rand = torch.as_tensor(np.random.randn(batch_size, self.num_patches))
        idx = torch.topk(rand, self.num_mask, dim=-1, sorted=False).indices
        return torch.zeros(batch_size, self.num_patches).scatter_(-1, idx, 1)


MASK_TYPES = {"random": RandomMaskingGenerator, "t_consist": TemporalConsistencyMaskingGenerator,
              "t_progressive": TemporalProgressiveMaskingGenerator, "t_center_prog": TemporalCenteringProgressiveMaskingGenerator}


def build_mask_generator(mask_type: str, window_size, mask_ratio: float) -> MaskingGenerator:
    """the `--mask_type` switch of datasets.py:54-66 (`DataAugmentationForVideoMAE`); an unknown type is an error here where the reference
    would fail later with a missing attribute."""
    if mask_type not in MASK_TYPES:
        raise ValueError(f"mask_type {mask_type!r}: one of {sorted(MASK_TYPES)}")
    return MASK_TYPES[mask_type](tuple(window_size), mask_ratio)
