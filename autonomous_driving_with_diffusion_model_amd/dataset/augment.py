"""Image augmentation for training, on the GPU: stand-in for the reference's imgaug pipeline.

Reference: `dataset/augment.py:10-77` builds, per sample, `iaa.Sequential([...], random_order=True)` of seven operators --
GaussianBlur, AdditiveGaussianNoise, CoarseDropout, Dropout, Add, Multiply, LinearContrast -- each wrapped in
`iaa.Sometimes(frequency_factor, ...)`, with strengths that grow with the number of images seen
(`image_iteration`, counted by `TrajDataset.count_access`, dataset/carla_dataset.py:24-31).  imgaug is not in this image
and a CPU pipeline per sample is the wrong place for it on an MI355X box anyway: here the HOST only draws the plan (which
operators fire, in which order, with which parameters) and the batch of uint8 frames is transformed where it already
lives (`adx_image_augment`, csrc/augment.hip).  `augment_factors` restates the reference's schedule exactly; the operators
follow imgaug's uint8 semantics (round, saturate after every operator) but the per-pixel random numbers come from a
counter-based hash, not from imgaug's generator: same distribution family, not the same samples ("stand-in").
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .. import _lib as L

BLUR, NOISE, COARSE, DROPOUT, ADD, MULTIPLY, CONTRAST = 1, 2, 3, 4, 5, 6, 7
N_SLOTS = 7


def augment_factors(image_iteration: float) -> Dict[str, float]:
    """dataset/augment.py:10-27, line for line (including `blur_factor`, whose min() pins it at 0.5)."""
    iteration = image_iteration / 32
    return {
        "frequency": min(0.05 + float(iteration) / 200000.0, 0.5),
        "color": min(float(iteration) / 1000000.0, 0.5),
        "dropout": 0.198667 + (0.03856658 - 0.198667) / (1 + (iteration / 196416.6) ** 1.863486),
        "blur": min(0.5 + (0.5 * iteration / 100000.0), 0.5),
        "add": 10 + 10 * iteration / 100000.0,
        "multiply_pos": 1 + (2.5 * iteration / 200000.0),
        "multiply_neg": 1 - (0.91 * iteration / 500000.0),
        "contrast_pos": 1 + (0.5 * iteration / 500000.0),
        "contrast_neg": 1 - (0.5 * iteration / 500000.0),
    }


def sample_plan(image_iterations, height: int, width: int, rng: np.random.Generator):
    """One plan per image.  Returns (plan float32 [n, 7, 8], seeds uint64 [n], ranges int32 [n, 4], blur_sigma float32 [n]).

    Per image: a random order of the seven operators (`random_order=True`), each kept with probability `frequency`
    (`iaa.Sometimes`), parameters drawn uniformly from the reference's intervals, `per_channel` with probability `color`.
    ranges[i] = (first, last) slots applied before the image's blur and (first, last) after it."""
    its = np.atleast_1d(np.asarray(image_iterations, dtype=np.float64))
    n = its.shape[0]
    plan = np.zeros((n, N_SLOTS, 8), dtype=np.float32)
    ranges = np.zeros((n, 4), dtype=np.int32)
    sigma = np.zeros((n,), dtype=np.float32)
    seeds = rng.integers(0, 2 ** 63 - 1, size=n, dtype=np.uint64) & np.uint64((1 << 52) - 1)   # the key's upper bits carry slot / channel
    for i in range(n):
        f = augment_factors(float(its[i]))
        order = rng.permutation(np.arange(1, N_SLOTS + 1))
        blur_at = N_SLOTS
        for s, code in enumerate(order):
            if rng.random() >= f["frequency"]:
                continue                                   # Sometimes(frequency, ...): operator skipped, slot stays 0
            pc = float(rng.random() < f["color"])
            row = plan[i, s]
            row[0], row[5] = code, pc
            if code == BLUR:
                row[1] = rng.uniform(0.0, f["blur"])
                if row[1] > 1e-3:
                    sigma[i], blur_at = row[1], s
                else:
                    row[0] = 0                             # imgaug skips the filter below sigma ~ 0
            elif code == NOISE:
                row[1] = rng.uniform(0.0, f["dropout"])    # scale=(0.0, dropout_factor): upstream reuses that factor
            elif code == COARSE:
                row[1] = rng.uniform(0.0, f["dropout"])
                sp = rng.uniform(0.08, 0.2)
                row[2], row[3] = max(1, int(round(height * sp))), max(1, int(round(width * sp)))
            elif code == DROPOUT:
                row[1] = rng.uniform(0.0, f["dropout"])
            else:
                lo, hi = {ADD: (-f["add"], f["add"]), MULTIPLY: (f["multiply_neg"], f["multiply_pos"]),
                          CONTRAST: (f["contrast_neg"], f["contrast_pos"])}[int(code)]
                vals = rng.uniform(lo, hi, size=3) if pc else np.repeat(rng.uniform(lo, hi), 3)
                if code == ADD:
                    vals = np.round(vals)                  # iaa.Add on uint8 adds integers
                row[1:4] = vals
        ranges[i] = (0, blur_at, min(blur_at + 1, N_SLOTS), N_SLOTS)
    return plan, seeds, ranges, sigma


class GpuAugmentor:
    """`augmentor(frames_u8 [N, H, W, 3] on the GPU, first_image_iteration)` -> the augmented batch (a new tensor).
    Sample i of the batch is image number `first_image_iteration + i` of the run, like `TrajDataset.count_access`."""

    def __init__(self, seed: int = 0):
        self.rng = np.random.default_rng(seed)
        self._scratch: Optional[torch.Tensor] = None

    def __call__(self, frames_u8: torch.Tensor, first_image_iteration: int, plan=None) -> torch.Tensor:
        if not frames_u8.is_cuda or frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[-1] != 3:
            raise L.AdxError("GpuAugmentor expects a uint8 [N, H, W, 3] tensor on the GPU (there is no CPU path)")
        out = frames_u8.contiguous().clone()
        n, h, w, _ = out.shape
        if plan is None:
            plan = sample_plan(first_image_iteration + 1 + np.arange(n), h, w, self.rng)
        p, seeds, ranges, sigma = plan
        dev = out.device
        any_blur = bool((sigma > 0).any())
        if any_blur and (self._scratch is None or self._scratch.numel() < out.numel() or self._scratch.device != dev):
            self._scratch = torch.empty(out.numel(), dtype=torch.uint8, device=dev)
        pd = torch.from_numpy(np.ascontiguousarray(p)).to(dev)
        sd = torch.from_numpy(seeds.view(np.int64).copy()).to(dev)
        rd = torch.from_numpy(np.ascontiguousarray(ranges)).to(dev)
        gd = torch.from_numpy(np.ascontiguousarray(sigma)).to(dev)
        L.check(L.lib().adx_image_augment(out.data_ptr(), L.ptr(self._scratch) if any_blur else None, n, h, w, pd.data_ptr(),
                                          sd.data_ptr(), rd.data_ptr(), gd.data_ptr(), int(any_blur),
                                          L.stream_ptr(dev)), "adx_image_augment")
        return out
