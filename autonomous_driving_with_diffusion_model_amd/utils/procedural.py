"""Closed-form (seeded, platform-independent) weights and synthetic inputs.

ImageNet/pretrained checkpoints are not available offline, so every parity test,
golden fixture and benchmark uses weights produced here.  Values depend only on
(tensor name, shape, seed) through numpy's PCG64 stream, so the container that
generates golden fixtures (where the reference is importable) and the GPU box
(where it is not) see bit-identical tensors.

Synthetic input distributions follow SURVEY.md §8(d) / BASELINE.md §3.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np
import torch


def _rng(name: str, seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([zlib.crc32(name.encode()), seed & 0xFFFFFFFF]))


def _uniform(name: str, seed: int, shape, lo: float, hi: float) -> torch.Tensor:
    a = _rng(name, seed).random(size=tuple(shape), dtype=np.float64)
    return torch.from_numpy((lo + (hi - lo) * a).astype(np.float32))


def procedural_tensor(name: str, shape: Tuple[int, ...], seed: int = 0) -> torch.Tensor:
    """One parameter/buffer, chosen so that activations stay O(1) through the net."""
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_mean":
        return _uniform(name, seed, shape, -0.1, 0.1)
    if leaf == "running_var":
        return _uniform(name, seed, shape, 0.8, 1.2)
    if len(shape) == 1:
        # norm scales sit near 1, every other vector (biases) near 0
        is_scale = leaf == "weight"
        if is_scale:
            return _uniform(name, seed, shape, 0.9, 1.1)
        return _uniform(name, seed, shape, -0.1, 0.1)
    if leaf == "in_proj_weight" or len(shape) >= 2:
        if ".conv.weight" in name and len(shape) == 3 and "ups." in name:
            # ConvTranspose1d weight is [Cin, Cout, k]; fan_in = Cin * k / stride
            fan_in = shape[0] * shape[2] // 2
        else:
            fan_in = int(np.prod(shape[1:]))
        bound = float(np.sqrt(3.0 / max(fan_in, 1)))
        return _uniform(name, seed, shape, -bound, bound)
    return _uniform(name, seed, shape, -0.1, 0.1)


def procedural_state_dict(spec: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0) -> Dict[str, torch.Tensor]:
    """spec: iterable of (key, shape), e.g. ((k, v.shape) for k, v in model.state_dict().items())."""
    return {k: procedural_tensor(k, tuple(s), seed) for k, s in spec}


def load_procedural(model: torch.nn.Module, seed: int = 0) -> Dict[str, torch.Tensor]:
    sd = procedural_state_dict(((k, tuple(v.shape)) for k, v in model.state_dict().items()), seed)
    model.load_state_dict(sd)
    return sd


def synthetic_batch(batch: int, horizon: int, transition_dim: int = 7, image_hw=(256, 900), seed: int = 0,
                    n_train: int = 100) -> Dict[str, torch.Tensor]:
    """imgs~N(0,1), trajs~U(-1,1) with [:,0,:3]=0, target~U(-1,1), t~U{0..n_train-1}, init/noise~N(0,1)."""
    def normal(name, shape):
        return torch.from_numpy(_rng(name, seed).standard_normal(size=shape, dtype=np.float32))

    h, w = image_hw
    trajs = _uniform("trajs", seed, (batch, horizon, transition_dim), -1.0, 1.0)
    trajs[:, 0, :3] = 0
    return {
        "imgs": normal("imgs", (batch, 3, h, w)),
        "trajs": trajs,
        "target": _uniform("target", seed, (batch, 2), -1.0, 1.0),
        "t": torch.from_numpy(_rng("t", seed).integers(0, n_train, size=(batch,), dtype=np.int64)),
        "init_trajs": normal("init_trajs", (batch, horizon, transition_dim)),
        "noise": normal("noise", (batch, horizon, transition_dim)),
    }


def step_noise(step: int, shape, seed: int = 0) -> torch.Tensor:
    """Injected per-step Gaussian noise for DDPM parity runs (the reference draws it on-device)."""
    return torch.from_numpy(_rng(f"step_noise.{step}", seed).standard_normal(size=tuple(shape), dtype=np.float32))


def control_inputs(tick: int, horizon: int = 16):
    """Waypoints [horizon, 2] (a gently curving path ahead of the ego vehicle, metres), speed [1] and target [2] for
    tick `tick` of the controller parity run (tests/golden/control.npz)."""
    r = _rng("control", tick)
    step = 0.2 + 1.2 * r.random()                       # mean segment length: both sides of BRAKE_SPEED
    curve = (r.random() - 0.5) * 0.12
    s = np.arange(1, horizon + 1, dtype=np.float64) * step
    x = curve * s * s + (r.random(horizon) - 0.5) * 0.05
    y = s + (r.random(horizon) - 0.5) * 0.05
    wp = torch.from_numpy(np.stack([x, y], 1).astype(np.float32))
    vel = torch.tensor([float(3.0 * r.random())], dtype=torch.float32)
    tgt = torch.from_numpy(np.array([(r.random() - 0.5) * 8.0, 2.0 + 14.0 * r.random()], dtype=np.float32))
    return wp, vel, tgt
