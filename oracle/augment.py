"""Oracle (test infrastructure): numpy restatement of csrc/augment.hip -- the GPU stand-in for the reference's imgaug
pipeline (dataset/augment.py:10-77).  Same plan format, same counter-based hash, same fp32 operation order; uint8 in,
uint8 out.  Only tests import this."""
from __future__ import annotations

import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
        return x ^ (x >> np.uint64(31))


def u01(key: np.ndarray) -> np.ndarray:
    return (splitmix64(key) >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def key(seed, slot, channel, index):
    return np.uint64(seed) ^ (np.uint64(slot) << np.uint64(56)) ^ (np.uint64(channel) << np.uint64(52)) ^ index.astype(np.uint64)


def to_u8(v: np.ndarray) -> np.ndarray:
    return np.clip(np.rint(v), 0, 255).astype(np.float32)


def pointwise(img: np.ndarray, plan_i: np.ndarray, seed, first: int, last: int) -> np.ndarray:
    h, w, _ = img.shape
    v = img.astype(np.float32)
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    pix = (yy * w + xx).astype(np.uint64)
    f32 = np.float32
    for s in range(first, last):
        r = plan_i[s]
        code, pc = int(r[0]), r[5] != 0
        for c in range(3):
            kc = c if pc else 0
            o = v[..., c]
            if code == 2:
                u1, u2 = u01(key(seed, s, kc, pix * np.uint64(2))), u01(key(seed, s, kc, pix * np.uint64(2) + np.uint64(1)))
                z = np.sqrt(f32(-2.0) * np.log(u1 + f32(2.98023224e-8))) * np.cos(f32(6.28318530718) * u2)
                o = o + r[1] * z.astype(np.float32)
            elif code == 3:
                gy = np.minimum((yy.astype(np.float32) * r[2] / f32(h)).astype(np.int32), int(r[2]) - 1)
                gx = np.minimum((xx.astype(np.float32) * r[3] / f32(w)).astype(np.int32), int(r[3]) - 1)
                drop = u01(key(seed, s, kc, gy.astype(np.uint64) * np.uint64(65536) + gx.astype(np.uint64))) < r[1]
                o = np.where(drop, f32(0), o)
            elif code == 4:
                o = np.where(u01(key(seed, s, kc, pix)) < r[1], f32(0), o)
            elif code == 5:
                o = o + r[1 + c]
            elif code == 6:
                o = o * r[1 + c]
            elif code == 7:
                o = f32(128.0) + r[1 + c] * (o - f32(128.0))
            v[..., c] = to_u8(o.astype(np.float32))
    return v.astype(np.uint8)


def blur(img: np.ndarray, sigma: float) -> np.ndarray:
    f32 = np.float32
    h, w, _ = img.shape
    d = np.arange(5, dtype=np.float32) - f32(2)
    k = np.exp(-(d * d) / (f32(2.0) * f32(sigma) * f32(sigma))).astype(np.float32)
    ks = f32(0)
    for t in range(5):
        ks = f32(ks + k[t])
    k = (k / ks).astype(np.float32)

    def refl(q, m):
        q = np.abs(q)
        return np.where(q >= m, 2 * m - 2 - q, q)

    src = img.astype(np.float32)
    ys, xs = np.arange(h), np.arange(w)
    acc = np.zeros((h, w, 3), dtype=np.float32)
    for dy in range(5):
        yy = refl(ys + dy - 2, h)
        row = np.zeros((h, w, 3), dtype=np.float32)
        for dx in range(5):
            xx = refl(xs + dx - 2, w)
            row = (row + k[dx] * src[yy][:, xx]).astype(np.float32)
        acc = (acc + k[dy] * row).astype(np.float32)
    return to_u8(acc).astype(np.uint8)


def augment(frames: np.ndarray, plan, seeds, ranges, sigma) -> np.ndarray:
    out = frames.copy()
    for i in range(frames.shape[0]):
        img = pointwise(out[i], plan[i], seeds[i], int(ranges[i, 0]), int(ranges[i, 1]))
        if sigma[i] > 0:
            img = blur(img, float(sigma[i]))
        out[i] = pointwise(img, plan[i], seeds[i], int(ranges[i, 2]), int(ranges[i, 3]))
    return out
