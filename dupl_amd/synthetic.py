"""Deterministic, platform-independent synthetic data: counter-based integer hash ->
float64 uniforms -> Box-Muller; synthetic uint8-derived image batches, multi-hot labels, crop boxes and
smooth CAM-like maps.  Pure numpy element-wise arithmetic, so the same call yields the same bits on the
authoring container and on the GPU box (unlike torch.manual_seed streams).  Data only: shared by the
oracle, the tests and bench.py."""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

Tensor = torch.Tensor

IMG_MEAN = (123.675, 116.28, 103.53)
IMG_STD = (58.395, 57.12, 57.375)


# ----------------------------------------------------------------------------------------------
# Deterministic, platform-independent tensor generator (not torch.manual_seed: version dependent)
# ----------------------------------------------------------------------------------------------
def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wraps mod 2**64)."""
    x = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def _name_seed(name: str, seed: int) -> np.uint64:
    h = np.uint64(1469598103934665603)
    with np.errstate(over="ignore"):
        for ch in name.encode():
            h = (h ^ np.uint64(ch)) * np.uint64(1099511628211)
        h = h ^ (np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15))
    return h


def hash_uniform(name: str, n: int, seed: int = 0, stream: int = 0) -> np.ndarray:
    """n float64 uniforms in (0,1) from a counter-based hash keyed on (name, seed, stream)."""
    base = _name_seed(name, seed)
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(2) + np.uint64(stream) + base
    bits = _mix64(_mix64(ctr) + np.uint64(0x9E3779B97F4A7C15))
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)


def hash_normal(name: str, shape: Sequence[int], std: float = 1.0, seed: int = 0) -> Tensor:
    """float32 N(0,std) tensor: Box-Muller in float64 over hash_uniform streams 0/1."""
    n = int(np.prod(shape)) if len(shape) else 1
    u1 = hash_uniform(name, n, seed, 0)
    u2 = hash_uniform(name, n, seed, 1)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy((z * std).astype(np.float32).reshape(tuple(shape)))


def hash_randint(name: str, shape: Sequence[int], lo: int, hi: int, seed: int = 0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = hash_uniform(name, n, seed, 0)
    return (lo + np.floor(u * (hi - lo))).astype(np.int64).reshape(tuple(shape))


# ----------------------------------------------------------------------------------------------
# Synthetic batch
# ----------------------------------------------------------------------------------------------
def synthetic_batch(b: int, num_fg: int = 20, size: int = 448, seed: int = 0, smooth: bool = True):
    """(inputs fp32 normalised from a uint8 image, cls_label multi-hot 1-3 positives, img_box i16).
    `smooth`: blocky low-frequency content (so PAR affinities are non-degenerate) + uint8 noise.
    Pure numpy float64 element-wise arithmetic -> bit-identical on every platform."""
    u = hash_uniform(f"img{b}x{size}", b * 3 * size * size, seed).reshape(b, 3, size, size)
    if smooth:
        cell = 28 if size % 28 == 0 else 16
        g = -(-size // cell)
        lo = hash_uniform(f"imglo{b}x{size}", b * 3 * g * g, seed).reshape(b, 3, g, g)
        lo = np.repeat(np.repeat(lo, cell, axis=2), cell, axis=3)[:, :, :size, :size]
        yy = (np.arange(size, dtype=np.float64) / size).reshape(1, 1, size, 1)
        xx = (np.arange(size, dtype=np.float64) / size).reshape(1, 1, 1, size)
        ramp = 0.5 * yy + 0.5 * xx
        u = 0.55 * lo + 0.25 * ramp + 0.2 * u
    img_u8 = np.clip(np.floor(u * 256.0), 0, 255).astype(np.float32)
    mean = np.array(IMG_MEAN, dtype=np.float32).reshape(1, 3, 1, 1)
    std = np.array(IMG_STD, dtype=np.float32).reshape(1, 3, 1, 1)
    inputs = torch.from_numpy(((img_u8 - mean) / std).astype(np.float32))
    cls = np.zeros((b, num_fg), dtype=np.float32)
    npos = hash_uniform(f"npos{b}", b, seed)
    picks = hash_randint(f"pick{b}", (b, 3), 0, num_fg, seed)
    for i in range(b):
        k = 1 if npos[i] < 0.6 else (2 if npos[i] < 0.9 else 3)
        for j in range(k):
            cls[i, picks[i, j]] = 1.0
    box = np.zeros((b, 4), dtype=np.int16)
    r = hash_randint(f"box{b}", (b, 4), 0, size // 2, seed)
    for i in range(b):
        if i % 2 == 0:
            box[i] = [0, size, 0, size]
        else:
            y0, x0 = int(r[i, 0]) // 2, int(r[i, 1]) // 2
            y1 = min(size, y0 + size // 2 + int(r[i, 2]) // 2)
            x1 = min(size, x0 + size // 2 + int(r[i, 3]) // 2)
            box[i] = [y0, y1, x0, x1]
    return inputs, torch.from_numpy(cls), torch.from_numpy(box)


def synthetic_cams(b: int, C: int, h: int, w: int, seed: int = 0) -> Tensor:
    """Smooth synthetic CAMs in [0,1]: a sum of 4 hash-placed Gaussian bumps per (image, class),
    min-max normalised like a10.  numpy float64 -> float32 (platform independent up to libm exp)."""
    prm = hash_uniform(f"cams{b}x{C}x{h}", b * C * 4 * 4, seed).reshape(b, C, 4, 4)
    yy = (np.arange(h, dtype=np.float64) / h).reshape(1, 1, h, 1)
    xx = (np.arange(w, dtype=np.float64) / w).reshape(1, 1, 1, w)
    cam = np.zeros((b, C, h, w), dtype=np.float64)
    for k in range(4):
        cy = prm[:, :, k, 0].reshape(b, C, 1, 1)
        cx = prm[:, :, k, 1].reshape(b, C, 1, 1)
        sg = 0.05 + 0.25 * prm[:, :, k, 2].reshape(b, C, 1, 1)
        am = 0.3 + prm[:, :, k, 3].reshape(b, C, 1, 1)
        cam += am * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * sg * sg))
    cam = cam - cam.min(axis=(2, 3), keepdims=True)
    cam = cam / (cam.max(axis=(2, 3), keepdims=True) + 1e-5)
    return torch.from_numpy(cam.astype(np.float32))
