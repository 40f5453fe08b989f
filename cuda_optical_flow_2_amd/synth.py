"""Seeded synthetic frame pairs.

A low-resolution LCG noise grid (cell = 8 px) is bilinearly interpolated into a
smooth texture in [0, 255]; ``next`` is the same texture translated by
``(dx, dy)``:  next(x, y) = tex(x - dx, y - dy).  Pure numpy, deterministic
across machines (integer LCG + float64 interpolation + floor).
"""
from __future__ import annotations

import numpy as np

_A = np.uint32(1664525)
_C = np.uint32(1013904223)


def lcg_stream(n: int, seed: int) -> np.ndarray:
    """n successive values of s = s*1664525 + 1013904223 (mod 2^32), first value is after one step."""
    with np.errstate(over="ignore"):
        apow = np.cumprod(np.full(n, _A, dtype=np.uint32), dtype=np.uint32)          # a^1 .. a^n
        geo = np.concatenate(([np.uint32(1)], apow[:-1])).cumsum(dtype=np.uint32)    # 1 + a + .. + a^(k-1)
        return (apow * np.uint32(seed) + _C * geo).astype(np.uint32)


def _texture(w: int, h: int, seed: int, cell: int, ox: float, oy: float) -> np.ndarray:
    gw, gh = (w + 4 * cell) // cell + 3, (h + 4 * cell) // cell + 3
    grid = (lcg_stream(gw * gh, seed) >> np.uint32(8)).astype(np.float64).reshape(gh, gw) / float(1 << 24)
    xs = (np.arange(w, dtype=np.float64) + 2 * cell - ox) / cell
    ys = (np.arange(h, dtype=np.float64) + 2 * cell - oy) / cell
    x0 = np.clip(np.floor(xs).astype(np.int64), 0, gw - 2)
    y0 = np.clip(np.floor(ys).astype(np.int64), 0, gh - 2)
    fx = (xs - x0)[None, :]
    fy = (ys - y0)[:, None]
    g00 = grid[y0[:, None], x0[None, :]]
    g01 = grid[y0[:, None], x0[None, :] + 1]
    g10 = grid[y0[:, None] + 1, x0[None, :]]
    g11 = grid[y0[:, None] + 1, x0[None, :] + 1]
    t = (g00 * (1 - fx) + g01 * fx) * (1 - fy) + (g10 * (1 - fx) + g11 * fx) * fy
    return np.clip(np.floor(t * 255.0), 0, 255).astype(np.uint8)


def smooth_pair(w: int, h: int, dx: float = 2.0, dy: float = 1.0, seed: int = 12345, cell: int = 8):
    """(prev, next) 1-channel u8 planes of shape (h, w)."""
    return _texture(w, h, seed, cell, 0.0, 0.0), _texture(w, h, seed, cell, dx, dy)


def random_pair(w: int, h: int, seed: int = 1):
    """Uniform-random u8 pair: worst case for value ranges."""
    v = (lcg_stream(2 * w * h, seed) >> np.uint32(24)).astype(np.uint8)
    return v[: w * h].reshape(h, w).copy(), v[w * h:].reshape(h, w).copy()


def to_3ch(plane: np.ndarray) -> np.ndarray:
    """Replicate a 1-channel plane into the reference's interleaved 3-channel layout."""
    return np.repeat(plane[:, :, None], 3, axis=2).copy()
