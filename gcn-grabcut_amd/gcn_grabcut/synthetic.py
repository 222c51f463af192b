"""
Seeded synthetic DUTS-shaped images for benchmarks and parity tests
(the reference's own generator is dataset.py:667-749).

A low-frequency colour background, 1-3 filled ellipses covering roughly 5-25 %
of the frame (DUTS mean foreground is 11 %), and additive integer noise.
Pure numpy so that it also runs under the fixture-generation interpreter.
"""
from __future__ import annotations

import numpy as np


def synthetic_image(height: int, width: int, seed: int, return_mask: bool = False):
    """BGR uint8 (H, W, 3) [and the ground-truth foreground mask uint8 (H, W)]."""
    rng = np.random.default_rng(seed)
    gh, gw = max(2, height // 40 + 2), max(2, width // 40 + 2)
    grid = rng.integers(20, 221, size=(gh, gw, 3)).astype(np.float64)
    ys = np.linspace(0.0, gh - 1.0, height)
    xs = np.linspace(0.0, gw - 1.0, width)
    y0 = np.minimum(ys.astype(np.int64), gh - 2)
    x0 = np.minimum(xs.astype(np.int64), gw - 2)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    g00 = grid[y0][:, x0]
    g01 = grid[y0][:, x0 + 1]
    g10 = grid[y0 + 1][:, x0]
    g11 = grid[y0 + 1][:, x0 + 1]
    img = (g00 * (1 - fy) * (1 - fx) + g01 * (1 - fy) * fx + g10 * fy * (1 - fx) + g11 * fy * fx)

    mask = np.zeros((height, width), dtype=bool)
    yy, xx = np.mgrid[0:height, 0:width]
    for _ in range(int(rng.integers(1, 4))):
        cy = rng.uniform(0.25, 0.75) * height
        cx = rng.uniform(0.25, 0.75) * width
        ry = rng.uniform(0.08, 0.22) * height
        rx = rng.uniform(0.08, 0.22) * width
        th = rng.uniform(0.0, np.pi)
        dy, dx = yy - cy, xx - cx
        u = (dx * np.cos(th) + dy * np.sin(th)) / rx
        v = (-dx * np.sin(th) + dy * np.cos(th)) / ry
        inside = (u * u + v * v) <= 1.0
        colour = rng.integers(120, 241, size=3).astype(np.float64)
        shade = 1.0 - 0.25 * np.clip(u * 0.5 + v * 0.5, -1.0, 1.0)
        img[inside] = (colour[None, :] * shade[inside][:, None])
        mask |= inside
    img += rng.integers(-10, 11, size=img.shape)
    out = np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))
    if return_mask:
        return out, mask.astype(np.uint8)
    return out


def synthetic_batch(n: int, height: int, width: int, config_id: int = 3, first_index: int = 0) -> np.ndarray:
    """(n, H, W, 3) uint8; image i uses seed 10_000 * config_id + first_index + i."""
    return np.ascontiguousarray(
        np.stack([synthetic_image(height, width, 10_000 * config_id + first_index + i) for i in range(n)]))
