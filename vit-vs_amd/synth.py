"""Synthetic servo inputs (no datasets offline): frame pairs, depth maps.

Recipe follows SURVEY.md §8(d): the desired frame is a band-limited texture,
the current frame is the desired frame under a small similarity warp plus
sensor noise, so token correspondences are non-trivial; depth is the
reference's sensor format, uint16 millimetres with 0 = invalid
(reference: realsense_gazebo_plugin/src/RealSensePlugin.cpp:250-262), around
the 0.61 m working distance of the experiments (reference: vitvs_v2.py:1391).
"""
from __future__ import annotations

import numpy as np


def texture(size: int, seed: int, waves: int = 96) -> np.ndarray:
    """Band-limited RGB texture, uint8 [size, size, 3]."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(size, dtype=np.float64), np.arange(size, dtype=np.float64), indexing="ij")
    img = np.zeros((size, size, 3), dtype=np.float64)
    for _ in range(waves):
        wavelength = rng.uniform(6.0, 80.0)
        theta = rng.uniform(0.0, 2.0 * np.pi)
        fx, fy = np.cos(theta) / wavelength, np.sin(theta) / wavelength
        amp = rng.uniform(0.3, 1.0, size=3) * (wavelength / 80.0) ** 0.5
        phase = rng.uniform(0.0, 2.0 * np.pi, size=3)
        arg = 2.0 * np.pi * (fx * xx + fy * yy)
        for c in range(3):
            img[..., c] += amp[c] * np.sin(arg + phase[c])
    img -= img.min(axis=(0, 1), keepdims=True)
    img /= img.max(axis=(0, 1), keepdims=True)
    return np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8)


def _bilinear(img: np.ndarray, xs: np.ndarray, ys: np.ndarray) -> np.ndarray:
    h, w = img.shape[:2]
    xs = np.clip(xs, 0.0, w - 1.0)
    ys = np.clip(ys, 0.0, h - 1.0)
    x0 = np.floor(xs).astype(np.int64)
    y0 = np.floor(ys).astype(np.int64)
    x1 = np.minimum(x0 + 1, w - 1)
    y1 = np.minimum(y0 + 1, h - 1)
    ax = (xs - x0)[..., None]
    ay = (ys - y0)[..., None]
    f = img.astype(np.float64)
    top = f[y0, x0] * (1 - ax) + f[y0, x1] * ax
    bot = f[y1, x0] * (1 - ax) + f[y1, x1] * ax
    return top * (1 - ay) + bot * ay


def warp_similarity(img: np.ndarray, shift=(12.0, -9.0), rot_deg: float = 4.0, scale: float = 1.03,
                    noise_sigma: float = 2.0, seed: int = 0) -> np.ndarray:
    """``img`` seen after a small similarity motion (+ N(0, sigma²) noise), uint8 same shape."""
    h, w = img.shape[:2]
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    th = np.deg2rad(rot_deg)
    c, s = np.cos(th) / scale, np.sin(th) / scale
    dx, dy = xx - cx - shift[0], yy - cy - shift[1]
    xs = c * dx + s * dy + cx
    ys = -s * dx + c * dy + cy
    out = _bilinear(img, xs, ys)
    rng = np.random.default_rng(seed)
    out = out + rng.normal(0.0, noise_sigma, size=out.shape)
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def frame_pair(size: int, seed: int = 20250704):
    """(I_des, I_cur) uint8 [size, size, 3] each; the motion depends on the seed."""
    rng = np.random.default_rng(seed + 7919)
    des = texture(size, seed)
    frac = size / 224.0
    shift = (rng.uniform(8, 24) * rng.choice([-1, 1]) * frac, rng.uniform(8, 24) * rng.choice([-1, 1]) * frac)
    cur = warp_similarity(des, shift=shift, rot_deg=rng.uniform(-10, 10), scale=rng.uniform(0.95, 1.05),
                          noise_sigma=2.0, seed=seed + 1)
    return des, cur


def depth_map(seed: int = 20250704, height: int = 480, width: int = 640, plane_mm: int = 610,
              jitter_mm: int = 50, zero_fraction: float = 0.002) -> np.ndarray:
    """uint16 millimetre depth image; a few zeros exercise the invalid-depth sentinel
    (reference: vitvs_v2.py:582, 0 -> 100 m)."""
    rng = np.random.default_rng(seed + 104729)
    z = plane_mm + rng.integers(-jitter_mm, jitter_mm + 1, size=(height, width))
    holes = rng.random((height, width)) < zero_fraction
    z[holes] = 0
    return z.astype(np.uint16)


def depth_pattern(height: int = 480, width: int = 640) -> np.ndarray:
    """Closed-form uint16 millimetre depth image (no RNG, so fixtures need not store it):
    560..660 mm ripple with a sparse lattice of zeros (invalid pixels)."""
    v, u = np.meshgrid(np.arange(height, dtype=np.int64), np.arange(width, dtype=np.int64), indexing="ij")
    z = 560 + (u * 7 + v * 13) % 101
    z[(u * 3 + v * 5) % 37 == 0] = 0
    return z.astype(np.uint16)


# Frame seeds per BASELINE config (synthetic checkpoint seed 0).  For the first three the
# fixture meets the acceptance rule of SURVEY.md §8(d) (4 <= mutual NN < T, mean(sim_1) <= 0.99,
# top-1/top-2 similarity margins >= 1e-4 — for vits14_308 on the binned descriptors, the
# reference's default), so argmax parity is demanded bit-exact.  With thousands of tokens
# (vitb8_448, vitl14_518) the smallest margin over all rows is ~1e-6 for every seed, so those
# fixtures are marked non-strict and argmax parity is judged tie-tolerantly (tests/).
ACCEPTED_FRAME_SEEDS = {
    "vits16_224": 20250705,
    "vitb16_224": 20250715,
    "vits14_308": 20250738,
    "vitb8_448": 20250705,
    "vitl14_518": 20250705,
}

# BASELINE.json configs[3] (8-camera rig: 8 ViT-B/16 224² pairs, one per GPU): the first 8 frame seeds from the headline
# seed upward that meet the same acceptance rule (tests/golden/rig8_vitb16_224.npz, oracle/make_golden.py rig8).
RIG8_FRAME_SEEDS = (20250715, 20250716, 20250717, 20250726, 20250727, 20250728, 20250730, 20250737)
