"""CPU restatement of the frame resize in front of the hot path — TEST INFRASTRUCTURE ONLY (imported by tests/,
__graft_entry__.smoke() and nothing under vit-vs_amd/).

Reference call site: ``goal_image.resize((S, S))`` / ``latest_pil_image.resize((S, S))``
(/root/reference/catkin_ws/ibvs/src/vitvs_v2.py:474-475), i.e. PIL's default filter (BICUBIC for RGB images).  The
arithmetic lives in Pillow, a third-party dependency that is not under /root/reference and is unpinned there
(requirements.txt lists no Pillow version).  This file restates the published algorithm of Pillow's
``src/libImaging/Resample.c`` for 8-bit channels:

  * support = 2 * max(in/out, 1): the bicubic kernel (a = -0.5) is stretched when shrinking (antialiasing);
  * per output sample the taps' weights are normalised in double precision and rounded to 22-bit fixed point
    (``normalize_coeffs_8bpc``: ``(int)(±0.5 + w * 2^22)``);
  * horizontal pass into an intermediate uint8 image: ``clip8((2^21 + sum(px * k)) >> 22)``, then the vertical pass.

Pinned: tests/test_resize.py requires bit-identity with ``PIL.Image.resize`` (Pillow 12.2.0, the version in this image) on
random and structured images for several geometries, so the restatement — and through it the HIP kernel — is anchored on
the dependency itself.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def coefficients(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc -> (bounds [out, 2] (first tap, taps), coeffs [out, ksize] int)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    coeffs = np.zeros((out_size, ksize), dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            coeffs[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, coeffs


def _pass(img: np.ndarray, bounds: np.ndarray, coeffs: np.ndarray) -> np.ndarray:
    """Filter axis 1 of img [rows, in, ch] uint8 -> [rows, out, ch] uint8."""
    rows, _, ch = img.shape
    out = np.empty((rows, bounds.shape[0], ch), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx, (xmin, xcnt) in enumerate(bounds):
        acc = (src[:, xmin:xmin + xcnt, :] * coeffs[xx, :xcnt, None]).sum(axis=1) + (1 << (PRECISION_BITS - 1))
        out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bicubic_u8(img: np.ndarray, out_size: int) -> np.ndarray:
    """img uint8 [H, W, 3] -> uint8 [out_size, out_size, 3], as PIL.Image.fromarray(img).resize((out_size, out_size))."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, _ = img.shape
    xb, xk = coefficients(w, out_size)
    yb, yk = coefficients(h, out_size)
    tmp = _pass(img, xb, xk)                                            # horizontal: [H, out, 3]
    return _pass(tmp.transpose(1, 0, 2), yb, yk).transpose(1, 0, 2)     # vertical on the transposed image
