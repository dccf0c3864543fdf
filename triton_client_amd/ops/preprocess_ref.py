"""Host-side (numpy / Pillow) pieces of the image preprocessing pipeline: the
resize coefficient tables the HIP kernel consumes, the scaling-mode table and a
pure CPU rendition of the whole pipeline that serves as the test oracle and as
the fallback for CPU-only servers.  No GPU library is touched here."""

from __future__ import annotations

import math
from functools import lru_cache

import numpy as np

PRECISION_BITS = 22
TILE_W = 64
_LDS_BUDGET = 48 * 1024


# ---------------------------------------------------------------------------
# coefficient tables (host)
# ---------------------------------------------------------------------------
@lru_cache(maxsize=256)
def pil_bilinear_coeffs(in_size, out_size):
    """Windows and fixed-point weights of Pillow's BILINEAR resampling.

    Returns ``(bounds[out,2] int32 (first, count), coeffs[out,ksize] int32, ksize)``.
    """
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coeffs = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = np.empty(xmax, dtype=np.float64)
        for x in range(xmax):
            t = (x + xmin - center + 0.5) * ss
            t = -t if t < 0 else t
            k[x] = 1.0 - t if t < 1.0 else 0.0
        ww = k.sum()
        if ww != 0.0:
            k = k / ww
        bounds[xx] = (xmin, xmax)
        # Pillow: (int)(k * (1 << PRECISION_BITS) + 0.5) for k >= 0 (triangle weights are never negative)
        coeffs[xx, :xmax] = (k * (1 << PRECISION_BITS) + 0.5).astype(np.int64).astype(np.int32)
    return bounds, coeffs, ksize


@lru_cache(maxsize=256)
def cv_bilinear_coeffs(in_size, out_size):
    """2-tap half-pixel-centre bilinear windows (OpenCV INTER_LINEAR geometry)."""
    scale = in_size / out_size
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coeffs = np.zeros((out_size, 2), dtype=np.int32)
    one = 1 << PRECISION_BITS
    for xx in range(out_size):
        fx = (xx + 0.5) * scale - 0.5
        sx = int(math.floor(fx))
        fx -= sx
        if sx < 0:
            sx, fx = 0, 0.0
        if sx >= in_size - 1:
            sx, fx = max(in_size - 2, 0), 1.0 if in_size > 1 else 0.0
        if in_size == 1:
            bounds[xx] = (0, 1)
            coeffs[xx, 0] = one
            continue
        w1 = int(fx * one + 0.5)
        bounds[xx] = (sx, 2)
        coeffs[xx] = (one - w1, w1)
    return bounds, coeffs, 2


def _tables(mode, in_size, out_size):
    return pil_bilinear_coeffs(in_size, out_size) if mode == "pil" else cv_bilinear_coeffs(in_size, out_size)


def scaling_to_affine(scaling, channels):
    """``(scale[4], bias[4])`` such that ``out = pixel * scale + bias``."""
    scale = [1.0] * 4
    bias = [0.0] * 4
    if scaling is None or scaling == "NONE":
        pass
    elif scaling == "INCEPTION":
        scale = [1.0 / 127.5] * 4
        bias = [-1.0] * 4
    elif scaling == "VGG":
        means = (128.0,) if channels == 1 else (123.0, 117.0, 104.0)
        for c, m in enumerate(means):
            bias[c] = -m
    elif isinstance(scaling, (tuple, list)) and len(scaling) == 2:
        mean, std = scaling  # in [0,1] units (torchvision convention)
        for c in range(channels):
            m = mean[c] if hasattr(mean, "__len__") else mean
            s = std[c] if hasattr(std, "__len__") else std
            scale[c] = 1.0 / (255.0 * s)
            bias[c] = -m / s
    else:
        raise ValueError(f"unknown scaling mode {scaling!r}")
    return scale, bias


def reference_preprocess(image, out_h, out_w, out_c=3, layout="NCHW", np_dtype=np.float32, scaling="NONE",
                         resize="pil", bgr=False):
    """Pure numpy/Pillow rendition of the same pipeline (the test oracle).

    ``resize="pil"`` uses Pillow itself; ``"cv"`` evaluates the 2-tap tables in
    numpy.
    """
    from PIL import Image

    arr = np.asarray(image)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    if arr.shape[2] >= 3 and bgr:
        arr = arr[:, :, [2, 1, 0] + list(range(3, arr.shape[2]))]
    if arr.shape[2] == 1:
        pil = Image.fromarray(arr[:, :, 0], "L")
    else:
        pil = Image.fromarray(np.ascontiguousarray(arr[:, :, :3]), "RGB")
    pil = pil.convert("L" if out_c == 1 else "RGB")
    if resize == "pil":
        resized = np.asarray(pil.resize((out_w, out_h), Image.BILINEAR))
    else:
        src = np.asarray(pil).astype(np.int64)
        if src.ndim == 2:
            src = src[:, :, None]
        xb, xk, _ = cv_bilinear_coeffs(src.shape[1], out_w)
        yb, yk, _ = cv_bilinear_coeffs(src.shape[0], out_h)
        half = 1 << (PRECISION_BITS - 1)
        tmp = np.zeros((src.shape[0], out_w, src.shape[2]), dtype=np.int64)
        for x in range(out_w):
            acc = half
            for t in range(xb[x, 1]):
                acc = acc + src[:, xb[x, 0] + t, :] * int(xk[x, t])
            tmp[:, x, :] = np.clip(acc >> PRECISION_BITS, 0, 255)
        resized = np.zeros((out_h, out_w, src.shape[2]), dtype=np.int64)
        for y in range(out_h):
            acc = half
            for t in range(yb[y, 1]):
                acc = acc + tmp[yb[y, 0] + t, :, :] * int(yk[y, t])
            resized[y] = np.clip(acc >> PRECISION_BITS, 0, 255)
        resized = resized.astype(np.uint8)
    if resized.ndim == 2:
        resized = resized[:, :, None]
    scale, bias = scaling_to_affine(scaling, out_c)
    val = resized.astype(np.float32) * np.array(scale[:out_c], dtype=np.float32) + \
        np.array(bias[:out_c], dtype=np.float32)
    if np.issubdtype(np.dtype(np_dtype), np.integer):
        info = np.iinfo(np_dtype)
        val = np.clip(np.rint(val), info.min, info.max)
    out = val.astype(np_dtype)
    if layout == "NCHW":
        out = np.transpose(out, (2, 0, 1))
    return out


