"""CPU oracle for the frame pre-processing.  TEST INFRASTRUCTURE ONLY.

The reference pre-processes every camera image on the CPU loader
(datasets/ego4view_syn/ego4view_syn_pose3d.py:41-44,159-162; identical in the other dataset classes):

    PIL.Image.open(path).convert("RGB").resize([256, 256], Image.BICUBIC)  -> uint8 (256,256,3)
    transforms.ToTensor()                                                    -> float32 CHW, x / 255
    transforms.Normalize((0.485,0.456,0.406), (0.229,0.224,0.225))          -> (x - mean) / std

`Image.resize` is Pillow's ImagingResample (third-party, absent from /root/reference; the container has
Pillow 12.2.0; the algorithm — src/libImaging/Resample.c — has been stable since Pillow 3.4): antialiased
separable resampling, horizontal pass then vertical pass, each pass with per-output-pixel windows
[xmin, xmin+xmax), double-precision bicubic (a = -0.5) weights normalised to sum 1, converted to 22-bit fixed point
(round half away from zero), accumulated in int32 from 2^21, shifted right by 22 and clipped to [0,255].
This file restates that algorithm in numpy; tests pin it against Pillow itself (golden vectors in tests/golden and,
where Pillow is importable, live).
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def _bicubic(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0
    if x < 2.0:
        return (((x - 5.0) * x + 8.0) * x - 4.0) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for box = (0, in_size).
    Returns (bounds int32 (out,2) [xmin, count], coeffs int32 (out, ksize), ksize)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
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
        k = np.array([_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)], dtype=np.float64)
        ww = 0.0
        for v in k:          # Pillow accumulates in this order
            ww += v
        if ww != 0.0:
            k = k / ww
        # (int)(+-0.5 + k * 2^22): C casts truncate toward zero
        fixed = np.trunc(np.where(k < 0, -0.5 + k * (1 << PRECISION_BITS), 0.5 + k * (1 << PRECISION_BITS))).astype(np.int32)
        coeffs[xx, :xmax] = fixed
        bounds[xx] = (xmin, xmax)
    return bounds, coeffs, ksize


def _pass(img: np.ndarray, bounds: np.ndarray, coeffs: np.ndarray, axis: int) -> np.ndarray:
    """One 8-bit resampling pass along `axis` (0 = vertical, 1 = horizontal) of an (H, W, C) uint8 image."""
    src = img.astype(np.int64)
    out_n = bounds.shape[0]
    shape = list(img.shape)
    shape[axis] = out_n
    out = np.empty(shape, dtype=np.uint8)
    for o in range(out_n):
        lo, cnt = int(bounds[o, 0]), int(bounds[o, 1])
        k = coeffs[o, :cnt].astype(np.int64)
        if axis == 1:
            acc = (src[:, lo:lo + cnt, :] * k[None, :, None]).sum(axis=1) + (1 << (PRECISION_BITS - 1))
            out[:, o, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
        else:
            acc = (src[lo:lo + cnt, :, :] * k[:, None, None]).sum(axis=0) + (1 << (PRECISION_BITS - 1))
            out[o, :, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_bicubic_resize_u8(img: np.ndarray, out_h: int = 256, out_w: int = 256) -> np.ndarray:
    """Image.resize([out_w, out_h], Image.BICUBIC) on an (H, W, 3) uint8 array: horizontal pass, then vertical."""
    assert img.dtype == np.uint8 and img.ndim == 3
    bh, ch, _ = precompute_coeffs(img.shape[1], out_w)
    bv, cv, _ = precompute_coeffs(img.shape[0], out_h)
    tmp = _pass(img, bh, ch, axis=1)
    return _pass(tmp, bv, cv, axis=0)


def to_tensor_normalize(img_u8: np.ndarray) -> np.ndarray:
    """ToTensor + Normalize in float32: ((x / 255) - mean) / std, HWC uint8 -> CHW float32."""
    x = img_u8.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    mean = np.array(MEAN, dtype=np.float32).reshape(3, 1, 1)
    std = np.array(STD, dtype=np.float32).reshape(3, 1, 1)
    return ((x - mean) / std).astype(np.float32)


def preprocess_frames(frames_u8: np.ndarray, out_h: int = 256, out_w: int = 256) -> np.ndarray:
    """(..., H, W, 3) uint8 -> (..., 3, out_h, out_w) float32, the model-contract input."""
    lead = frames_u8.shape[:-3]
    flat = frames_u8.reshape((-1,) + frames_u8.shape[-3:])
    out = np.stack([to_tensor_normalize(pil_bicubic_resize_u8(f, out_h, out_w)) for f in flat])
    return out.reshape(lead + (3, out_h, out_w))
