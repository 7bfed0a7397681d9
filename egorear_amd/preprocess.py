"""On-GPU frame pre-processing: uint8 (…, H, W, 3) camera frames -> fp32 (…, 3, 256, 256) model input.

Replaces the reference's CPU loader transform (datasets/ego4view_syn/ego4view_syn_pose3d.py:41-44,159-162):
`Image.resize([256, 256], Image.BICUBIC)` (Pillow's antialiased fixed-point resampling) + `ToTensor` +
`Normalize(ImageNet mean/std)`.  The resampling tables are computed here on the host, in float64 and in the
exact order Pillow's precompute_coeffs / normalize_coeffs_8bpc use, so the GPU result equals Pillow's uint8 image
bit for bit.  Two passes through HBM by default (egr_preprocess_u8_f32); a one-launch form that keeps the uint8 intermediate of
Pillow's two passes in LDS exists too (egr_preprocess_fused_u8_f32, EGR_PREPROCESS_FUSED=1: identical bytes, measured slower -
the arithmetic, not the memory traffic, is what bounds this step; see FramePreprocessor.__init__).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Tuple

import numpy as np
import torch

from . import hip

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
_PRECISION_BITS = 22


def _cubic(x: float) -> float:
    x = abs(x)
    if x < 1.0:
        return (1.5 * x - 2.5) * x * x + 1.0          # a = -0.5: ((a+2)x - (a+3))x^2 + 1
    if x < 2.0:
        return (((x - 5.0) * x + 8.0) * x - 4.0) * -0.5
    return 0.0


def resample_tables(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, int]:
    """Window bounds (out,2) and 22-bit fixed-point weights (out, ksize) of one resampling pass."""
    scale = in_size / out_size
    fscale = max(scale, 1.0)
    support = 2.0 * fscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    coef = np.zeros((out_size, ksize), np.int32)
    inv = 1.0 / fscale
    one = float(1 << _PRECISION_BITS)
    for o in range(out_size):
        center = (o + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        w = [_cubic((x + lo - center + 0.5) * inv) for x in range(hi - lo)]
        total = 0.0
        for v in w:
            total += v
        for i, v in enumerate(w):
            if total != 0.0:
                v = v / total
            coef[o, i] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)   # C cast: truncation toward zero
        bounds[o] = (lo, hi - lo)
    return bounds, coef, ksize


class FramePreprocessor:
    """Callable: uint8 frames on the GPU -> normalised fp32 network input."""

    def __init__(self, in_hw=(872, 872), out_hw=(256, 256), device="cuda"):
        self.in_hw, self.out_hw = tuple(in_hw), tuple(out_hw)
        self.device = torch.device(device)
        bh, ch, self.kh = resample_tables(in_hw[1], out_hw[1])   # horizontal pass: widths
        bv, cv, self.kv = resample_tables(in_hw[0], out_hw[0])   # vertical pass: heights
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self.bh, self.ch, self.bv, self.cv = up(bh), up(ch), up(bv), up(cv)
        self._mean = (C.c_float * 3)(*MEAN)
        self._std = (C.c_float * 3)(*STD)
        # source rows the largest 32-output-row band of the vertical pass touches (sizes the fused kernel's LDS band)
        bvh = np.ascontiguousarray(bv, dtype=np.int32)
        self.band_rows = int(hip.lib.egr_preprocess_band_rows(bvh.ctypes.data_as(C.c_void_p), out_hw[0]))
        # Measured on MI355X (64 views of 872 x 872): two passes 108 us, fused 125 us.  Pillow's exact 22-bit fixed-point
        # resampling costs ~125 integer VALU operations per horizontally resized pixel, so both forms are VALU-bound (2.1 G
        # lane-operations per 64 views ~ 66 us of pure issue time on the whole chip), not HBM-bound; the fused kernel saves the
        # intermediate's HBM round trip but recomputes 11 % of the rows (band overlap) and runs at one workgroup per CU (its
        # 94 KB LDS band).  The two-pass form stays the default; EGR_PREPROCESS_FUSED=1 selects the one-launch form (it falls
        # back to two passes by itself when it refuses a shape).
        import os
        self.fused = os.environ.get("EGR_PREPROCESS_FUSED", "0") == "1"

    def __call__(self, frames: torch.Tensor, return_u8: bool = False):
        if not frames.is_cuda or frames.dtype != torch.uint8:
            raise RuntimeError("egorear_amd.preprocess: uint8 frames on the HIP device expected (no CPU path)")
        if tuple(frames.shape[-3:]) != self.in_hw + (3,):
            raise RuntimeError(f"egorear_amd.preprocess: expected (..., {self.in_hw[0]}, {self.in_hw[1]}, 3), got {tuple(frames.shape)}")
        frames = frames.contiguous()
        lead = tuple(frames.shape[:-3])
        n = int(np.prod(lead)) if lead else 1
        H, W = self.in_hw
        oh, ow = self.out_hw
        dst = torch.empty(lead + (3, oh, ow), device=frames.device, dtype=torch.float32)
        u8 = torch.empty(lead + (oh, ow, 3), device=frames.device, dtype=torch.uint8) if return_u8 else None
        vp = hip._pv
        if self.fused:
            try:
                hip._launch("egr_preprocess_u8_f32", hip.lib.egr_preprocess_fused_u8_f32, vp(frames), n, H, W, oh, ow, vp(self.bh), vp(self.ch),
                            self.kh, vp(self.bv), vp(self.cv), self.kv, self.band_rows, C.cast(self._mean, C.c_void_p),
                            C.cast(self._std, C.c_void_p), vp(dst), vp(u8), hip._stream(), nbytes=float(n) * (H * W * 3 + oh * ow * 12))
                return (dst, u8) if return_u8 else dst
            except RuntimeError as exc:
                if "EGR_EINVAL" not in str(exc):
                    raise
                self.fused = False      # this shape is outside the fused kernel's limits: two passes from now on
        tmp = torch.empty((n, H, ow, 3), device=frames.device, dtype=torch.uint8)
        hip._launch("egr_preprocess_u8_f32", hip.lib.egr_preprocess_u8_f32, vp(frames), n, H, W, oh, ow, vp(self.bh), vp(self.ch),
                    self.kh, vp(self.bv), vp(self.cv), self.kv, C.cast(self._mean, C.c_void_p), C.cast(self._std, C.c_void_p),
                    vp(tmp), vp(dst), vp(u8), hip._stream(),
                    nbytes=float(n) * (H * W * 3 + 2 * H * ow * 3 + oh * ow * 12))
        return (dst, u8) if return_u8 else dst
