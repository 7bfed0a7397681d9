"""Deterministic, platform-independent synthetic weights and inputs.

There is no network on the build or GPU boxes, so neither trained EgoRear
checkpoints nor the Ego4View datasets exist.  Everything that needs numbers
(parity tests, goldens, bench.py, smoke) draws them from the counter-based
generator below: value(key, seed, i) = splitmix64-finalizer(fnv1a(key) ^ seed + i),
evaluated with wrapping uint64 numpy arithmetic, so the same key/seed gives the
same float32 bits on any host.  The 504 MB full-model state_dict is therefore
regenerated on the GPU box rather than shipped.

Scaling is chosen per tensor family so that activations stay O(1) through the
ReLU stacks, BatchNorm running statistics are non-trivial, heatmap maxima fall
on both sides of the 0.5 validity threshold, and the deformable-attention
offset / weight projections are non-zero (the reference zero-initialises them,
deform_attn.py:67-88, which would leave the sampling path unexercised).
"""
from __future__ import annotations

import math
import re
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

_M64 = (1 << 64) - 1


def _fnv1a(text: str) -> int:
    h = 0xCBF29CE484222325
    for b in text.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _M64
    return h


def _mix_scalar(x: int) -> int:
    x &= _M64
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & _M64
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & _M64
    x ^= x >> 31
    return x


def uniform01(key: str, seed: int, n: int) -> np.ndarray:
    """n float32 values in [0,1) with 24 random bits each (exact in fp32)."""
    base = np.uint64(_mix_scalar(_fnv1a(key) ^ _mix_scalar(seed + 0x9E3779B97F4A7C15)))
    out = np.empty(n, dtype=np.float32)
    step = 1 << 24
    with np.errstate(over="ignore"):
        for s in range(0, n, step):
            e = min(n, s + step)
            x = np.arange(s, e, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + base
            x ^= x >> np.uint64(30)
            x *= np.uint64(0xBF58476D1CE4E5B9)
            x ^= x >> np.uint64(27)
            x *= np.uint64(0x94D049BB133111EB)
            x ^= x >> np.uint64(31)
            out[s:e] = (x >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return out


def uniform(key: str, seed: int, shape: Tuple[int, ...], lo: float, hi: float) -> torch.Tensor:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(key, seed, n)
    v = u * np.float32(hi - lo) + np.float32(lo)
    return torch.from_numpy(v.reshape(shape))


def normalish(key: str, seed: int, shape: Tuple[int, ...]) -> torch.Tensor:
    """Zero-mean unit-variance, bell-shaped: sum of four uniforms (Irwin-Hall)."""
    n = int(np.prod(shape))
    u = uniform01(key, seed, 4 * n).reshape(4, n)
    v = (u.sum(axis=0, dtype=np.float32) - np.float32(2.0)) * np.float32(math.sqrt(3.0))
    return torch.from_numpy(v.reshape(shape))


# --------------------------------------------------------------------------- weights

_BN_RE = re.compile(r"(\.bn\d\.|\.downsample\.1\.|layer_s2\.1\.)")
_HM_OUT_RE = re.compile(r"(conv_heatmap\.(weight|bias)$|conv_heatmap_layers_stereo_(front|back)\.9\.|conv_heatmap_layers\.0\.7\.)")
_LN_RE = re.compile(r"(norm_cross|norm_spatial|norm_ffn|post_norm\.\d+)\.")


def _fan_in(shape: Tuple[int, ...]) -> int:
    f = 1
    for d in shape[1:]:
        f *= d
    return max(f, 1)


def synth_tensor(key: str, shape: Tuple[int, ...], dtype: torch.dtype, seed: int) -> torch.Tensor | None:
    """One state_dict entry.  Returns None for structural integer buffers
    (spatial_shapes / start_index) whose values the module constructor fixes."""
    shape = tuple(int(s) for s in shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.tensor(100, dtype=torch.long)
    if not dtype.is_floating_point:
        return None
    if _BN_RE.search(key):
        if leaf == "weight":
            if ".bn2." in key:  # damp the residual branch so the trunk's second moment stays O(1)
                return uniform(key, seed, shape, 0.2, 0.5)
            return uniform(key, seed, shape, 0.6, 1.4)
        if leaf == "bias":
            return uniform(key, seed, shape, -0.2, 0.2)
        if leaf == "running_mean":
            return uniform(key, seed, shape, -0.2, 0.2)
        if leaf == "running_var":
            return uniform(key, seed, shape, 0.5, 1.5)
    if _LN_RE.search(key):
        if leaf == "weight":
            return uniform(key, seed, shape, 0.6, 1.4)
        return uniform(key, seed, shape, -0.1, 0.1)
    if key.endswith("frame_feat_multi_view_pos_embed"):
        return uniform(key, seed, shape, -0.5, 0.5)
    if key.endswith("joint_query_embed.weight"):
        return uniform(key, seed, shape, -1.0, 1.0)
    if ".sampling_offsets." in key:
        if leaf == "weight":
            a = 2.0 / math.sqrt(_fan_in(shape))
            return uniform(key, seed, shape, -a, a)
        return uniform(key, seed, shape, -8.0, 8.0)  # pixels, cf. the ring init's 1..16
    if ".attention_weights." in key:
        if leaf == "weight":
            a = 2.0 / math.sqrt(_fan_in(shape))
            return uniform(key, seed, shape, -a, a)
        return uniform(key, seed, shape, -1.0, 1.0)
    if key.endswith("mlp_pred.2.weight"):
        a = 20.0 * math.sqrt(3.0 / _fan_in(shape))
        return uniform(key, seed, shape, -a, a)
    if key.endswith("mlp_pred.2.bias"):
        # plausible device-relative joints (cm): spread over a body-sized box
        b = uniform(key, seed, shape, -1.0, 1.0).reshape(-1, 3)
        b = b * torch.tensor([35.0, 30.0, 45.0]) + torch.tensor([0.0, 15.0, -35.0])
        return b.reshape(shape).contiguous()
    if _HM_OUT_RE.search(key):
        # 15-channel heatmap output convs: keep map maxima near the 0.5 validity threshold
        if leaf == "bias":
            return uniform(key, seed, shape, -0.05, 0.05)
        a = math.sqrt(0.06 / _fan_in(shape))
        return uniform(key, seed, shape, -a, a)
    if leaf == "bias":
        return uniform(key, seed, shape, -0.1, 0.1)
    if leaf == "weight" and len(shape) >= 2:
        gain = 6.0 if ("conv" in key or "layer_s" in key or "proj_layers" in key or "head" in key) else 3.0
        a = math.sqrt(gain / _fan_in(shape))
        return uniform(key, seed, shape, -a, a)
    return uniform(key, seed, shape, -0.5, 0.5)


def synth_state_dict(spec: Iterable[Tuple[str, Tuple[int, ...], torch.dtype]], seed: int = 42) -> Dict[str, torch.Tensor]:
    """spec: iterable of (key, shape, dtype) — typically from module.state_dict()."""
    out: Dict[str, torch.Tensor] = {}
    for key, shape, dtype in spec:
        t = synth_tensor(key, tuple(shape), dtype, seed)
        if t is not None:
            out[key] = t.to(dtype)
    return out


def spec_of(module: torch.nn.Module):
    return [(k, tuple(v.shape), v.dtype) for k, v in module.state_dict().items()]


def load_synth(module: torch.nn.Module, seed: int = 42) -> Dict[str, torch.Tensor]:
    """Fill `module` in place with synthetic weights (strict over the keys generated)."""
    sd = synth_state_dict(spec_of(module), seed)
    full = module.state_dict()
    for k, v in sd.items():
        full[k] = v
    module.load_state_dict(full, strict=True)
    return sd


# --------------------------------------------------------------------------- inputs

def synth_images(batch: int, views: int, seed: int = 1234, size: int = 256, scale: float = 1.0) -> torch.Tensor:
    """Model-contract input (B,V,3,size,size) fp32: bell-shaped noise blended with a
    low-frequency pattern so that heatmaps are not flat (the 872x872
    uint8 -> 256 bicubic pre-processing is a "next" row and is not on the timed path)."""
    shape = (batch, views, 3, size, size)
    noise = normalish("img", seed, shape)
    yy = torch.linspace(-1.0, 1.0, size).view(1, 1, 1, size, 1)
    xx = torch.linspace(-1.0, 1.0, size).view(1, 1, 1, 1, size)
    ph = uniform("img.phase", seed, (batch, views, 3, 1, 1), 0.0, 6.2831853)
    low = torch.sin(3.0 * xx + ph) * torch.cos(2.0 * yy - ph)
    return ((0.7 * noise + 0.8 * low) * scale).contiguous()


def synth_raw_frames(batch: int, views: int, seed: int = 1234, size: int = 872) -> torch.Tensor:
    """Raw camera frames as the loader sees them: uint8 (B, V, size, size, 3), uniform noise blended with a
    low-frequency pattern (SURVEY.md §8d "Synthetic inputs")."""
    shape = (batch, views, size, size, 3)
    n = int(np.prod(shape))
    noise = uniform01("raw", seed, n).reshape(shape)
    yy = np.linspace(0.0, 1.0, size, dtype=np.float32).reshape(1, 1, size, 1, 1)
    xx = np.linspace(0.0, 1.0, size, dtype=np.float32).reshape(1, 1, 1, size, 1)
    ph = uniform01("raw.phase", seed, batch * views * 3).reshape(batch, views, 1, 1, 3)
    low = 0.5 + 0.5 * np.sin(6.2831853 * (1.5 * xx + ph)) * np.cos(6.2831853 * (yy - ph))
    img = np.clip((0.55 * noise + 0.45 * low) * 255.0, 0, 255).astype(np.uint8)
    return torch.from_numpy(img)


def synth_gt_pose(batch: int, seed: int = 1235) -> torch.Tensor:
    g = normalish("gt_pose", seed, (batch, 16, 3)) * 30.0
    return g + torch.tensor([0.0, 20.0, 40.0])


def synth_joint_px(batch: int, views: int = 4, joints: int = 15, seed: int = 1237, image_size: int = 872) -> torch.Tensor:
    """(B,V,J,2) fp32 joint positions in image pixels for GT heat maps; ~10 % fall outside the image (empty maps)."""
    return uniform("gt_joint_px", seed, (batch, views, joints, 2), -0.06 * image_size, 1.06 * image_size)


def synth_coord_trans_mat(batch: int, seed: int = 1236) -> torch.Tensor:
    """(B,4,4,4) fp32 rigid transforms: rotation <= 15 deg about a random axis, |t| <= 0.1 m."""
    ax = normalish("ctm.axis", seed, (batch, 4, 3))
    ax = ax / ax.norm(dim=-1, keepdim=True).clamp_min(1e-6)
    ang = uniform("ctm.angle", seed, (batch, 4, 1), -0.2618, 0.2618)
    t = uniform("ctm.t", seed, (batch, 4, 3), -0.1, 0.1)
    K = torch.zeros(batch, 4, 3, 3)
    K[..., 0, 1], K[..., 0, 2] = -ax[..., 2], ax[..., 1]
    K[..., 1, 0], K[..., 1, 2] = ax[..., 2], -ax[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -ax[..., 1], ax[..., 0]
    eye = torch.eye(3).expand(batch, 4, 3, 3)
    s, c = torch.sin(ang)[..., None], torch.cos(ang)[..., None]
    R = eye + s * K + (1.0 - c) * (K @ K)
    M = torch.zeros(batch, 4, 4, 4)
    M[..., :3, :3] = R
    M[..., :3, 3] = t
    M[..., 3, 3] = 1.0
    return M.float().contiguous()
