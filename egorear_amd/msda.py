"""The reference's native-op boundary on MI355X: `mmcv.ops.multi_scale_deform_attn.MultiScaleDeformableAttnFunction`.

`models/utils/deform_attn.py:9,155-162` imports this autograd Function from mmcv==2.2.0 (a CUDA extension, pin
README.md:134) and calls `.apply(value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
im2col_step)`.  This module offers the same class, argument order, shapes and gradient set on top of
`egr_msda_fwd_f32` / `egr_msda_bwd_f32` (include/egorear_hip.h), and `install_mmcv_shim()` registers it under the mmcv
module path so that the reference's `deform_attn.py` runs unmodified on ROCm, where the mmcv extension is not built.

The estimator classes of this package do not go through here (they use the sample-then-project kernels);
this is the narrow replacement for a maintainer who keeps the reference modules and only needs the op.
There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

import sys
import types
from typing import Tuple

import torch

from . import hip


def _aligned(t: torch.Tensor) -> torch.Tensor:
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone(memory_format=torch.contiguous_format)


def _check_shapes(value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
    if value.dim() != 4:
        raise ValueError(f"value must be (N, Lin, heads, D), got {tuple(value.shape)}")
    n, lin, heads, d = value.shape
    if sampling_locations.dim() != 6 or sampling_locations.shape[-1] != 2:
        raise ValueError(f"sampling_locations must be (N, Lq, heads, L, P, 2), got {tuple(sampling_locations.shape)}")
    _, lq, _, levels, points, _ = sampling_locations.shape
    if tuple(sampling_locations.shape[:3]) != (n, lq, heads):
        raise ValueError("sampling_locations does not match value in batch / heads")
    if tuple(attention_weights.shape) != (n, lq, heads, levels, points):
        raise ValueError(f"attention_weights must be {(n, lq, heads, levels, points)}, got {tuple(attention_weights.shape)}")
    if tuple(spatial_shapes.shape) != (levels, 2) or tuple(level_start_index.shape) != (levels,):
        raise ValueError("spatial_shapes must be (L, 2) and level_start_index (L,)")
    if spatial_shapes.dtype != torch.int64 or level_start_index.dtype != torch.int64:
        raise ValueError("spatial_shapes and level_start_index must be int64")
    return n, lin, heads, d, lq, levels, points


def msda_forward(value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
    """out (N, Lq, heads*D) = egr_msda_fwd_f32(...).  All tensors on the HIP device, fp32 / int64."""
    n, lin, heads, d, lq, levels, points = _check_shapes(value, spatial_shapes, level_start_index, sampling_locations, attention_weights)
    value, loc, aw = _aligned(value), _aligned(sampling_locations), _aligned(attention_weights)
    shapes, starts = spatial_shapes.contiguous(), level_start_index.contiguous()
    out = torch.empty((n, lq, heads * d), device=value.device, dtype=torch.float32)
    hip._launch("egr_msda_fwd_f32", hip.lib.egr_msda_fwd_f32, hip._p(value), hip._p(shapes, torch.int64), hip._p(starts, torch.int64),
                hip._p(loc), hip._p(aw), n, lin, heads, d, lq, levels, points, hip._p(out), hip._stream(),
                flops=2.0 * 4 * n * lq * heads * levels * points * d, nbytes=4.0 * (4 * n * lq * heads * levels * points * d + out.numel()))
    return out


def msda_backward(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, grad_output):
    """(grad_value, grad_sampling_locations, grad_attention_weights) = egr_msda_bwd_f32(...)."""
    n, lin, heads, d, lq, levels, points = _check_shapes(value, spatial_shapes, level_start_index, sampling_locations, attention_weights)
    if tuple(grad_output.shape) != (n, lq, heads * d):
        raise ValueError(f"grad_output must be {(n, lq, heads * d)}, got {tuple(grad_output.shape)}")
    value, loc, aw, go = _aligned(value), _aligned(sampling_locations), _aligned(attention_weights), _aligned(grad_output)
    shapes, starts = spatial_shapes.contiguous(), level_start_index.contiguous()
    gv = torch.empty_like(value)
    gl = torch.empty_like(loc)
    ga = torch.empty_like(aw)
    hip._launch("egr_msda_bwd_f32", hip.lib.egr_msda_bwd_f32, hip._p(value), hip._p(shapes, torch.int64), hip._p(starts, torch.int64),
                hip._p(loc), hip._p(aw), hip._p(go), n, lin, heads, d, lq, levels, points, hip._p(gv), hip._p(gl), hip._p(ga),
                hip._stream())
    return gv, gl, ga


# ---- the op as a torch.library operator ("wrapped as torch.library ops with ... Meta implementations") ----------
# `egorear_amd::msda_fwd` / `egorear_amd::msda_bwd`: opaque to Dynamo (the body launches through ctypes), shape-inferred through the
# fake (Meta) implementations below, differentiable through register_autograd - so a module that keeps the reference's own
# MSDeformAttn.forward (deform_attn.py:90-168) compiles under run.py:7-9's torch.compile(model.network) WITHOUT a graph break at the
# op.  Device: the HIP device only; CPU tensors have no kernel (NotImplementedError from the dispatcher) - there is no CPU path.

@torch.library.custom_op("egorear_amd::msda_fwd", mutates_args=(), device_types="cuda")
def msda_fwd_op(value: torch.Tensor, spatial_shapes: torch.Tensor, level_start_index: torch.Tensor, sampling_locations: torch.Tensor,
                attention_weights: torch.Tensor) -> torch.Tensor:
    return msda_forward(value, spatial_shapes, level_start_index, sampling_locations, attention_weights)


@msda_fwd_op.register_fake
def _msda_fwd_fake(value, spatial_shapes, level_start_index, sampling_locations, attention_weights):
    n, lin, heads, d, lq, levels, points = _check_shapes(value, spatial_shapes, level_start_index, sampling_locations, attention_weights)
    return value.new_empty((n, lq, heads * d), dtype=torch.float32)


@torch.library.custom_op("egorear_amd::msda_bwd", mutates_args=(), device_types="cuda")
def msda_bwd_op(value: torch.Tensor, spatial_shapes: torch.Tensor, level_start_index: torch.Tensor, sampling_locations: torch.Tensor,
                attention_weights: torch.Tensor, grad_output: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    return msda_backward(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, grad_output)


@msda_bwd_op.register_fake
def _msda_bwd_fake(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, grad_output):
    n, lin, heads, d, lq, levels, points = _check_shapes(value, spatial_shapes, level_start_index, sampling_locations, attention_weights)
    if tuple(grad_output.shape) != (n, lq, heads * d):
        raise ValueError(f"grad_output must be {(n, lq, heads * d)}, got {tuple(grad_output.shape)}")
    return (value.new_empty(value.shape, dtype=torch.float32), sampling_locations.new_empty(sampling_locations.shape, dtype=torch.float32),
            attention_weights.new_empty(attention_weights.shape, dtype=torch.float32))


def _msda_setup_context(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _msda_autograd(ctx, grad_output):
    value, shapes, starts, loc, aw = ctx.saved_tensors
    gv, gl, ga = msda_bwd_op(value, shapes, starts, loc, aw, grad_output.to(torch.float32).contiguous())
    return gv, None, None, gl, ga


msda_fwd_op.register_autograd(_msda_autograd, setup_context=_msda_setup_context)


class MultiScaleDeformableAttnFunction:
    """Same call as mmcv's autograd Function: `apply(value, value_spatial_shapes, value_level_start_index, sampling_locations,
    attention_weights, im2col_step) -> (N, Lq, heads*D)`; gradients for value, sampling_locations, attention_weights (once
    differentiable, like mmcv's).  `apply` is a plain function around the `egorear_amd::msda_fwd` operator, so both eager autograd and
    torch.compile see one differentiable op."""

    @staticmethod
    def apply(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
        # mmcv chunks the batch by im2col_step and requires it to divide the batch; the result does not depend on it
        step = min(int(value.shape[0]), int(im2col_step)) if int(value.shape[0]) > 0 else 1
        if step <= 0 or value.shape[0] % step != 0:
            raise RuntimeError(f"batch({value.shape[0]}) must divide im2col_step({step})")
        if not value.is_cuda and value.device.type != "meta":
            raise RuntimeError("egorear_amd.msda: no CPU path - tensors must live on the HIP device")
        return msda_fwd_op(value.to(torch.float32), value_spatial_shapes, value_level_start_index, sampling_locations.to(torch.float32),
                           attention_weights.to(torch.float32))


def install_mmcv_shim(force: bool = False) -> None:
    """Register `mmcv.ops.multi_scale_deform_attn` (and its parents, when mmcv itself is absent) in `sys.modules` so that
    `from mmcv.ops.multi_scale_deform_attn import MultiScaleDeformableAttnFunction` (deform_attn.py:9) resolves to the
    class above.  A real mmcv installation is left alone unless `force`."""
    name = "mmcv.ops.multi_scale_deform_attn"
    if not force:
        try:
            __import__(name)
            return
        except Exception:  # noqa: BLE001 - absent or built without the extension
            pass
    for parent in ("mmcv", "mmcv.ops"):
        if parent not in sys.modules or force:
            m = types.ModuleType(parent)
            m.__path__ = []  # a package
            sys.modules[parent] = m
    mod = types.ModuleType(name)
    mod.MultiScaleDeformableAttnFunction = MultiScaleDeformableAttnFunction
    mod.__doc__ = "egorear_amd.msda shim of mmcv.ops.multi_scale_deform_attn (MI355X)"
    sys.modules[name] = mod
    sys.modules["mmcv.ops"].multi_scale_deform_attn = mod
    sys.modules["mmcv"].ops = sys.modules["mmcv.ops"]
    sys.modules["mmcv.ops"].MultiScaleDeformableAttnFunction = MultiScaleDeformableAttnFunction
