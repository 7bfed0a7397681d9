"""ctypes binding of the training-step entry points (include/egorear_train.h) of libegorear_hip.so.

Same rules as egorear_amd.hip: torch only owns memory and streams, shapes are checked on the host before a pointer is
handed to a kernel, and there is no fallback.
"""
from __future__ import annotations

import ctypes as C
import threading
from typing import Optional

import torch

from .hip import Img, NMap, _cont, _launch, _p, _stream, lib

TRAIN_EXPORTS = [
    "egr_bn_blocks", "egr_bn_stats_f32", "egr_scale_shift_f32", "egr_bn_backward_f32", "egr_relu_bwd_f32", "egr_add_f32", "egr_mse_loss_f32",
    "egr_bn_stats_ex_f32", "egr_bn_backward_ex_f32", "egr_record_bound_f32", "egr_bn_finalize_f32", "egr_transpose_f32",
    "egr_gelu_f32", "egr_gelu_bwd_f32", "egr_rowmask_f32", "egr_fill_f32", "egr_maxpool_train_f32", "egr_maxpool_bwd_f32",
    "egr_upsample2x_bwd_f32", "egr_stem_wgrad_f32", "egr_planes_to_nhwc_f32", "egr_nhwc_to_planes_f32", "egr_stem_im2col_f32", "egr_layernorm_bwd_f32", "egr_joint_mha_bwd_f32",
    "egr_msda_gather_bwd_f32", "egr_colsum_f32", "egr_fold_rows_f32", "egr_jqa_sum_bwd_f32", "egr_rownorm_loss_f32",
    "egr_sumsq_f32", "egr_adamw_f32", "egr_adamw_dev_f32", "egr_set4_f32", "egr_bn_relu_maxpool_f32", "egr_bn_pool_backward_f32",
    "egr_repack_f32",   # the last one is bound in egorear_amd.repack
]


def _bind():
    vp, i32, i64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
    lib.egr_bn_blocks.argtypes = [i64]
    lib.egr_bn_stats_f32.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, sz, vp]
    lib.egr_scale_shift_f32.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]
    lib.egr_bn_backward_f32.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp, vp, vp, vp, sz, vp]
    lib.egr_relu_bwd_f32.argtypes = [vp, vp, vp, i64, vp]
    lib.egr_add_f32.argtypes = [vp, vp, vp, i64, vp]
    lib.egr_gelu_f32.argtypes = [vp, vp, i64, vp]
    lib.egr_gelu_bwd_f32.argtypes = [vp, vp, vp, i64, vp]
    lib.egr_rowmask_f32.argtypes = [vp, vp, i64, i32, vp]
    lib.egr_fill_f32.argtypes = [vp, f32, i64, vp]
    lib.egr_maxpool_train_f32.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.egr_maxpool_bwd_f32.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.egr_upsample2x_bwd_f32.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    lib.egr_planes_to_nhwc_f32.argtypes = [vp, i32, i64, i64, vp, i32, i32, i32, i32, vp]
    lib.egr_nhwc_to_planes_f32.argtypes = [vp, vp, i32, i64, i64, i32, i32, i32, i32, vp]
    lib.egr_stem_wgrad_f32.argtypes = [vp, i32, i64, i64, i32, i32, i32, vp, vp, vp, sz, i32, i64, vp]
    lib.egr_stem_im2col_f32.argtypes = [vp, i32, i64, i64, i32, i32, i32, vp, vp]
    lib.egr_layernorm_bwd_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, i32, vp]
    lib.egr_joint_mha_bwd_f32.argtypes = [vp, vp, vp, i32, i32, i32, i32, f32, vp]
    lib.egr_msda_gather_bwd_f32.argtypes = [vp, i32, vp, i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.egr_colsum_f32.argtypes = [vp, i64, i64, i32, vp, vp, i32, i32, i64, i64, i32, i64, vp]
    lib.egr_fold_rows_f32.argtypes = [vp, vp, i64, i32, i32, vp]
    lib.egr_jqa_sum_bwd_f32.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
    lib.egr_rownorm_loss_f32.argtypes = [vp, vp, i64, i32, i32, i64, i64, f32, vp, vp, vp]
    lib.egr_sumsq_f32.argtypes = [vp, i64, vp, i32, vp]
    lib.egr_mse_loss_f32.argtypes = [vp, vp, i64, f32, vp, vp, vp]
    lib.egr_bn_stats_ex_f32.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp]
    lib.egr_bn_backward_ex_f32.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, vp, vp, vp, vp, sz, vp, vp, vp]
    lib.egr_record_bound_f32.argtypes = [vp, vp, f32, f32, vp, vp]
    lib.egr_transpose_f32.argtypes = [vp, i32, i32, vp, vp]
    lib.egr_bn_finalize_f32.argtypes = [vp, i32, i64, i32, i32, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.egr_adamw_f32.argtypes = [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp, f32, vp]
    lib.egr_set4_f32.argtypes = [vp, f32, f32, f32, f32, vp]
    lib.egr_adamw_dev_f32.argtypes = [vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, vp, f32, vp]
    lib.egr_bn_relu_maxpool_f32.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]
    lib.egr_bn_pool_backward_f32.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp, vp, vp]
    for name in TRAIN_EXPORTS:
        if name != "egr_repack_f32":
            getattr(lib, name).restype = C.c_int32 if name == "egr_bn_blocks" else C.c_int


_bind()


def _dense(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_contiguous():
        raise RuntimeError(f"egorear_amd.train: {what} must be contiguous")
    return t


def _same(a: torch.Tensor, b: torch.Tensor, what: str):
    if a.shape != b.shape:
        raise RuntimeError(f"egorear_amd.train: {what}: shapes {tuple(a.shape)} vs {tuple(b.shape)}")


# --------------------------------------------------------------------------- BatchNorm (training mode)

class BNCtx:
    """What the backward of one grouped BatchNorm needs: raw conv output, batch statistics, alpha."""
    __slots__ = ("x", "mean", "invstd", "alpha", "rpg", "c", "groups", "xhat_max", "shift", "slot", "pool")


def bn_workspace(device) -> torch.Tensor:
    return torch.empty(3 * 512 * 2 * 1024 + 16, device=device, dtype=torch.float64)     # (sums + the batch extremes behind them)


def bn_train(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, running_mean: Optional[torch.Tensor],
             running_var: Optional[torch.Tensor], groups: int, ws: torch.Tensor, *, res: Optional[torch.Tensor] = None,
             relu: bool = True, momentum: float = 0.1, eps: float = 1e-5, out: Optional[torch.Tensor] = None,
             amax_out: Optional[torch.Tensor] = None, want_extremes: bool = False, slabs: Optional[int] = None,
             pool: Optional[tuple] = None):
    """x: dense (groups*n, h, w, c) raw conv output.  gamma/beta/running_*: (groups, c) contiguous (running_* updated in
    place).  Returns (y, ctx).  pool = (k, stride, pad): the MaxPool2d behind BatchNorm + ReLU in the same pass (the stem) - y is the
    POOLED tensor, the normalised one is never written (egr_bn_relu_maxpool_f32); bn_backward then takes the pooled gradient."""
    _dense(x, "bn input")
    c = x.shape[-1]
    rows = x.numel() // c
    if rows % groups:
        raise RuntimeError("egorear_amd.train.bn_train: rows not divisible by groups")
    rpg = rows // groups
    for t in (gamma, beta, running_mean, running_var):
        if t is not None and (t.numel() != groups * c or not t.is_contiguous()):
            raise RuntimeError("egorear_amd.train.bn_train: per-channel arrays must be contiguous (groups, c)")
    ctx = BNCtx()
    ctx.x, ctx.rpg, ctx.c, ctx.groups = x, rpg, c, groups
    st = torch.empty((5, groups, c), device=x.device, dtype=torch.float32)
    ctx.mean, ctx.invstd, ctx.alpha, shift, ctx.xhat_max = st[0], st[1], st[2], st[3], st[4]
    ctx.shift, ctx.slot, ctx.pool = shift, None, None
    if pool is not None and (res is not None or not relu or x.dim() != 4):
        raise RuntimeError("egorear_amd.train.bn_train: the pooled form is BatchNorm + ReLU + MaxPool2d on an NHWC tensor, no residual")
    # amax_out: the abs-max BOUND of y from the batch extremes (no pass over y); with a residual only when that carries a record
    res_rec = getattr(res, "_egr_amax", None) if res is not None else None
    if amax_out is not None and res is not None and res_rec is None:
        amax_out = None
    need = int(lib.egr_bn_blocks(rpg)) * groups * 3 * c
    if ws.numel() < need:                       # (workspace of an older size: statistics only)
        amax_out, want_extremes = None, False
    ext = amax_out is not None or want_extremes      # (the batch extremes cost the statistics pass ~25 %: only when a bound is wanted)
    if slabs is not None:
        # the conv launch that produced x left the per-tile statistics in `ws` (hip.conv2d bn_ws): only the finalize step remains
        _launch("egr_bn_stats_f32", lib.egr_bn_finalize_f32, _p(ws, torch.float64), int(slabs), rpg, c, groups, _p(gamma), _p(beta),
                _p(running_mean), _p(running_var), momentum, eps, _p(ctx.mean), _p(ctx.invstd), _p(ctx.alpha), _p(shift), _p(ctx.xhat_max),
                _p(res_rec, torch.int32) if amax_out is not None else None, _p(amax_out, torch.int32), _stream())
        ext = True
    else:
        _launch("egr_bn_stats_f32", lib.egr_bn_stats_ex_f32, _p(x), rpg, c, groups, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                momentum, eps, _p(ctx.mean), _p(ctx.invstd), _p(ctx.alpha), _p(shift), _p(ws, torch.float64), ws.numel(),
                _p(ctx.xhat_max) if ext else None, _p(res_rec, torch.int32) if amax_out is not None else None, _p(amax_out, torch.int32),
                _stream(), nbytes=4.0 * x.numel())
    if not ext:
        ctx.xhat_max = None
    if pool is not None:
        k, stride, pad = pool
        n, h, w, _ = x.shape
        ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        y = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.float32)
        ctx.slot, ctx.pool = torch.empty((n, ho, wo, c), device=x.device, dtype=torch.uint8), (k, stride, pad)
        _launch("egr_bn_relu_maxpool_f32", lib.egr_bn_relu_maxpool_f32, _p(x), _p(ctx.alpha), _p(shift), _p(y), _p(ctx.slot, torch.uint8), n, h, w, c,
                groups, k, stride, pad, _stream(), nbytes=4.0 * x.numel() + 5.0 * y.numel())
        if amax_out is not None:
            y._egr_amax = amax_out          # a maximum of normalised values: the bound of the normalised tensor holds
        return y, ctx
    y = out if out is not None else torch.empty_like(x)
    if res is not None:
        _same(res, x, "bn residual")
        _dense(res, "bn residual")
    _launch("egr_scale_shift_f32", lib.egr_scale_shift_f32, _p(x), _p(ctx.alpha), _p(shift), _p(res), _p(_dense(y, "bn out")), rpg, c,
            groups, 1 if relu else 0, _stream(), nbytes=4.0 * x.numel() * (3 if res is not None else 2))
    if amax_out is not None:
        y._egr_amax = amax_out          # (hip.Img picks it up: the consuming conv launch's pre-scale)
    return y, ctx


def bn_backward(ctx: BNCtx, dy: torch.Tensor, y: Optional[torch.Tensor], ws: torch.Tensor, want_dz: bool = False,
                amax_dx: Optional[torch.Tensor] = None):
    """dy: gradient w.r.t. the BN(+res)(+ReLU) output y (pass y=None when no ReLU follows).  Returns
    (dx, dgamma (groups,c), dbeta (groups,c), dz | None) with dz = dy*[y>0], the gradient of the residual branch.
    A ctx made with pool=...: dy is the gradient of the POOLED output (y is not needed: the mask is recomputed from x)."""
    if ctx.pool is not None:
        if tuple(dy.shape) != tuple(ctx.slot.shape) or want_dz:
            raise RuntimeError("egorear_amd.train.bn_backward: pooled form: dy has the pooled shape, no residual branch")
        _dense(dy, "bn backward dy")
        k, stride, pad = ctx.pool
        n, h, w, c = ctx.x.shape
        dx = torch.empty_like(ctx.x)
        dgb = torch.empty((2, ctx.groups, ctx.c), device=dy.device, dtype=torch.float32)
        if amax_dx is not None and (ctx.xhat_max is None or ws.numel() < int(lib.egr_bn_blocks(ctx.rpg)) * ctx.groups * 3 * ctx.c):
            amax_dx = None
        _launch("egr_bn_backward_f32", lib.egr_bn_pool_backward_f32, _p(dy), _p(ctx.slot, torch.uint8), _p(ctx.x), _p(ctx.mean), _p(ctx.invstd),
                _p(ctx.alpha), _p(ctx.shift), n, h, w, c, ctx.groups, k, stride, pad, _p(dgb[0]), _p(dgb[1]), _p(dx), _p(ws, torch.float64),
                ws.numel(), _p(ctx.xhat_max) if amax_dx is not None else None, _p(amax_dx, torch.int32), _stream(),
                nbytes=4.0 * ctx.x.numel() * 3 + 10.0 * dy.numel())
        if amax_dx is not None:
            dx._egr_amax = amax_dx
        return dx, dgb[0], dgb[1], None
    _same(dy, ctx.x, "bn backward dy")
    _dense(dy, "bn backward dy")
    if y is not None:
        _same(y, ctx.x, "bn backward y")
        _dense(y, "bn backward y")
    dx = torch.empty_like(ctx.x)
    dz = torch.empty_like(ctx.x) if want_dz else None
    dgb = torch.empty((2, ctx.groups, ctx.c), device=dy.device, dtype=torch.float32)
    if amax_dx is not None and (ctx.xhat_max is None or ws.numel() < int(lib.egr_bn_blocks(ctx.rpg)) * ctx.groups * 3 * ctx.c):
        amax_dx = None
    _launch("egr_bn_backward_f32", lib.egr_bn_backward_ex_f32, _p(dy), _p(y), _p(ctx.x), _p(ctx.mean), _p(ctx.invstd), _p(ctx.alpha),
            ctx.rpg, ctx.c, ctx.groups, _p(dgb[0]), _p(dgb[1]), _p(dx), _p(dz), _p(ws, torch.float64), ws.numel(),
            _p(ctx.xhat_max) if amax_dx is not None else None, _p(amax_dx, torch.int32), _stream(),
            nbytes=4.0 * dy.numel() * (7 if y is not None else 5))
    if amax_dx is not None:
        dx._egr_amax = amax_dx          # an upper bound of |dx| from the batch extremes
    if dz is not None and getattr(dy, "_egr_amax", None) is not None:
        dz._egr_amax = dy._egr_amax     # dz = dy * [y > 0]: bounded by dy
    return dx, dgb[0], dgb[1], dz


# --------------------------------------------------------------------------- element-wise

# The step in progress hands its abs-max arena (hip.AmaxArena) to the element-wise launches below: an output that is bounded by its
# inputs' records (a sum, a masked gradient, a pooling / up-sampling gradient) gets its own record from them - one 64-thread launch or
# none at all - instead of a pass over the tensor.  Per thread: one step per thread at a time.
_TLS = threading.local()


def set_arena(arena) -> None:
    _TLS.arena = arena


def transpose_into(dst: torch.Tensor, src: torch.Tensor) -> None:
    """dst (cols, rows) = src (rows, cols)^T, both dense fp32 (the tiled LDS transpose: a large Linear's data-gradient operand)."""
    rows, cols = src.shape
    if tuple(dst.shape) != (cols, rows):
        raise RuntimeError("egorear_amd.transpose_into: dst must be (cols, rows)")
    _launch("egr_transpose_f32", lib.egr_transpose_f32, _p(_dense(src, "src")), rows, cols, _p(_dense(dst, "dst")), _stream(),
            nbytes=8.0 * rows * cols)


def record_bound(out_t: torch.Tensor, a: Optional[torch.Tensor], b: Optional[torch.Tensor] = None, sa: float = 1.0, sb: float = 1.0) -> None:
    """Tag `out_t` with a record holding sa * max|a| [+ sb * max|b|] when the inputs carry records (a / b: tensors)."""
    arena = getattr(_TLS, "arena", None)
    ra = getattr(a, "_egr_amax", None) if a is not None else None
    rb = getattr(b, "_egr_amax", None) if b is not None else None
    if arena is None or ra is None or (b is not None and rb is None):
        return
    if b is None and sa == 1.0:
        out_t._egr_amax = ra                # the same bound: share the record
        return
    rec = arena.new()
    if rec is None:
        return
    _launch("egr_record_bound_f32", lib.egr_record_bound_f32, _p(ra, torch.int32), _p(rb, torch.int32), float(sa), float(sb),
            _p(rec, torch.int32), _stream())
    out_t._egr_amax = rec


def _elt(name, cfunc, n, *ptrs, nbytes=0.0):
    if n % 4:
        raise RuntimeError(f"egorear_amd.train.{name}: element count must be a multiple of 4")
    _launch(name, cfunc, *ptrs, n, _stream(), nbytes=nbytes)


def relu_bwd(dy: torch.Tensor, y: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _same(dy, y, "relu_bwd")
    out = out if out is not None else torch.empty_like(dy)
    _elt("egr_relu_bwd_f32", lib.egr_relu_bwd_f32, dy.numel(), _p(_dense(dy, "dy")), _p(_dense(y, "y")), _p(_dense(out, "dx")),
         nbytes=12.0 * dy.numel())
    if out is not dy:
        record_bound(out, dy)               # dx = dy * [y > 0]
    return out


def add(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _same(a, b, "add")
    out = out if out is not None else torch.empty_like(a)
    _elt("egr_add_f32", lib.egr_add_f32, a.numel(), _p(_dense(a, "a")), _p(_dense(b, "b")), _p(_dense(out, "out")), nbytes=12.0 * a.numel())
    if out is a or out is b:
        if getattr(out, "_egr_amax", None) is not None:
            out._egr_amax = None            # an in-place sum: the old record no longer bounds the tensor
    else:
        record_bound(out, a, b)             # |a + b| <= max|a| + max|b|
    return out


def gelu(z: torch.Tensor) -> torch.Tensor:
    h = torch.empty_like(z)
    _elt("egr_gelu_f32", lib.egr_gelu_f32, z.numel(), _p(_dense(z, "z")), _p(h))
    return h


def gelu_bwd(dh: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    _same(dh, z, "gelu_bwd")
    dz = torch.empty_like(z)
    _elt("egr_gelu_bwd_f32", lib.egr_gelu_bwd_f32, z.numel(), _p(_dense(dh, "dh")), _p(_dense(z, "z")), _p(dz))
    return dz


def fill(x: torch.Tensor, v: float = 0.0) -> torch.Tensor:
    _dense(x, "fill target")
    if x.numel() % 4:
        raise RuntimeError("egorear_amd.train.fill: element count must be a multiple of 4")
    _launch("egr_fill_f32", lib.egr_fill_f32, _p(x), float(v), x.numel(), _stream())
    return x


def zeros(shape, device) -> torch.Tensor:
    return fill(torch.empty(shape, device=device, dtype=torch.float32))


def rowmask_(x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """x (rows, c) *= mask[rows] (uint8), in place."""
    rows, c = x.shape
    if mask.numel() != rows:
        raise RuntimeError("egorear_amd.train.rowmask_: mask length")
    _launch("egr_rowmask_f32", lib.egr_rowmask_f32, _p(_dense(x, "x")), _p(mask, torch.uint8), rows, c, _stream())
    return x


# --------------------------------------------------------------------------- pooling / resampling

def maxpool_train(x: Img, k: int, stride: int, pad: int):
    if not x.t.is_contiguous():
        raise RuntimeError("egorear_amd.train.maxpool_train: contiguous NHWC input expected")
    ho = (x.h + 2 * pad - k) // stride + 1
    wo = (x.w + 2 * pad - k) // stride + 1
    y = torch.empty((x.n, ho, wo, x.c), device=x.t.device, dtype=torch.float32)
    slot = torch.empty((x.n, ho, wo, x.c), device=x.t.device, dtype=torch.uint8)
    _launch("egr_maxpool_train_f32", lib.egr_maxpool_train_f32, _p(x.t), _p(y), _p(slot, torch.uint8), x.n, x.h, x.w, x.c, k, stride,
            pad, _stream())
    out = Img(y)
    out.tag(x.amax)                           # a maximum of inputs: the input's record bounds the output
    return out, slot


def maxpool_bwd(dy: torch.Tensor, slot: torch.Tensor, in_hw, k: int, stride: int, pad: int) -> torch.Tensor:
    n, ho, wo, c = dy.shape
    h, w = in_hw
    if tuple(slot.shape) != (n, ho, wo, c) or (h + 2 * pad - k) // stride + 1 != ho or (w + 2 * pad - k) // stride + 1 != wo:
        raise RuntimeError("egorear_amd.train.maxpool_bwd: geometry mismatch")
    dx = torch.empty((n, h, w, c), device=dy.device, dtype=torch.float32)
    _launch("egr_maxpool_bwd_f32", lib.egr_maxpool_bwd_f32, _p(_dense(dy, "dy")), _p(_dense(slot, "slot"), torch.uint8), _p(dx), n, h, w, c,
            k, stride, pad, _stream())
    record_bound(dx, dy, None, float(((k + stride - 1) // stride) ** 2))       # an input pixel is the maximum of at most that many windows
    return dx


def upsample2x_bwd(dy: torch.Tensor, y: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dy (n, 2h, 2w, c) -> dx (n, h, w, c); with y given, dy is first masked by [y > 0] (fused up+ReLU forward)."""
    n, ho, wo, c = dy.shape
    if ho % 2 or wo % 2:
        raise RuntimeError("egorear_amd.train.upsample2x_bwd: odd output size")
    if y is not None:
        _same(dy, y, "upsample2x_bwd")
        _dense(y, "y")
    dx = torch.empty((n, ho // 2, wo // 2, c), device=dy.device, dtype=torch.float32)
    _launch("egr_upsample2x_bwd_f32", lib.egr_upsample2x_bwd_f32, _p(_dense(dy, "dy")), _p(y), _p(dx), n, ho // 2, wo // 2, c, _stream())
    # an input pixel collects interpolation weights of total ((2h - 1) / (h - 1))^2 <= 5.5 for h >= 4 (align_corners): 6 is a safe bound
    record_bound(dx, dy, None, 6.0)
    return dx


def planes_to_nhwc(planes: torch.Tensor, nmap: NMap, n: int, c: int, hw: int, cpad: int, base_offset: int = 0) -> torch.Tensor:
    """Channel-major planes (image n at nmap(n) floats from `base_offset`) -> (n, hw, cpad) channels-last, zero padded."""
    flat = planes.reshape(-1)
    last = (n - 1) // nmap.n_inner * nmap.stride_outer + (n - 1) % nmap.n_inner * nmap.stride_inner + c * hw
    if not planes.is_contiguous() or base_offset + last > flat.numel():
        raise RuntimeError("egorear_amd.train.planes_to_nhwc: map runs outside the tensor")
    y = torch.empty((n, hw, cpad), device=planes.device, dtype=torch.float32)
    _launch("egr_planes_to_nhwc_f32", lib.egr_planes_to_nhwc_f32, _p(flat[base_offset:]), nmap.n_inner, nmap.stride_inner,
            nmap.stride_outer, _p(y), n, c, hw, cpad, _stream())
    return y


def nhwc_to_planes(x: torch.Tensor, planes: torch.Tensor, nmap: NMap, c: int, base_offset: int = 0) -> torch.Tensor:
    """x (n, hw, cpad) channels-last -> its first c channels as (c, hw) planes inside `planes` (image n at nmap(n))."""
    n, hw, cpad = x.shape
    flat = planes.reshape(-1)
    last = (n - 1) // nmap.n_inner * nmap.stride_outer + (n - 1) % nmap.n_inner * nmap.stride_inner + c * hw
    if not planes.is_contiguous() or base_offset + last > flat.numel() or c > cpad:
        raise RuntimeError("egorear_amd.train.nhwc_to_planes: map runs outside the tensor")
    _launch("egr_nhwc_to_planes_f32", lib.egr_nhwc_to_planes_f32, _p(_dense(x, "x")), _p(flat[base_offset:]), nmap.n_inner,
            nmap.stride_inner, nmap.stride_outer, n, c, hw, cpad, _stream())
    return planes


def stem_wgrad(img: torch.Tensor, view0: int, nviews: int, dy: torch.Tensor, workspace: torch.Tensor, groups: int = 1) -> torch.Tensor:
    """Weight gradient of the stem conv for `groups` encoders (group g = views [view0 + g*nviews, ...)): img (B, V, 3, H, W),
    dy (groups*nviews*B, H/2, W/2, 64) view-major NHWC -> (groups, 64, 3, 7, 7)."""
    B, V, Cc, H, W = img.shape
    n = nviews * B
    if Cc != 3 or view0 + groups * nviews > V or tuple(dy.shape) != (groups * n, H // 2, W // 2, 64):
        raise RuntimeError("egorear_amd.train.stem_wgrad: shapes do not match")
    _cont(img, "input image batch")
    dw = torch.empty((groups, 64, 3, 7, 7), device=img.device, dtype=torch.float32)
    base = img.reshape(-1)[view0 * 3 * H * W:]
    _launch("egr_stem_wgrad_f32", lib.egr_stem_wgrad_f32, _p(base), B, V * 3 * H * W, 3 * H * W, n, H, W, _p(_dense(dy, "dy")), _p(dw),
            _p(workspace), workspace.numel(), groups, nviews * 3 * H * W, _stream(), flops=2.0 * groups * n * (H // 2) * (W // 2) * 64 * 147)
    return dw


def stem_im2col(img: torch.Tensor, view0: int, nviews: int) -> torch.Tensor:
    """img (B, V, 3, H, W) -> (nviews*B*H/2*W/2, 160) patch rows of views [view0, view0+nviews), view-major."""
    B, V, Cc, H, W = img.shape
    if Cc != 3 or view0 + nviews > V:
        raise RuntimeError("egorear_amd.train.stem_im2col: bad views / channels")
    _cont(img, "input image batch")
    n = nviews * B
    cols = torch.empty((n * (H // 2) * (W // 2), 160), device=img.device, dtype=torch.float32)
    base = img.reshape(-1)[view0 * 3 * H * W:]
    _launch("egr_stem_im2col_f32", lib.egr_stem_im2col_f32, _p(base), B, V * 3 * H * W, 3 * H * W, n, H, W, _p(cols), _stream())
    return cols


# --------------------------------------------------------------------------- LayerNorm / attention

def layernorm_bwd(dy: torch.Tensor, pre: torch.Tensor, gamma: torch.Tensor, groups: int = 1, eps: float = 1e-5):
    """-> (ds (rows,c), dgamma (groups*c), dbeta (groups*c))."""
    rows, c = pre.shape
    _same(dy, pre, "layernorm_bwd")
    if gamma.numel() != groups * c or rows % groups:
        raise RuntimeError("egorear_amd.train.layernorm_bwd: gamma / groups mismatch")
    ds = torch.empty_like(pre)
    dgb = torch.empty((2, groups * c), device=pre.device, dtype=torch.float32)
    stats = torch.empty((rows, 2), device=pre.device, dtype=torch.float32)
    _launch("egr_layernorm_bwd_f32", lib.egr_layernorm_bwd_f32, _p(_dense(dy, "dy")), _p(_dense(pre, "pre")), _p(_dense(gamma, "gamma")), _p(ds),
            _p(dgb[0]), _p(dgb[1]), _p(stats), rows, c, eps, rows // groups if groups > 1 else 0, _stream())
    return ds, dgb[0], dgb[1]


def joint_mha_bwd(qkv: torch.Tensor, dout: torch.Tensor, b: int, j: int, heads: int, d: int, scale: float) -> torch.Tensor:
    if tuple(qkv.shape) != (b * j, 3 * heads * d) or tuple(dout.shape) != (b * j, heads * d):
        raise RuntimeError("egorear_amd.train.joint_mha_bwd: shape mismatch")
    dqkv = torch.empty_like(qkv)
    _launch("egr_joint_mha_bwd_f32", lib.egr_joint_mha_bwd_f32, _p(_dense(qkv, "qkv")), _p(_dense(dout, "dout")), _p(dqkv), b, j, heads, d,
            scale, _stream())
    return dqkv


def msda_gather_bwd(feat: torch.Tensor, pos: Optional[torch.Tensor], offs_logits: torch.Tensor, anchors: torch.Tensor,
                    valid: torch.Tensor, B: int, V: int, J: int, heads: int, dh: int, hgt: int, wid: int, dg: torch.Tensor,
                    da: torch.Tensor, cfold: torch.Tensor, dfeat: Optional[torch.Tensor], dpos: Optional[torch.Tensor],
                    groups: int = 1) -> torch.Tensor:
    """Shapes as hip.msda_gather; dg (groups, rows, heads, cf), da (groups*rows, heads*dh), cfold (groups, heads*dh).
    dfeat like feat / dpos like pos are accumulated into (zero them first).  Returns dol (groups*rows, heads*48)."""
    cf = feat.shape[-1]
    rows = B * J * V
    C_ = heads * dh
    if tuple(feat.shape) != (V, B, hgt * wid, cf) or offs_logits.shape != (groups * B * J, heads * 48):
        raise RuntimeError("egorear_amd.train.msda_gather_bwd: feat / offs_logits shape")
    if dg.numel() != groups * rows * heads * cf or da.numel() != groups * rows * C_ or cfold.numel() != groups * C_:
        raise RuntimeError("egorear_amd.train.msda_gather_bwd: gradient shapes")
    if pos is not None and tuple(pos.shape) != (groups, V, hgt * wid, C_):
        raise RuntimeError("egorear_amd.train.msda_gather_bwd: pos shape")
    if dfeat is not None:
        _same(dfeat, feat, "dfeat")
    if dpos is not None:
        if pos is None:
            raise RuntimeError("egorear_amd.train.msda_gather_bwd: dpos without pos")
        _same(dpos, pos, "dpos")
    if anchors.numel() != B * V * J * 2 or valid.numel() != B * V * J:
        raise RuntimeError("egorear_amd.train.msda_gather_bwd: anchors / valid shape")
    dol = torch.empty((groups * rows, heads * 48), device=feat.device, dtype=torch.float32)
    _launch("egr_msda_gather_bwd_f32", lib.egr_msda_gather_bwd_f32, _p(_dense(feat, "feat")), cf, _p(pos), dh, _p(_dense(offs_logits, "ol")),
            _p(_dense(anchors, "anchors")), _p(_dense(valid, "valid"), torch.uint8), B, V, J, heads, hgt, wid, _p(_dense(dg, "dg")),
            _p(_dense(da, "da")), _p(_dense(cfold, "cfold")), _p(dol), _p(dfeat), _p(dpos), groups, _stream())
    return dol


# --------------------------------------------------------------------------- reductions

def colsum(x: torch.Tensor, ld: int, rows: int, c: int, scale: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
           accumulate: bool = False, groups: int = 1, gx: int = 0, gs: int = 0, cols_per_scale: int = 0, scale_stride: int = 0) -> torch.Tensor:
    """out[g, c] (+)= sum_r scale[g*gs + r] * x[g*gx + r*ld + c]; x is a storage-rooted view (first element = group 0 row 0).
    cols_per_scale > 0: column ch uses the scale vector at scale + (ch // cols_per_scale) * scale_stride (one per head)."""
    need = (groups - 1) * gx + (rows - 1) * ld + c
    if x.storage_offset() + need > x.untyped_storage().nbytes() // 4:
        raise RuntimeError("egorear_amd.train.colsum: reads past the end of x")
    nvec = (c + cols_per_scale - 1) // cols_per_scale if cols_per_scale > 0 else 1
    if scale is not None and scale.numel() < (groups - 1) * gs + (nvec - 1) * scale_stride + rows:
        raise RuntimeError("egorear_amd.train.colsum: scale too short")
    if out is None:
        out = torch.empty((groups, c), device=x.device, dtype=torch.float32)
    _launch("egr_colsum_f32", lib.egr_colsum_f32, _p(x), ld, rows, c, _p(scale), _p(out), 1 if accumulate else 0, groups, gx, gs,
            cols_per_scale, scale_stride, _stream())
    return out


def fold_rows(x: torch.Tensor, fold: int) -> torch.Tensor:
    rows, c = x.shape
    if rows % fold:
        raise RuntimeError("egorear_amd.train.fold_rows: rows not divisible")
    y = torch.empty((rows // fold, c), device=x.device, dtype=torch.float32)
    _launch("egr_fold_rows_f32", lib.egr_fold_rows_f32, _p(_dense(x, "x")), _p(y), rows // fold, fold, c, _stream())
    return y


def jqa_sum_bwd(dx: torch.Tensor, b: int, j: int, c: int, groups: int):
    if dx.numel() != b * j * c or b % groups:
        raise RuntimeError("egorear_amd.train.jqa_sum_bwd: shape")
    d_embed = torch.empty((groups, j, c), device=dx.device, dtype=torch.float32)
    d_bfb = torch.empty((b, c), device=dx.device, dtype=torch.float32)
    _launch("egr_jqa_sum_bwd_f32", lib.egr_jqa_sum_bwd_f32, _p(_dense(dx, "dx")), _p(d_embed), _p(d_bfb), b, j, c, b // groups, _stream())
    return d_embed, d_bfb


# --------------------------------------------------------------------------- loss / optimiser

def rownorm_loss(pred: torch.Tensor, gt: torch.Tensor, d: int, weight: float, loss: torch.Tensor, want_grad: bool = True,
                 rows: Optional[int] = None, inner: int = 1, ld_pred: Optional[int] = None, ld_gt: Optional[int] = None):
    """loss (device float64 scalar) += weight * mean_rows ||gt - pred||_2 over rows of length d; returns dpred | None.
    Default: dense (rows, d) operands.  With inner / ld_*: row r starts at (r // inner)*ld + (r % inner)*d, which reads
    channel-padded buffers in place; dpred then has pred's layout with the padding zeroed."""
    if loss.dtype != torch.float64 or loss.numel() != 1:
        raise RuntimeError("egorear_amd.train.rownorm_loss: loss must be one float64")
    if rows is None:
        _same(pred, gt, "rownorm_loss")
        if pred.numel() % d:
            raise RuntimeError("egorear_amd.train.rownorm_loss: bad shapes")
        rows, inner, ld_pred, ld_gt = pred.numel() // d, 1, d, d
    outer = (rows + inner - 1) // inner
    if rows % inner or outer * ld_pred > pred.numel() + (ld_pred - inner * d) or outer * ld_gt > gt.numel() + (ld_gt - inner * d):
        raise RuntimeError("egorear_amd.train.rownorm_loss: layout runs outside the operands")
    dpred = None
    if want_grad:
        dpred = torch.empty_like(pred) if ld_pred == inner * d else fill(torch.empty_like(pred))
    _launch("egr_rownorm_loss_f32", lib.egr_rownorm_loss_f32, _p(_dense(pred, "pred")), _p(_dense(gt, "gt")), rows, d, inner, ld_pred, ld_gt,
            float(weight), _p(loss, torch.float64), _p(dpred), _stream())
    return dpred


def mse_loss(pred: torch.Tensor, gt: torch.Tensor, weight: float, loss: torch.Tensor, want_grad: bool = True) -> Optional[torch.Tensor]:
    """loss (device float64 scalar) += weight * mean (pred - gt)^2; returns dpred = 2 weight (pred - gt) / n | None."""
    if loss.dtype != torch.float64 or loss.numel() != 1:
        raise RuntimeError("egorear_amd.train.mse_loss: loss must be one float64")
    _same(pred, gt, "mse_loss")
    if pred.numel() % 4:
        raise RuntimeError("egorear_amd.train.mse_loss: element count must be a multiple of 4")
    dpred = torch.empty_like(pred) if want_grad else None
    _launch("egr_mse_loss_f32", lib.egr_mse_loss_f32, _p(_dense(pred, "pred")), _p(_dense(gt, "gt")), pred.numel(), float(weight),
            _p(loss, torch.float64), _p(dpred), _stream(), nbytes=12.0 * pred.numel())
    return dpred


def sumsq(g: torch.Tensor, out: torch.Tensor, accumulate: bool = False):
    if out.dtype != torch.float64 or out.numel() != 1:
        raise RuntimeError("egorear_amd.train.sumsq: out must be one float64")
    _launch("egr_sumsq_f32", lib.egr_sumsq_f32, _p(_dense(g, "g")), g.numel(), _p(out, torch.float64), 1 if accumulate else 0, _stream())


def adamw(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, beta1: float, beta2: float, eps: float,
          weight_decay: float, step: int, grad_sumsq: Optional[torch.Tensor], clip: float):
    n = p.numel()
    if g.numel() != n or m.numel() != n or v.numel() != n:
        raise RuntimeError("egorear_amd.train.adamw: size mismatch")
    for t in (p, g, m, v):
        _dense(t, "adamw operand")
    _launch("egr_adamw_f32", lib.egr_adamw_f32, _p(p), _p(g), _p(m), _p(v), n, lr, beta1, beta2, eps, weight_decay, step,
            _p(grad_sumsq, torch.float64) if grad_sumsq is not None else None, clip, _stream(), nbytes=28.0 * n)


def adamw_dev(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, hyper: torch.Tensor, beta1: float, beta2: float,
              eps: float, weight_decay: float, grad_sumsq: Optional[torch.Tensor], clip: float):
    """adamw() with {lr, 1 - beta1^t, sqrt(1 - beta2^t)} read from the 3-float device tensor `hyper`."""
    n = p.numel()
    if g.numel() != n or m.numel() != n or v.numel() != n or hyper.numel() < 3 or hyper.dtype != torch.float32:
        raise RuntimeError("egorear_amd.train.adamw_dev: size mismatch")
    for t in (p, g, m, v, hyper):
        _dense(t, "adamw operand")
    _launch("egr_adamw_dev_f32", lib.egr_adamw_dev_f32, _p(p), _p(g), _p(m), _p(v), n, _p(hyper), beta1, beta2, eps, weight_decay,
            _p(grad_sumsq, torch.float64) if grad_sumsq is not None else None, clip, _stream(), nbytes=28.0 * n)


def set4(dst: torch.Tensor, a: float, b: float, c: float, d: float = 0.0):
    if dst.numel() < 4 or dst.dtype != torch.float32:
        raise RuntimeError("egorear_amd.train.set4: a 4-float device tensor expected")
    _launch("egr_set4_f32", lib.egr_set4_f32, _p(_dense(dst, "dst")), float(a), float(b), float(c), float(d), _stream())


from .hip import _guard_module  # noqa: E402  (a failed host-side check must not leave the launch-device record set)

_guard_module(globals())
