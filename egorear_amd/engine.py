"""Execution of the hot path on the HIP kernels.

Layout in HBM: every activation is fp32 channels-last, *view-major*:
(V, B, H, W, C).  The four camera views of a batch are four contiguous image
batches, so a stereo estimator (views 0-1 / 2-3), one refiner's own view, and the
4-view memory the deformable attention samples are all plain slices — no gather,
transpose or torch.cat anywhere on the path.  The reference's batch-major NCHW only
exists at the boundary: the stem kernel reads the (B,V,3,H,W) input through an image
map, the 15-channel heatmap convs write (B,V,15,64,64) planes through one, and feature
tensors handed back to callers are permuted *views* of the view-major buffers.

Grouped launches: modules that share an architecture but own their weights — the two
stereo estimators (front/back), the two initial-heatmap stacks, the four per-view
refiners — run as ONE launch per op with a group index selecting weights and
activations at uniform strides (view-major layout makes every group a contiguous
slice).  A forward of the full model is ~120 launches instead of ~330, and small-grid
layers fill the chip.

Weights are re-packed once per module/device (conv OIHW -> [cout_pad][cin/32][kh*kw][32],
BatchNorm as per-channel scale/shift applied in the conv epilogue, q/k/v and
offset/logit projections concatenated, the deformable-attention value path folded for
the sample-then-project form, group members stacked) and cached; `invalidate(module)`
drops the cache (load_state_dict does it automatically).
"""
from __future__ import annotations

import math
import os
import threading
from typing import Dict, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import hip
from .hip import ACT_GELU, ACT_NONE, ACT_RELU, RES_AFTER_ACT, RES_BEFORE_ACT, RES_NONE, RES_UP2_BEFORE_ACT, Img, NMap

_WORKSPACE_FLOATS = 16 << 20  # split-K partial slabs (64 MB)


# --------------------------------------------------------------------------- packed parameters

class PConv:
    """Packed conv / linear weights of `groups` same-shape modules (groups == 1: plain 2-D tensors)."""
    __slots__ = ("w", "scale", "shift", "cout", "cin", "kh", "kw", "stride", "pad", "groups")


# Weight operand of the implicit-GEMM launches: "f16x2" (default) = hip.pack_w6 image + its fp16 companion (hip.add_wh2): forward
# launches whose input carries an abs-max record multiply on the fp16 matrix cores (two planes, three products),
# the others on the bf16 matrix cores; "bf16x3" = the bf16 image only (three planes, six products, fp32-exact, §5b);
# "f32" = the packed fp32 matrix (fp32 matrix cores).  Matrices above W6_MAX_ELEMS are streamed once per step from HBM
# (mlp_pred.0: 67 M weights) and stay in the 4-byte format.
# (W_FORMAT, LAYER_H2, FUSED_LAYER, FUSED_QUERY are fields of hip.LaunchPolicy; the module attributes of those names are properties
# onto it - installed at the end of this file.)
W6_MAX_ELEMS = 1 << 24


def _w_operand(w: Optional[torch.Tensor], h2: bool = True):
    """h2=False: no fp16 companion image (the training step: its launches carry no abs-max records, and its images are re-split
    after every update)."""
    fmt = hip.policy().w_format
    if w is None or fmt not in ("bf16x3", "f16x2") or not w.is_cuda or w.shape[-1] % 32 != 0 or w.shape[-2] * w.shape[-1] > W6_MAX_ELEMS:
        return w
    w6 = hip.pack_w6(w)
    return hip.add_wh2(w6) if (h2 and fmt == "f16x2") else w6


def _npad(cout: int) -> int:
    return (cout + 31) // 32 * 32


def _pad_rows(w2d: torch.Tensor) -> torch.Tensor:
    cout = w2d.shape[0]
    npad = _npad(cout)
    if npad == cout:
        return w2d.contiguous()
    out = torch.zeros((npad, w2d.shape[1]), device=w2d.device, dtype=w2d.dtype)
    out[:cout] = w2d
    return out


def _pad_vec(v: Optional[torch.Tensor], cout: int) -> Optional[torch.Tensor]:
    if v is None:
        return None
    out = torch.zeros(_npad(cout), device=v.device, dtype=torch.float32)
    out[:cout] = v.float()
    return out


def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """OIHW -> [cout][cin/32][kh*kw][32] (the kernel's K order: channel chunk, tap, channel-in-chunk)."""
    co, ci, kh, kw = w.shape
    return w.float().permute(0, 2, 3, 1).reshape(co, kh * kw, ci // 32, 32).permute(0, 2, 1, 3).reshape(co, -1).contiguous()


def unpack_conv_weight(wp: torch.Tensor, cin: int, kh: int, kw: int) -> torch.Tensor:
    """Inverse of pack_conv_weight: [cout][cin/32][kh*kw][32] -> OIHW (used for weight gradients)."""
    co = wp.shape[0]
    return wp.reshape(co, cin // 32, kh * kw, 32).permute(0, 2, 1, 3).reshape(co, kh, kw, cin).permute(0, 3, 1, 2).contiguous()


def _bn_affine(bn: nn.BatchNorm2d, bias: Optional[torch.Tensor]):
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    shift = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
    if bias is not None:
        shift = shift + bias.detach().double() * scale
    return scale.float(), shift.float()


def _stack(ts: Sequence[Optional[torch.Tensor]]) -> Optional[torch.Tensor]:
    if ts[0] is None:
        return None
    return (torch.stack(list(ts), 0) if len(ts) > 1 else ts[0]).contiguous()


def pack_convs(convs: Sequence[nn.Conv2d], bns: Optional[Sequence[nn.BatchNorm2d]] = None) -> PConv:
    p = PConv()
    c0 = convs[0]
    p.cout, p.cin, p.kh, p.kw = c0.weight.shape
    p.stride, p.pad, p.groups = c0.stride[0], c0.padding[0], len(convs)
    ws, scs, shs = [], [], []
    for i, c in enumerate(convs):
        ws.append(_pad_rows(pack_conv_weight(c.weight.detach())))
        if bns is not None:
            sc, sh = _bn_affine(bns[i], c.bias)
            scs.append(_pad_vec(sc, p.cout))
            shs.append(_pad_vec(sh, p.cout))
        else:
            scs.append(None)
            shs.append(_pad_vec(c.bias.detach(), p.cout) if c.bias is not None else None)
    p.w, p.scale, p.shift = _w_operand(_stack(ws)), _stack(scs), _stack(shs)
    return p


def pack_convs_cin_slice(convs: Sequence[nn.Conv2d], c0: int, c1: int, with_bias: bool) -> PConv:
    """Input-channel slice [c0, c1) of 1x1 convs as its own conv (used to split a conv over a channel concat)."""
    p = PConv()
    k0 = convs[0]
    p.cout, p.cin, p.kh, p.kw = k0.weight.shape[0], c1 - c0, 1, 1
    p.stride, p.pad, p.groups = 1, 0, len(convs)
    p.w = _w_operand(_stack([_pad_rows(pack_conv_weight(c.weight.detach()[:, c0:c1])) for c in convs]))
    p.scale = None
    p.shift = _stack([_pad_vec(c.bias.detach(), p.cout) for c in convs]) if (with_bias and k0.bias is not None) else None
    return p


def pack_linears(pairs: Sequence) -> PConv:
    """pairs: (weight (N,K), bias (N,) | None) per group member."""
    p = PConv()
    w0 = pairs[0][0]
    p.cout, p.cin = w0.shape
    p.kh = p.kw = p.stride = 1
    p.pad, p.groups = 0, len(pairs)
    p.w = _w_operand(_stack([_pad_rows(w.detach().float()) for w, _ in pairs]))
    p.scale = None
    p.shift = _stack([_pad_vec(b.detach(), p.cout) if b is not None else None for _, b in pairs])
    return p


def pack_linear_mods(lins: Sequence[nn.Linear]) -> PConv:
    return pack_linears([(l.weight, l.bias) for l in lins])


def _rows(t: torch.Tensor) -> Img:
    """(rows, c) matrix as a batch of 1x1 images."""
    r, c = t.shape
    return Img(t.view(r, 1, 1, c))


# Lanes: independent forwards of ONE module in flight at the same time (runner.PipelinedForward replays two captured forwards on two
# streams so that the low-occupancy tail of one overlaps the convolutions of the other).  A lane has its own scratch (split-K /
# weight-stream workspace, abs-max arena) and shares the read-only packed weights with lane 0.  Thread-local; lane 0 by default.
_LANE = threading.local()


def lane() -> int:
    return getattr(_LANE, "v", 0)


class use_lane:
    def __init__(self, v: int):
        self.v, self.old = int(v), 0

    def __enter__(self):
        self.old = lane()
        _LANE.v = self.v

    def __exit__(self, *exc):
        _LANE.v = self.old


class State:
    """Per-module cache of packed weights + scratch."""

    def __init__(self, device, share: Optional["State"] = None):
        self.device = device
        self.packs: Dict[object, object] = share.packs if share is not None else {}
        self.workspace = torch.empty(_WORKSPACE_FLOATS, device=device, dtype=torch.float32)
        # abs-max records of the forward's activations (hip.AmaxArena): the pre-scales of the fp16-scheme launches come from them
        self.amax = hip.AmaxArena(device) if hip.policy().w_format == "f16x2" else None

    def begin_forward(self):
        if self.amax is not None:
            self.amax.begin()

    def fork(self, images: int):
        """A side branch for launches that do not depend on what the main stream does next (round 5): `with st.fork(n) as f: ...; f.join(t)`.
        Only small forwards under graph capture take it (FORK_MAX_IMAGES: their launches leave most of the chip idle, so independent
        branches - the down-sample conv of a residual block, the refiners' own-view projection stack - run side by side as parallel
        branches of the captured graph).  Otherwise the block runs inline."""
        small = images <= FORK_MAX_IMAGES
        if small and getattr(self, "side", None) is None and torch.cuda.is_available():
            # made at the first small forward - the warm-up in front of a capture - so that the stream and the 64-MB workspace do not
            # come out of a graph's private pool
            self.side = torch.cuda.Stream(device=self.device)
            self.ws_side = torch.empty(_WORKSPACE_FLOATS, device=self.device, dtype=torch.float32)
        return _Fork(self, small)

    def new_amax(self):
        return self.amax.new() if self.amax is not None else None

    def get(self, key, builder):
        k = key if isinstance(key, (str, tuple)) else id(key)
        v = self.packs.get(k)
        if v is None:
            with torch.no_grad():
                v = builder()
            self.packs[k] = v
        return v


# Independent branches on a side stream, while a hipGraph is being captured, for forwards of at most this many images (views x
# frames); 0 = never (EGR_FORK_MAX_IMAGES).  Measured as single-lane graph replays: batch 2 1.75 -> 1.71 ms, batch 8 2.72 -> 2.56,
# batch 16 3.53 -> 3.39, batch 32 5.65 -> 5.50; batch 1 unchanged (1.43); batch 64 with two lanes in flight 6890 -> 6600 frames/s -
# there the other lane already fills what the branch would.
FORK_MAX_IMAGES = int(os.environ.get("EGR_FORK_MAX_IMAGES", "64"))


class _Fork:
    """Context of State.fork: inside it the current stream is the state's side stream (which first waits for the main stream) and the
    split-K workspace is the side branch's own; join(tensors...) makes the main stream wait for the branch and hands its results over."""

    def __init__(self, st: "State", on: bool):
        # (graph capture only: eagerly the stream switches cost the host more than the overlap returns - batch 1: 2.07 -> 2.40 ms)
        self.st, self.on = st, on and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        self.ctx = None

    def __enter__(self):
        if not self.on:
            return self
        st = self.st
        self.main = torch.cuda.current_stream(st.device)
        st.side.wait_stream(self.main)
        self.ws_main, st.workspace = st.workspace, st.ws_side
        self.ctx = torch.cuda.stream(st.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
            self.st.workspace = self.ws_main
        return False

    def join(self, *tensors):
        """Call on the main stream, before the first launch there that reads the branch's results."""
        if not self.on:
            return
        self.main.wait_stream(self.st.side)
        for t in tensors:
            if isinstance(t, torch.Tensor):
                t.record_stream(self.main)      # allocated on the side stream, read (and later freed) under the main one


# Bumped whenever packed weights are created or dropped: a captured hipGraph holds raw pointers into the packs it was recorded
# with (runner.GraphedForward), so a replay is only valid while this number has not moved since the capture.
GENERATION = [0]


def _state(mod: nn.Module, device) -> State:
    st = mod.__dict__.get("_egr_state")
    if st is None or st.device != device:
        st = State(device)
        GENERATION[0] += 1
        mod.__dict__["_egr_state"] = st
        mod.__dict__.pop("_egr_state_lanes", None)
        if "_egr_hook" not in mod.__dict__:
            mod.__dict__["_egr_hook"] = mod.register_load_state_dict_post_hook(lambda m, _k: invalidate(m))
    ln = lane()
    if ln == 0:
        return st
    lanes = mod.__dict__.setdefault("_egr_state_lanes", {})
    s = lanes.get(ln)
    if s is None or s.device != device or s.packs is not st.packs:
        s = State(device, share=st)       # (no GENERATION bump: the packs recorded graphs point into are unchanged)
        lanes[ln] = s
    return s


def invalidate(mod: nn.Module):
    """Drop packed weights (call after mutating parameters in place; load_state_dict does it itself)."""
    mod.__dict__.pop("_egr_state_lanes", None)
    if mod.__dict__.pop("_egr_state", None) is not None:
        GENERATION[0] += 1


def set_policy(mod: nn.Module, pol: Optional["hip.LaunchPolicy"]) -> None:
    """Give `mod` (and everything below it) its own launch policy - None: follow the process default again.  Packs are dropped (they
    are made under a policy: weight format, fused-layer images); two modules with different policies coexist in one process, and
    a forward activates its module's policy for the calling thread only (hip.use_policy).  The inference entry points honour it; the
    training step (egorear_amd.train) follows the AMBIENT policy of the thread that drives it - wrap the step in hip.use_policy(p) to
    train under another one."""
    for m in mod.modules():
        if pol is None:
            m.__dict__.pop("_egr_policy", None)
        else:
            m.__dict__["_egr_policy"] = pol
        invalidate(m)


def _under_module_policy(fn):
    """Entry points run under their module's own policy when it has one (set_policy); nested engine calls inherit it."""
    import functools

    @functools.wraps(fn)
    def wrapped(mod, *a, **k):
        pol = mod.__dict__.get("_egr_policy")
        if pol is None:
            return fn(mod, *a, **k)
        with hip.use_policy(pol):
            return fn(mod, *a, **k)
    return wrapped


def as_rgb(img: torch.Tensor) -> torch.Tensor:
    """(B, V, H, W) grayscale frames -> (B, V, 3, H, W) with the plane repeated, as the reference's trunk does
    (models/backbones/resnet.py:44-46); 5-D input is returned unchanged."""
    if img.dim() == 4:
        B, V, H, W = img.shape
        return img.unsqueeze(2).expand(B, V, 3, H, W).contiguous()
    if img.dim() != 5:
        raise RuntimeError(f"egorear_amd: expected a (B, V, 3, H, W) or (B, V, H, W) image batch, got {tuple(img.shape)}")
    return img


def _check_input(img: torch.Tensor, mod: nn.Module):
    if not img.is_cuda:
        raise RuntimeError("egorear_amd: input is on %s; the hot path runs on a HIP device only (no CPU fallback)" % img.device)
    p = next(mod.parameters())
    if p.device != img.device:
        raise RuntimeError(f"egorear_amd: module parameters on {p.device}, input on {img.device}")
    if img.dtype != torch.float32:
        raise RuntimeError("egorear_amd: fp32 input expected (the reference runs precision: 32)")


# --------------------------------------------------------------------------- generic steps

def conv(st: State, x: Img, p: PConv, act=ACT_NONE, **kw) -> Optional[Img]:
    """x / out / res hold the images of all p.groups groups back to back (see hip.conv2d)."""
    return hip.conv2d(x, p.w, p.cout, p.kh, p.kw, p.stride, p.pad, scale=p.scale, shift=p.shift, act=act,
                      workspace=st.workspace, split_k=kw.pop("split_k", 0), groups=p.groups,
                      amax_out=None if "out_nchw" in kw else st.new_amax(), **kw)


def linear(st: State, x: torch.Tensor, p: PConv, act=ACT_NONE, **kw) -> torch.Tensor:
    """x (groups*rows, cin) -> (groups*rows, cout)."""
    out = conv(st, _rows(x), p, act, **kw)
    return out.t.view(x.shape[0], p.cout)


def _unpadded(m: nn.Conv2d) -> bool:
    """A padded 1x1 conv grows the image: conv1x1_chain evaluates pad 0 only, such a pair stays on the single launches."""
    return tuple(m.padding) == (0, 0) if not isinstance(m.padding, (str, int)) else m.padding in (0, "valid")


def run_stack(st: State, seqs: Sequence[nn.Sequential], x: Img, out: Optional[Img] = None,
              last_kw: Optional[dict] = None) -> Optional[Img]:
    """Execute tree.stack()s (one per group, identical structure): Conv2d(+ReLU) fuse into one launch; Upsample /
    MaxPool2d are their own kernels (no weights: they simply run over all groups' images).  `out` / `last_kw`
    apply to the final op."""
    mods = [list(s) for s in seqs]
    m0 = mods[0]
    i = 0
    pre_lo = None          # the conv in front of an Upsample, already evaluated by a chained launch
    while i < len(m0):
        m = m0[i]
        if isinstance(m, nn.Conv2d):
            act = ACT_RELU if (i + 1 < len(m0) and isinstance(m0[i + 1], nn.ReLU)) else ACT_NONE
            last = (i + (2 if act else 1)) >= len(m0)
            p = st.get(m, lambda i=i: pack_convs([g[i] for g in mods]))
            # Conv2d(1x1)[+ReLU] -> [Upsample ->] Conv2d(1x1)[+ReLU]: both convs in ONE launch, the intermediate stays in registers
            # (egr_conv1x1_chain_f32); with the Upsample in between the second conv commutes in front of it as below
            j = i + (2 if act else 1)
            up = j < len(m0) and isinstance(m0[j], nn.Upsample)
            k = j + 1 if up else j
            nxt = m0[k] if k < len(m0) else None
            if (m.kernel_size == (1, 1) and m.stride == (1, 1) and _unpadded(m) and isinstance(nxt, nn.Conv2d)
                    and nxt.kernel_size == (1, 1) and nxt.stride == (1, 1) and _unpadded(nxt)):
                relu2 = k + 1 < len(m0) and isinstance(m0[k + 1], nn.ReLU)
                end = k + (2 if relu2 else 1)
                fin = end >= len(m0)
                ok = (not up) or relu2                                         # ReLU(conv(up(x))) = ReLU(up(conv(x))) needs the ReLU
                if up and end + 1 == len(m0) and last_kw and "out_nchw" in last_kw:
                    ok = False                                                 # tail of a heat-map head: egr_up2_relu_head_f32's pattern below
                    # ... whose first two convs (256 -> 256 + ReLU, then 256 -> 128 commuted in front of the up-sampling) are one
                    # launch where the streamed chain covers them; the Upsample branch below then starts from its result
                    ph = st.get(nxt, lambda k=k: pack_convs([g[k] for g in mods]))
                    if relu2 and hip.chain_eligible(x, p.w, ph.w, p.cout, ph.cout, p.groups, p.scale, ph.scale):
                        pre_lo = hip.conv1x1_chain(x, p.w, ph.w, p.cout, ph.cout, shift1=p.shift, shift2=ph.shift, act1=act, act2=ACT_NONE,
                                                   groups=p.groups, amax_out=st.new_amax())
                        i = j
                        continue
                if fin and last_kw:
                    ok = False                                                 # (a placed / channel-major final output stays on the single launches)
                p2 = st.get(nxt, lambda k=k: pack_convs([g[k] for g in mods])) if ok else None
                if ok and hip.chain_eligible(x, p.w, p2.w, p.cout, p2.cout, p.groups, p.scale, p2.scale):
                    y = hip.conv1x1_chain(x, p.w, p2.w, p.cout, p2.cout, shift1=p.shift, shift2=p2.shift, act1=act,
                                          act2=ACT_NONE if up else (ACT_RELU if relu2 else ACT_NONE), groups=p.groups,
                                          out=out if (fin and not up) else None, amax_out=st.new_amax())
                    x = hip.upsample2x(y, out=out if fin else None, relu=True) if up else y
                    i = end
                    continue
            kw = dict(last_kw or {}) if last else {}
            if last and out is not None:
                kw["out"] = out
            x = conv(st, x, p, act, **kw)
            i += 2 if act else 1
        elif isinstance(m, nn.Upsample):
            # Upsample -> Conv2d(1x1) -> ReLU is evaluated as Conv2d(1x1) -> Upsample(+ReLU): the bias-carrying
            # 1x1 conv commutes with bilinear interpolation, and runs on a quarter of the pixels this way.
            nxt = m0[i + 1] if i + 1 < len(m0) else None
            if (isinstance(nxt, nn.Conv2d) and nxt.kernel_size == (1, 1) and nxt.stride == (1, 1) and i + 2 < len(m0)
                    and isinstance(m0[i + 2], nn.ReLU)):
                p = st.get(nxt, lambda i=i: pack_convs([g[i + 1] for g in mods]))
                lo, pre_lo = (pre_lo, None) if pre_lo is not None else (conv(st, x, p, ACT_NONE), None)
                fin = m0[i + 3] if i + 4 == len(m0) else None
                if (isinstance(fin, nn.Conv2d) and fin.kernel_size == (1, 1) and fin.stride == (1, 1) and fin.out_channels <= 16
                        and fin.in_channels <= 128 and fin.in_channels % 16 == 0 and last_kw is not None and "out_nchw" in last_kw and (2 * lo.h) % 8 == 0
                        and (2 * lo.w) % 32 == 0):
                    # tail of a heat-map head: up x2 + ReLU + the final narrow 1x1 conv in one pass, straight into the
                    # (B, V, 15, H, W) planes; the 128-channel full-resolution tensor in between is never written
                    wf = st.get((id(fin), "plain"), lambda i=i: (
                        torch.stack([g[i + 3].weight.detach().float().reshape(fin.out_channels, fin.in_channels) for g in mods]).contiguous(),
                        torch.stack([g[i + 3].bias.detach().float() for g in mods]).contiguous() if fin.bias is not None else None))
                    hip.up2_relu_head(lo, wf[0], wf[1], last_kw["out_nchw"], last_kw["ymap"], last_kw.get("gy", 0), groups=len(mods))
                    return None
                last = (i + 3) >= len(m0)
                x = hip.upsample2x(lo, out=out if last else None, relu=True)
                i += 3
            else:
                x = hip.upsample2x(x)
                i += 1
        elif isinstance(m, nn.MaxPool2d):
            k = m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
            s = m.stride if isinstance(m.stride, int) else m.stride[0]
            pd = m.padding if isinstance(m.padding, int) else m.padding[0]
            x = hip.maxpool(x, k, s, pd)
            i += 1
        else:
            raise RuntimeError(f"egorear_amd.run_stack: unexpected module {type(m).__name__}")
    return x


# --------------------------------------------------------------------------- backbone (a1-a3)

def _pack_stems(trunks) -> tuple:
    wps, scs, shs = [], [], []
    for t in trunks:
        conv1, bn = t.layer_s2[0], t.layer_s2[1]
        w = conv1.weight.detach().float().reshape(64, 147)
        wp = torch.zeros((64, 148), device=w.device, dtype=torch.float32)
        wp[:, :147] = w
        sc, sh = _bn_affine(bn, None)
        wps.append(wp)
        scs.append(sc)
        shs.append(sh)
    return torch.stack(wps).contiguous(), torch.stack(scs).contiguous(), torch.stack(shs).contiguous()


def _basic_block(st: State, blks, x: Img, out: Optional[Img] = None) -> Img:
    b0 = blks[0]
    identity = x
    f = None
    if b0.downsample is not None:
        pd = st.get(b0.downsample, lambda: pack_convs([b.downsample[0] for b in blks], [b.downsample[1] for b in blks]))
        f = st.fork(x.n)                 # (small forwards: the 1x1 / stride-2 conv of the shortcut runs beside conv1)
        with f:
            identity = conv(st, x, pd, ACT_NONE)
    p1 = st.get(b0.conv1, lambda: pack_convs([b.conv1 for b in blks], [b.bn1 for b in blks]))
    p2 = st.get(b0.conv2, lambda: pack_convs([b.conv2 for b in blks], [b.bn2 for b in blks]))
    y = conv(st, x, p1, ACT_RELU)
    if f is not None:
        f.join(identity.t)
    return conv(st, y, p2, ACT_RELU, res=identity, res_mode=RES_BEFORE_ACT, out=out)


# Stem conv + BatchNorm + ReLU + MaxPool2d(3, 2, 1) in one pass (egr_stem_conv7x7_pool_f32, bit-identical to the two kernels);
# EGR_STEM_POOL=0 keeps the two launches.
STEM_POOL = os.environ.get("EGR_STEM_POOL", "1") != "0"
# The stem on the bf16 matrix cores (egr_stem_conv7x7_x6_f32: split-bf16 operands like the other convolutions); EGR_STEM_X6=0: fp32 MFMA.
STEM_X6 = os.environ.get("EGR_STEM_X6", "1") != "0"


def run_backbone(st: State, encs, img: torch.Tensor, view0: int, nviews: int, feat_out: Img, s32_out: Optional[Img] = None):
    """ResNet-18 trunk + FPN (resnet.py:43-74,121-137) of len(encs) encoders in grouped launches: encoder g processes
    views [view0 + g*nviews, view0 + (g+1)*nviews).  Writes the stride-4 features into `feat_out`
    (G*nviews*B, 64, 64, 128) and, if given, the stride-32 features into `s32_out`; returns the pyramid [s4..s32]."""
    G = len(encs)
    trunks, necks = [e.backbone for e in encs], [e.neck for e in encs]
    t0, n0 = trunks[0], necks[0]
    wp, sc, sh = st.get(t0.layer_s2, lambda: _pack_stems(trunks))
    if STEM_X6 and img.shape[3] % 32 == 0 and img.shape[4] % 64 == 0:     # the split kernel's tile is 16 x 32 output pixels (fp32 kernel: 8 x 32)
        if hip.policy().w_format == "f16x2":       # the fp16 scheme (per-tile pre-scale of the input patch)
            wh2, wds = st.get((id(t0.layer_s2), "wh2"), lambda: hip.pack_stem_wh2(wp))
            x = hip.stem_x6(img, view0, nviews, wh2, sc, sh, groups=G, pool=STEM_POOL, w_descale=wds,
                            amax_out=st.new_amax() if STEM_POOL else None)
        else:
            w6 = st.get((id(t0.layer_s2), "w6"), lambda: hip.pack_stem_w6(wp))
            x = hip.stem_x6(img, view0, nviews, w6, sc, sh, groups=G, pool=STEM_POOL,     # layer_s2 (+ the max-pool of layer_s4 in the same pass)
                            amax_out=st.new_amax() if STEM_POOL else None)
    elif STEM_POOL:
        x = hip.stem_pool(img, view0, nviews, wp, sc, sh, groups=G)
    else:
        x = hip.stem(img, view0, nviews, wp, sc, sh, groups=G)
    if not STEM_POOL:
        x = hip.maxpool(x, 3, 2, 1)
    pyramid = []
    stages = [(t.layer_s4[1], t.layer_s8, t.layer_s16, t.layer_s32) for t in trunks]
    for si in range(4):
        nblk = len(stages[0][si])
        for bi in range(nblk):
            last = si == 3 and bi == nblk - 1
            x = _basic_block(st, [stages[g][si][bi] for g in range(G)], x, out=s32_out if last else None)
        pyramid.append(x)
    # FPN top-down (resnet.py:121-137).  The fuse conv over cat(lateral, up2(coarser)) is split by linearity:
    #   W . cat(a, up2(b)) + bias = W_a . a + bias + up2(W_b . b)
    # so the W_b half runs on the coarse grid (a quarter of the pixels) and enters the W_a conv as a pre-activation residual
    # that the epilogue interpolates itself (EGR_RES_UP2_BEFORE_ACT); neither the 256-wide concat buffer nor the upsampled
    # tensor is ever materialised.
    c = n0.out_channels
    lat = conv(st, pyramid[3], st.get(n0.lateral_convs[3], lambda: pack_convs([k.lateral_convs[3][0] for k in necks])), ACT_RELU)
    for i in (3, 2, 1):
        lo = pyramid[i - 1]
        pl = st.get(n0.lateral_convs[i - 1], lambda i=i: pack_convs([k.lateral_convs[i - 1][0] for k in necks]))
        fuse = n0.fuse_convs[i - 1]
        pa = st.get((id(fuse), "a"), lambda i=i: pack_convs_cin_slice([k.fuse_convs[i - 1][0] for k in necks], 0, c, True))
        pb = st.get((id(fuse), "b"), lambda i=i: pack_convs_cin_slice([k.fuse_convs[i - 1][0] for k in necks], c, 2 * c, False))
        coarse_lo = conv(st, lat, pb, ACT_NONE)          # stays at the coarse resolution: the fuse conv's epilogue upsamples it on the fly
        if hip.chain_eligible(lo, pl.w, pa.w, pl.cout, pa.cout, G, pl.scale, pa.scale):
            # lateral conv + ReLU -> fuse conv (+ up-sampled half, ReLU) in one launch: the lateral tensor is never written
            fused = hip.conv1x1_chain(lo, pl.w, pa.w, pl.cout, pa.cout, shift1=pl.shift, shift2=pa.shift, act1=ACT_RELU, act2=ACT_RELU,
                                      res=coarse_lo, res_mode=RES_UP2_BEFORE_ACT, groups=G, amax_out=st.new_amax())
        else:
            fine = conv(st, lo, pl, ACT_RELU)
            fused = conv(st, fine, pa, ACT_RELU, res=coarse_lo, res_mode=RES_UP2_BEFORE_ACT)
        lat = conv(st, fused, st.get(n0.fpn_convs[i - 1], lambda i=i: pack_convs([k.fpn_convs[i - 1][0] for k in necks])), ACT_RELU,
                   out=feat_out if i == 1 else None)
    return pyramid


def _vb_view(t: torch.Tensor, V: int, B: int) -> torch.Tensor:
    """(V*B, H, W, C) view-major NHWC buffer -> logical (B, V, C, H, W) view (no copy)."""
    n, h, w, c = t.shape
    return t.view(V, B, h, w, c).permute(1, 0, 4, 2, 3)


# --------------------------------------------------------------------------- EgoPoseFormerHeatmap API

def _heatmap_core(mod, img):
    st = _state(mod, img.device)
    st.begin_forward()
    B, V = img.shape[:2]
    H4, W4 = img.shape[3] // 4, img.shape[4] // 4
    feat = torch.empty((V * B, H4, W4, mod.encoder.neck.out_channels), device=img.device, dtype=torch.float32)
    pyr = run_backbone(st, [mod.encoder], img.contiguous(), 0, V, Img(feat))
    return st, feat, pyr


@_under_module_policy
def heatmap_backbone_api(mod, img):
    _check_input(img, mod)
    img = as_rgb(img)
    B, V = img.shape[:2]
    _, feat, pyr = _heatmap_core(mod, img)
    return _vb_view(feat, V, B), [_vb_view(p.t, V, B) for p in pyr]


@_under_module_policy
def heatmap_forward_api(mod, img, return_feat=False):
    """EgoPoseFormerHeatmap.forward (egoposeformer_heatmap.py:29-44)."""
    _check_input(img, mod)
    img = as_rgb(img)
    B, V = img.shape[:2]
    st, feat, pyr = _heatmap_core(mod, img)
    H4, W4 = feat.shape[1:3]
    hm = torch.empty((B, V, mod.num_heatmap, H4, W4), device=img.device, dtype=torch.float32)
    plane = mod.num_heatmap * H4 * W4
    conv(st, Img(feat), st.get(mod.conv_heatmap, lambda: pack_convs([mod.conv_heatmap])), ACT_NONE, out_nchw=hm,
         ymap=NMap(B, V * plane, plane))
    if return_feat:
        return hm, _vb_view(feat, V, B), [_vb_view(p.t, V, B) for p in pyr]
    return hm


# --------------------------------------------------------------------------- deformable-attention layer (a12-a15, a24)

class PLayer:
    __slots__ = ("offs_logits", "head_w", "head_shift", "head_w_all", "head_shift_all", "pos_proj", "out_proj", "fuse", "ln_cross", "qkv", "mha_out",
                 "ln_spatial", "ffn0", "ffn1", "ln_ffn", "heads", "dh", "C", "groups", "fused", "ol_plain")


# One launch per transformer layer behind the sampling (egr_joint_layer_f32) instead of ~14 small ones; EGR_FUSED_LAYER=0 keeps
# the per-op launches (the training forward always uses those: it needs the intermediates).
# (hip.LaunchPolicy.fused_layer)
# The small launches in front of / behind the fused layers as one launch each (round 5): the refiners' JQA query
# (egr_jqa_query_f32), the lifting head's proposal -> reprojection -> decoder query (egr_pose_query_f32), the refiners' head offset as
# a tail of the layer launch; EGR_FUSED_QUERY=0 keeps the per-op launches.
# (hip.LaunchPolicy.fused_query)
# the fused layer's contractions in the fp16 scheme (read when a module's layers are packed): follows EGR_W_FORMAT, EGR_LAYER_H2=0 keeps
# the fp32 matrix cores
# (hip.LaunchPolicy.layer_h2)


def pack_layers(layers, pres, poss) -> PLayer:
    """Pack `len(layers)` same-shape transformer layers as one group.  Per layer, fold the linear chain in front of the
    sampling so that sampling can come first:
        value = W_v (W_pre f + b_pre [+ pos]) + b_v
      sampled & weighted:  W_v W_pre g + (W_v b_pre + b_v) sigma + W_v e
    (g, e, sigma from egr_msda_gather_f32).  W_pre/b_pre is the 1x1 conv in front of the attention
    (frame_feat_multi_view_proj for MVFEx, feat_proj for the lifting head).  Folded in fp64, stored fp32."""
    P = PLayer()
    ca0 = layers[0].cross_attn
    P.heads, P.C, P.groups = ca0.n_heads, ca0.d_model, len(layers)
    P.dh = P.C // P.heads
    head_w = [[] for _ in range(P.heads)]
    head_shift = [[] for _ in range(P.heads)]
    pos_proj = []
    for layer, (pre_w, pre_b), pos in zip(layers, pres, poss):
        ca = layer.cross_attn
        Wv, bv = ca.value_proj.weight.detach().double(), ca.value_proj.bias.detach().double()
        Wp, bp = pre_w.detach().double().reshape(pre_w.shape[0], -1), pre_b.detach().double()
        Wfold = (Wv @ Wp).float()                      # (C, cf)
        cfold = (Wv @ bp + bv).float()                 # (C,)
        for h in range(P.heads):
            head_w[h].append(_pad_rows(Wfold[h * P.dh:(h + 1) * P.dh].contiguous()))
            head_shift[h].append(_pad_vec(cfold[h * P.dh:(h + 1) * P.dh], P.dh))
        if pos is not None:                            # (1, V, HW, C) -> (V, HW, C) projected by W_v
            pos_proj.append((pos.detach()[0].double() @ Wv.t()).float())
    P.head_w = [_w_operand(_stack(w)) for w in head_w]
    P.head_shift = [_stack(s) for s in head_shift]
    # one query set (the lifting head): the heads themselves become the groups of a single launch
    P.head_w_all = _w_operand(torch.stack([w[0] for w in head_w]).contiguous()) if P.groups == 1 else None
    P.head_shift_all = torch.stack([s[0] for s in head_shift]).contiguous() if P.groups == 1 else None
    P.pos_proj = torch.stack(pos_proj).contiguous() if pos_proj else None
    cas = [l.cross_attn for l in layers]
    P.offs_logits = pack_linears([(torch.cat([c.sampling_offsets.weight.detach(), c.attention_weights.weight.detach()], 0),
                                   torch.cat([c.sampling_offsets.bias.detach(), c.attention_weights.bias.detach()], 0)) for c in cas])
    P.out_proj = pack_linear_mods([c.output_proj for c in cas])
    P.fuse = pack_linear_mods([l.fuse_mlp for l in layers])
    sas = [l.spatial_attn for l in layers]
    P.qkv = pack_linears([(torch.cat([s.q_proj.weight.detach(), s.k_proj.weight.detach(), s.v_proj.weight.detach()], 0),
                           torch.cat([s.q_proj.bias.detach(), s.k_proj.bias.detach(), s.v_proj.bias.detach()], 0)) for s in sas])
    P.mha_out = pack_linear_mods([s.out_proj for s in sas])
    P.ffn0 = pack_linear_mods([l.ffn.layers[0][0] for l in layers])
    P.ffn1 = pack_linear_mods([l.ffn.layers[1] for l in layers])

    def ln(name):
        return (torch.cat([getattr(l, name).weight.detach().float() for l in layers]).contiguous(),
                torch.cat([getattr(l, name).bias.detach().float() for l in layers]).contiguous())
    P.ln_cross, P.ln_spatial, P.ln_ffn = ln("norm_cross"), ln("norm_spatial"), ln("norm_ffn")
    # ---- the same parameters as plain (groups, out, in) stacks for the fused layer kernel
    def stk(ts):
        return torch.stack([t.detach().float() for t in ts]).contiguous()
    wf, cf_ = [], []
    for layer, (pre_w, pre_b), _ in zip(layers, pres, poss):
        ca = layer.cross_attn
        Wv, bv = ca.value_proj.weight.detach().double(), ca.value_proj.bias.detach().double()
        Wp, bp = pre_w.detach().double().reshape(pre_w.shape[0], -1), pre_b.detach().double()
        wf.append((Wv @ Wp).float())
        cf_.append((Wv @ bp + bv).float())
    sas_ = [l.spatial_attn for l in layers]
    P.fused = {
        "w_fold": stk(wf), "c_fold": stk(cf_),
        "w_out": stk([c.output_proj.weight for c in cas]), "b_out": stk([c.output_proj.bias for c in cas]),
        "w_fuse": stk([l.fuse_mlp.weight for l in layers]), "b_fuse": stk([l.fuse_mlp.bias for l in layers]),
        "ln1_g": P.ln_cross[0], "ln1_b": P.ln_cross[1],
        "w_qkv": stk([torch.cat([s_.q_proj.weight, s_.k_proj.weight, s_.v_proj.weight], 0) for s_ in sas_]),
        "b_qkv": stk([torch.cat([s_.q_proj.bias, s_.k_proj.bias, s_.v_proj.bias], 0) for s_ in sas_]),
        "w_mo": stk([s_.out_proj.weight for s_ in sas_]), "b_mo": stk([s_.out_proj.bias for s_ in sas_]),
        "ln2_g": P.ln_spatial[0], "ln2_b": P.ln_spatial[1],
        "w_f0": stk([l.ffn.layers[0][0].weight for l in layers]), "b_f0": stk([l.ffn.layers[0][0].bias for l in layers]),
        "w_f1": stk([l.ffn.layers[1].weight for l in layers]), "b_f1": stk([l.ffn.layers[1].bias for l in layers]),
        "ln3_g": P.ln_ffn[0], "ln3_b": P.ln_ffn[1],
    }
    P.ol_plain = {"w": stk([torch.cat([c.sampling_offsets.weight, c.attention_weights.weight], 0) for c in cas]),
                  "b": stk([torch.cat([c.sampling_offsets.bias, c.attention_weights.bias], 0) for c in cas])}
    # the fused kernel covers the shipped shapes (4 heads, 128-channel memory, FFN 512, C 128 / 256, 4 views); any other layer keeps
    # the per-op launches of run_layer (which handle every cin % 32 == 0) instead of failing at pack time
    ffn = layers[0].ffn.layers[0][0].weight.shape[0]
    ok = (P.heads == 4 and P.C in (128, 256) and ffn == 512 and P.fused["w_fold"].shape[-1] == 128
          and P.fused["w_fuse"].shape[-1] == 4 * P.C and P.ol_plain["w"].shape[-2] % 16 == 0)
    if not ok:
        P.fused = P.ol_plain = None
        return P
    # the kernel streams its weight matrices as 1-KiB contiguous wave loads: fragment order, flagged in the dict - the fp16 scheme's
    # images (hip.pack_layer_wh2: the layer's contractions then run like the conv launches') or, with
    # EGR_W_FORMAT=bf16x3 / EGR_LAYER_H2=0, the fp32 matrices (hip.pack_layer_w: fp32 matrix cores, the reference's arithmetic class)
    pk = hip.pack_layer_wh2 if hip.policy().layer_h2 else hip.pack_layer_w
    for k in ("w_fold", "w_out", "w_fuse", "w_qkv", "w_mo", "w_f0", "w_f1"):
        P.fused[k] = pk(P.fused[k])
    P.fused["packed"] = 2 if hip.policy().layer_h2 else True
    P.ol_plain["w"] = pk(P.ol_plain["w"])
    return P


def run_layer_fused(st: State, P: PLayer, x: torch.Tensor, memory: torch.Tensor, anchors: torch.Tensor, valid: torch.Tensor,
                    B: int, V: int, J: int, hgt: int, wid: int, ol: Optional[torch.Tensor] = None, next_P: Optional[PLayer] = None,
                    post=None, reg=None, want_xn: bool = False, head=None):
    """run_layer as TWO launches: the sampling (egr_msda_gather_f32) and everything behind it (egr_joint_layer_f32), optionally
    with the next layer's offsets / logits, post_norm and the regression head as tails.  `ol`: this layer's offsets / logits if a
    previous fused layer already produced them.  Returns (x, ol_next | None, xn | None, pred | None)."""
    G, C, heads, dh = P.groups, P.C, P.heads, P.dh
    if ol is None:
        ol = linear(st, x, P.offs_logits)
    g, e, sigma, rowmask = hip.msda_gather(memory, P.pos_proj, ol, anchors, valid, B, V, J, heads, dh, hgt, wid, groups=G)
    return hip.joint_layer(x, g, e, sigma, rowmask, P.fused, B, J, V, C, G, ol=next_P.ol_plain if next_P is not None else None,
                           post=post, reg=reg, want_xn=want_xn, head=head)


def run_layer(st: State, P: PLayer, x: torch.Tensor, memory: torch.Tensor, anchors: torch.Tensor, valid: torch.Tensor,
              B: int, V: int, J: int, hgt: int, wid: int) -> torch.Tensor:
    """One MultiViewTransformerLayer / EgoPoseFormerTransformerLayer (heatmap_mvf_ex.py:874-935,
    egoposeformer_mvf_ex.py:546-588) for P.groups query sets at once.  x (G*B*J, C); memory (V, B, hgt*wid, cf)
    *un-projected* features shared by all groups; anchors / valid shared."""
    G, C, heads, dh = P.groups, P.C, P.heads, P.dh
    ol = linear(st, x, P.offs_logits)                                   # (G*B*J, heads*16*3); shared by all views
    g, e, sigma, rowmask = hip.msda_gather(memory, P.pos_proj, ol, anchors, valid, B, V, J, heads, dh, hgt, wid, groups=G)
    rows = B * J * V
    cf = g.shape[-1]
    a = torch.empty((G * rows, C), device=x.device, dtype=torch.float32)
    g2 = g.view(G * rows, heads * cf)
    e2 = e.view(G * rows, C) if e is not None else None
    sig = sigma.view(G * heads * rows)
    if G == 1 and P.head_w_all is not None:                             # heads as groups: one launch for the whole projection
        hip.conv2d(_rows(g2[:rows, :cf]), P.head_w_all, dh, 1, 1, 1, 0, shift=P.head_shift_all, rowscale=sig, grs=rows,
                   res=_rows(e2[:rows, :dh]) if e2 is not None else None, res_mode=RES_AFTER_ACT if e2 is not None else RES_NONE,
                   out=_rows(a[:rows, :dh]), workspace=None, split_k=1, groups=heads, gx=cf, gr=dh, gy=dh)
    for h in range(heads if not (G == 1 and P.head_w_all is not None) else 0):   # per-head folded value projection, all groups
        hip.conv2d(_rows(g2[:rows, h * cf:(h + 1) * cf]), P.head_w[h], dh, 1, 1, 1, 0, shift=P.head_shift[h],
                   rowscale=sig[h * rows:], grs=heads * rows,
                   res=_rows(e2[:rows, h * dh:(h + 1) * dh]) if e2 is not None else None,
                   res_mode=RES_AFTER_ACT if e2 is not None else RES_NONE,
                   out=_rows(a[:rows, h * dh:(h + 1) * dh]), workspace=None, split_k=1,
                   groups=G, gx=rows * heads * cf, gr=rows * C, gy=rows * C)
    o = linear(st, a, P.out_proj, rowmask=rowmask, grm=0)               # masked_fill(~valid) after output_proj
    f = linear(st, o.view(G * B * J, V * C), P.fuse)                    # cat over views is the row layout already
    x = hip.layernorm(f, P.ln_cross[0], P.ln_cross[1], res=x, groups=G)
    qkv = linear(st, x, P.qkv)
    att = hip.joint_mha(qkv, G * B, J, heads, dh, dh ** -0.5)
    x = hip.layernorm(linear(st, att, P.mha_out), P.ln_spatial[0], P.ln_spatial[1], res=x, groups=G)
    h1 = linear(st, x, P.ffn0, ACT_GELU)
    x = hip.layernorm(linear(st, h1, P.ffn1), P.ln_ffn[0], P.ln_ffn[1], res=x, groups=G)
    return x


# --------------------------------------------------------------------------- EgoPoseFormerHeatmapMVFEX (a5-a18)

class PRefiners:
    __slots__ = ("hp0", "hp2", "fc_bfb", "fc_query", "embed", "layer", "post_norm", "head0_w", "head0_b", "head3", "q")


def _pack_refiners(rs) -> PRefiners:
    P = PRefiners()
    P.hp0 = pack_linear_mods([r.heatmap_proj[0] for r in rs])
    P.hp2 = pack_linear_mods([r.heatmap_proj[2] for r in rs])
    P.fc_bfb = pack_linear_mods([r.fc_bfb for r in rs])
    P.fc_query = pack_linear_mods([r.fc_query[0] for r in rs])
    P.embed = torch.stack([r.joint_query_embed.weight.detach().float() for r in rs]).contiguous()
    P.layer = pack_layers([r.transformer_layers[0] for r in rs],
                          [(r.frame_feat_multi_view_proj.weight, r.frame_feat_multi_view_proj.bias) for r in rs],
                          [r.frame_feat_multi_view_pos_embed for r in rs])
    P.post_norm = (torch.cat([r.post_norm[0].weight.detach().float() for r in rs]).contiguous(),
                   torch.cat([r.post_norm[0].bias.detach().float() for r in rs]).contiguous())
    P.head0_w = torch.stack([r.head_layers[0].head[0].weight.detach().float().reshape(r.head_layers[0].head[0].weight.shape[0], -1)
                             for r in rs]).contiguous()                                   # (G, 64, 15)
    P.head0_b = torch.stack([r.head_layers[0].head[0].bias.detach().float() for r in rs]).contiguous()
    P.head3 = pack_convs([r.head_layers[0].head[3] for r in rs])
    # egr_jqa_query_f32's operands: the matrices in the fused layer's weight order (it shares the layer's offsets / logits image)
    P.q = None
    r0 = rs[0]
    if (P.layer.fused is not None and r0.embed_dims == 256 and r0.fc_bfb.in_features == 512 and r0.num_heatmap <= 16):
        pk = hip.pack_layer_wh2 if hip.policy().layer_h2 else hip.pack_layer_w

        def stk(ts):
            return torch.stack([t.detach().float() for t in ts]).contiguous()
        P.q = {"w_hp2": pk(stk([r.heatmap_proj[2].weight for r in rs])), "b_hp2": stk([r.heatmap_proj[2].bias for r in rs]),
               "w_bfb": pk(stk([r.fc_bfb.weight for r in rs])), "b_bfb": stk([r.fc_bfb.bias for r in rs]),
               "embed": P.embed,
               "w_q": pk(stk([r.fc_query[0].weight for r in rs])), "b_q": stk([r.fc_query[0].bias for r in rs]),
               "w_ol": P.layer.ol_plain["w"], "b_ol": P.layer.ol_plain["b"], "packed": 2 if hip.policy().layer_h2 else True}
    return P


# tests only: a dict here receives the refiners' intermediates of the next forward - "query" (G, B, J, C) behind fc_query, "post_norm"
# (G, B, J, C) behind the transformer layer + post_norm, "head_sum" (G*B, h, w, C) = head offset + own-view projection (NHWC) - the
# tensors tests/golden/mvfex_mid_s*.npz pin against the reference (heatmap_mvf_ex.py:665, 699-706, 715)
CAPTURE = None


def _run_refiners(st: State, rs, B: int, V: int, hm_init: torch.Tensor, feat_all: torch.Tensor, s32_all: torch.Tensor,
                  anchors, valid, feat_ref: torch.Tensor, hm_ref: torch.Tensor):
    """The four HeatmapMVF refiners (heatmap_mvf_ex.py:652-731), refiner g = view g, as one group of G = V."""
    G = len(rs)
    assert G == V
    r0 = rs[0]
    P: PRefiners = st.get(r0, lambda: _pack_refiners(rs))
    J, C = r0.num_heatmap, r0.embed_dims
    hgt, wid = r0.feat_shape
    hw = hgt * wid
    # --- own-view feature projection: group g reads feat_all[g*B:(g+1)*B].  It depends on the features only and is first read behind the
    # transformer layer: small forwards run it on the side branch, next to the query / sampling / layer launches
    fk = st.fork(G * B)
    with fk:
        ff = run_stack(st, [r.frame_feat_proj_layers for r in rs], Img(feat_all))  # (G*B, 32, 32, 128)
    # --- joint queries (JQA): heatmap_proj(hm) + fc_bfb(avgpool s32) + embedding -> fc_query
    hm_rows = Img(hm_init.view(B * V, J, 1, hw)[0::V])                        # group 0 = view 0 rows; group stride = J*hw
    t = conv(st, hm_rows, P.hp0, ACT_RELU, gx=J * hw)                        # (G*B, J, 1, C)
    pol = hip.policy()
    fused = pol.fused_layer and P.layer.fused is not None
    ol = None
    if fused and pol.fused_query and P.q is not None and s32_all.shape[-1] == 512:
        # heatmap_proj[2], the pooled fc_bfb, the sum, fc_query and the layer's offsets / logits: one launch
        x, ol = hip.jqa_query(t.t.view(G * B * J, C), s32_all, P.q, B, J, C, G)
    else:
        hm_embed = linear(st, t.t.view(G * B * J, C), P.hp2)
        bfb = linear(st, hip.avgpool(Img(s32_all)), P.fc_bfb)                # s32_all is (V*B, 8, 8, 512) = (G, B, ...)
        x = linear(st, hip.jqa_sum(hm_embed, P.embed, bfb, G * B, J, C, groups=G), P.fc_query, ACT_RELU)
    # --- transformer layer over the 4-view memory (sampled un-projected, see pack_layers)
    # --- head: LN -> (B, J, 16, 16) image with joints as channels -> 1x1 15->64, up x2, 1x1 64->128 (+ frame_feat)
    head = None
    if fused:
        # (the head offset rides on the layer launch when its shape is the kernel's: 16 x 16 tokens-as-image, 64 channels)
        if pol.fused_query and C == 256 and P.head0_w.shape[1] == 64:
            head = {"w": P.head0_w, "b": P.head0_b, "amax": st.new_amax()}
        _, _, xn, _ = run_layer_fused(st, P.layer, x, feat_all.view(V, B, hw, feat_all.shape[-1]), anchors, valid, B, V, J, hgt, wid, ol=ol,
                                      post={"g": P.post_norm[0], "b": P.post_norm[1]}, want_xn=head is None or CAPTURE is not None, head=head)
    else:
        x = run_layer(st, P.layer, x, feat_all.view(V, B, hw, feat_all.shape[-1]), anchors, valid, B, V, J, hgt, wid)
        xn = hip.layernorm(x, P.post_norm[0], P.post_norm[1], groups=G)
    if head is not None:
        h0 = Img(head["out"])
    else:
        side = int(math.isqrt(C))
        tok = hip.tokens_to_nhwc(xn, G * B, J, C)                            # (G*B, 256, J)
        h0 = hip.linear_smallk(tok, J, 1, P.head0_w, P.head0_b, G * B * C, P.head0_w.shape[1], J, ACT_RELU, groups=G)
        h0 = hip.upsample2x(Img(h0.view(G * B, side, side, -1)))
    fk.join(ff.t)
    summed = conv(st, h0, P.head3, ACT_RELU, res=ff, res_mode=RES_AFTER_ACT)   # offset_pred + frame_feat
    if CAPTURE is not None:
        CAPTURE.update(query=x.view(G, B, J, C).clone(), post_norm=xn.view(G, B, J, C).clone(), head_sum=summed.t.clone())
    run_stack(st, [r.frame_feat_refined_proj_layers[0] for r in rs], summed, out=Img(feat_ref))
    # --- refined heatmaps, written as (B, V, 15, 64, 64) planes; group g = view g
    plane = J * hw
    run_stack(st, [r.conv_heatmap_layers[0] for r in rs], Img(feat_ref),
              last_kw={"out_nchw": hm_ref, "ymap": NMap(B, V * plane, 0), "gy": plane})


def _anchors(hm: torch.Tensor, thr: float):
    B, V, J = hm.shape[:3]
    anchors, maxvals, valid, index = hip.argmax_rows(hm, thr)
    return anchors.view(B, V, J, 2), maxvals.view(B, V, J), valid.view(B, V, J), index.view(B, V, J)


@_under_module_policy
def anchors_from_heatmap_api(mod, heatmap):
    """get_anchors_2d_from_hm (heatmap_mvf_ex.py:128-143) -> (pts2d, maxvals, mask_valid[bool])."""
    a, m, v, _ = _anchors(heatmap.contiguous(), mod.heatmap_threshold)
    return a, m, v.bool()


def _mvfex(mod, img: torch.Tensor, heatmap_for_anchor=None):
    st = _state(mod, img.device)
    st.begin_forward()
    B, V = img.shape[:2]
    img = img.contiguous()
    H4, W4 = img.shape[3] // 4, img.shape[4] // 4
    J = mod.num_heatmap
    dev = img.device
    feat_all = torch.empty((V * B, H4, W4, 128), device=dev, dtype=torch.float32)
    s32_all = torch.empty((V * B, H4 // 8, W4 // 8, 512), device=dev, dtype=torch.float32)
    front, back = mod.heatmap_estimator_stereo_front, mod.heatmap_estimator_stereo_back
    run_backbone(st, [front.encoder, back.encoder], img, 0, 2, Img(feat_all), Img(s32_all))   # G = 2 stereo estimators
    # --- initial heatmaps per stereo pair (G = 2), written straight into (B, V, J, H4, W4)
    hm_init = torch.empty((B, V, J, H4, W4), device=dev, dtype=torch.float32)
    plane = J * H4 * W4
    run_stack(st, [mod.conv_heatmap_layers_stereo_front, mod.conv_heatmap_layers_stereo_back], Img(feat_all),
              last_kw={"out_nchw": hm_init, "ymap": NMap(B, V * plane, plane), "gy": 2 * plane})
    src = heatmap_for_anchor.contiguous() if isinstance(heatmap_for_anchor, torch.Tensor) else hm_init
    anchors, maxvals, valid, index = _anchors(src, mod.heatmap_threshold)
    # --- four refiners (own weights each, G = 4), every one attending to all four views
    feat_ref = torch.empty_like(feat_all)
    hm_ref = torch.empty_like(hm_init)
    _run_refiners(st, mod.refiners(), B, V, hm_init, feat_all, s32_all, anchors, valid, feat_ref, hm_ref)
    aux = {"anchors_2d": anchors, "maxvals": maxvals, "anchors_valid": valid, "argmax_idx": index}
    return hm_init, hm_ref, feat_all, feat_ref, aux


@_under_module_policy
def heatmap_mvfex_forward_api(mod, img, heatmap_for_anchor=None):
    """EgoPoseFormerHeatmapMVFEX.forward -> ([hm_init, hm_refined], [feat_init, feat_refined])."""
    _check_input(img, mod)
    img = as_rgb(img)
    B, V = img.shape[:2]
    hm_init, hm_ref, feat_all, feat_ref, aux = _mvfex(mod, img, heatmap_for_anchor)
    mod.__dict__["_egr_last_aux"] = aux
    return [hm_init, hm_ref], [_vb_view(feat_all, V, B), _vb_view(feat_ref, V, B)]


# --------------------------------------------------------------------------- EgoPoseFormerPose3D (a20-a24)

class PPose:
    __slots__ = ("mlp0", "mlp0_ws", "mlp0_src", "mlp1", "mlp2", "qg0_w", "qg0_b", "qg2", "qg4", "layers", "post", "reg0", "reg2", "cams", "reg_plain",
                 "q")


def _pack_pose3d(p3) -> PPose:
    P = PPose()
    V = p3.num_views
    w0 = p3.mlp_pred[0][0].weight.detach()
    n_out = w0.shape[0]
    # reference flattens "(b v) c h w -> b (v c h w)" (egoposeformer_mvf_ex.py:317); ours is (v, h, w, c)
    w0p = w0.view(n_out, V, 128, 8, 8).permute(0, 1, 3, 4, 2).reshape(n_out, -1)
    P.mlp0 = P.mlp0_ws = None
    P.mlp0_src = (w0p, p3.mlp_pred[0][0].bias)
    if hip.policy().w_format == "f16x2" and n_out % 64 == 0 and w0p.shape[1] % 256 == 0:
        # the weight-stream launch (egr_linear_wstream_f32): two fp16 planes at the same 4 bytes per weight
        img, ds = hip.pack_wstream(w0p.float().contiguous())
        P.mlp0_ws = (img, ds, p3.mlp_pred[0][0].bias.detach().float().contiguous())
    else:
        P.mlp0 = pack_linears([(w0p, p3.mlp_pred[0][0].bias)])
    P.mlp1 = pack_linear_mods([p3.mlp_pred[1][0]])
    P.mlp2 = pack_linear_mods([p3.mlp_pred[2]])
    qg = p3.query_gen_mlp
    P.qg0_w, P.qg0_b = qg[0].weight.detach().float().contiguous(), qg[0].bias.detach().float().contiguous()
    P.qg2, P.qg4 = pack_linear_mods([qg[2]]), pack_linear_mods([qg[4]])
    fp = p3.feat_proj
    P.layers = [pack_layers([l], [(fp.weight, fp.bias)], [None]) for l in p3.layers]

    def f32(t):
        return t.detach().float().contiguous()
    P.post = [(f32(n.weight), f32(n.bias)) for n in p3.post_norm]
    P.reg0 = [pack_linear_mods([r[0]]) for r in p3.reg_mlp]
    P.reg2 = [pack_linear_mods([r[2]]) for r in p3.reg_mlp]
    # the fused layer's regression tail (w0 in fragment order); None when a layer is outside the fused kernel's shapes (pack_layers)
    P.reg_plain = None
    if all(L.fused is not None for L in P.layers):
        pk = hip.pack_layer_wh2 if hip.policy().layer_h2 else hip.pack_layer_w
        P.reg_plain = [(pk(f32(r[0].weight).contiguous()), f32(r[0].bias), f32(r[2].weight), f32(r[2].bias)) for r in p3.reg_mlp]
    rec = np.stack([c.packed() for c in p3.cameras()])
    P.cams = torch.from_numpy(rec).to(w0.device)
    # egr_pose_query_f32's operands (mlp_pred[2], query_gen_mlp, the first layer's offsets / logits) in the fused layers' weight order
    P.q = None
    if (P.reg_plain is not None and p3.embed_dims == 128 and p3.num_joints == 16 and p3.mlp_pred[2].in_features == 128
            and qg[0].in_features == 4 and P.layers[0].ol_plain is not None):
        pk = hip.pack_layer_wh2 if hip.policy().layer_h2 else hip.pack_layer_w
        P.q = {"w_m2": pk(f32(p3.mlp_pred[2].weight)), "b_m2": f32(p3.mlp_pred[2].bias), "w_qg0": P.qg0_w, "b_qg0": P.qg0_b,
               "w_qg2": pk(f32(qg[2].weight)), "b_qg2": f32(qg[2].bias), "w_qg4": pk(f32(qg[4].weight)), "b_qg4": f32(qg[4].bias),
               "w_ol": P.layers[0].ol_plain["w"], "b_ol": P.layers[0].ol_plain["b"], "packed": 2 if hip.policy().layer_h2 else True}
    return P


def _pose3d(p3, st: State, feat_init: torch.Tensor, feat_final: torch.Tensor, B: int, V: int, ctm, behind: Optional[State] = None):
    """EgoPoseFormerPose3D.forward (egoposeformer_mvf_ex.py:422-452).  feat_*: (V*B, 64, 64, 128) view-major NHWC.
    behind: the state of the heat-map estimator whose forward has just produced feat_* - the lifting head then goes on in ITS abs-max
    arena (one clear per forward instead of two; the records of feat_* sit in it anyway)."""
    own = st.amax
    if behind is not None and behind.amax is not None and own is not None:
        st.amax = behind.amax          # for THIS call only (restored below): a later stand-alone call clears its own arena, not the estimator's
    else:
        st.begin_forward()
    try:
        return _pose3d_body(p3, st, feat_init, feat_final, B, V, ctm)
    finally:
        st.amax = own


def _pose3d_body(p3, st: State, feat_init: torch.Tensor, feat_final: torch.Tensor, B: int, V: int, ctm):
    P: PPose = st.get(p3, lambda: _pack_pose3d(p3))
    for t in (feat_init, feat_final):      # feature maps handed in by a caller carry no abs-max record: make one (one read each)
        if st.amax is not None and getattr(t, "_egr_amax", None) is None:
            rec = st.new_amax()
            if rec is not None:
                hip.absmax_record(t, rec)
    dev = feat_init.device
    J = p3.num_joints
    hgt, wid = p3.feat_shape
    src = feat_init if p3.use_pred_heatmap_init else feat_final
    # --- proposal: conv stack on the refined features, flattened per frame, 3-layer MLP (_forward_mlp_conv)
    flat = torch.empty((B, V * 8 * 8 * 128), device=dev, dtype=torch.float32)
    fo = run_stack(st, [p3.conv_frame_feat], Img(feat_final),
                   out=Img(flat.view(B * V, 8, 8, 128)), last_kw={"ymap": NMap(B, V * 8192, 8192)})  # (v,b) -> (b,v)
    if P.mlp0_ws is not None and fo.amax is not None:
        flat._egr_amax = fo.amax
        h = hip.linear_wstream(flat, P.mlp0_ws[0], P.mlp0_ws[1], P.mlp0_ws[2], ACT_GELU, st.workspace, amax_out=st.new_amax())
    else:
        if P.mlp0 is None:       # (rows without a record: the 4-byte split-K launch)
            with torch.no_grad():
                P.mlp0 = pack_linears([P.mlp0_src])
        h = linear(st, flat, P.mlp0, ACT_GELU)
    h = linear(st, h, P.mlp1, ACT_GELU)
    ctm32 = None
    if p3.camera_model.startswith("ego4view_rw"):
        if ctm is None:
            raise RuntimeError("egorear_amd: camera_model ego4view_rw needs coord_trans_mat (B,4,4,4)")
        ctm32 = ctm.to(device=dev, dtype=torch.float32).contiguous()         # any float dtype accepted
    C = p3.embed_dims
    ol = None
    pol = hip.policy()
    if pol.fused_layer and pol.fused_query and P.q is not None and all(L.fused is not None for L in P.layers):
        # mlp_pred[2], the reprojection, query_gen_mlp and the first layer's offsets / logits: one launch
        mlp_pred, anchors_3d, anchors_2d, valid, x, ol = hip.pose_query(h, ctm32, P.cams, P.q, B, J, C)
    else:
        mlp_pred = linear(st, h, P.mlp2).view(B, J, 3)
        anchors_3d = torch.empty_like(mlp_pred)                              # init_anchors_3d = mlp_pred.clone().detach(): written by the projection kernel
        anchors_2d, valid, q4 = hip.fisheye_project(mlp_pred, ctm32, P.cams, out=anchors_3d)   # syn: anchors_3d = the points after the in-place chain (F7)
        # --- decoder
        x = hip.linear_smallk(q4, 4, 1, P.qg0_w, P.qg0_b, B * J, C, 4, ACT_RELU)
        x = linear(st, x, P.qg2, ACT_RELU)
        x = linear(st, x, P.qg4)
    memory = src.view(V, B, hgt * wid, src.shape[-1])
    preds = [mlp_pred]
    a3 = anchors_3d.view(B * J, 3)
    for i, L in enumerate(P.layers):
        if pol.fused_layer and L.fused is not None and P.reg_plain is not None:
            nxt = P.layers[i + 1] if i + 1 < len(P.layers) else None
            x, ol, _, pred = run_layer_fused(st, L, x, memory, anchors_2d, valid, B, V, J, hgt, wid, ol=ol, next_P=nxt,
                                             post={"g": P.post[i][0], "b": P.post[i][1]},
                                             reg={"w0": P.reg_plain[i][0], "b0": P.reg_plain[i][1], "w2": P.reg_plain[i][2],
                                                  "b2": P.reg_plain[i][3], "anchors": a3})
        else:
            x = run_layer(st, L, x, memory, anchors_2d, valid, B, V, J, hgt, wid)
            xn = hip.layernorm(x, P.post[i][0], P.post[i][1])
            r = linear(st, xn, P.reg0[i], ACT_GELU)
            pred = linear(st, r, P.reg2[i], res=_rows(a3), res_mode=RES_AFTER_ACT)   # offset + init_anchors_3d
        preds.append(pred.view(B, J, 3))
    aux = {"anchors_2d": anchors_2d, "anchors_valid": valid, "anchors_3d_after": anchors_3d}
    return preds, aux


def _to_view_major(t: torch.Tensor) -> torch.Tensor:
    """(B, V, C, H, W) logical tensor -> (V*B, H, W, C) contiguous buffer (zero-copy when it is one of ours)."""
    B, V, C, H, W = t.shape
    p = t.permute(1, 0, 3, 4, 2)
    if not p.is_contiguous():
        p = p.contiguous()
    return p.reshape(V * B, H, W, C)


@_under_module_policy
def pose3d_forward_api(p3, feat_init, feat_final, ctm=None):
    _check_input(feat_init, p3)
    st = _state(p3, feat_init.device)
    B, V = feat_init.shape[:2]
    preds, aux = _pose3d(p3, st, _to_view_major(feat_init.float()), _to_view_major(feat_final.float()), B, V, ctm)
    p3.__dict__["_egr_last_aux"] = aux
    return preds


@_under_module_policy
def mvfex_forward_api(mod, img, ctm=None):
    """EgoPoseFormerMVFEX.forward (egoposeformer_mvf_ex.py:50-59) -> (list_pred_pose3d, list_pred_heatmap)."""
    _check_input(img, mod)
    img = as_rgb(img)
    B, V = img.shape[:2]
    he = mod.heatmap_estimator
    hm_init, hm_ref, feat_all, feat_ref, aux_h = _mvfex(he, img)
    st = _state(mod.pose3d_estimator, img.device)
    preds, aux_p = _pose3d(mod.pose3d_estimator, st, feat_all, feat_ref, B, V, ctm, behind=_state(he, img.device))
    mod.__dict__["_egr_last_aux"] = {"heatmap": aux_h, "pose3d": aux_p}
    return preds, [hm_init, hm_ref]


hip.install_policy_properties(__name__, {"W_FORMAT": "w_format", "LAYER_H2": "layer_h2", "FUSED_LAYER": "fused_layer", "FUSED_QUERY": "fused_query"})
