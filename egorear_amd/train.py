"""Training step of the hot path on the HIP kernels (BASELINE.json config 5).

What the reference does per optimisation step (pl_wrappers/egoposeformer/pose_3d_mvf_ex.py:114-150, 212-248):
`network.train()`, forward of EgoPoseFormerMVFEX, MPJPE loss on the four pose predictions + row-norm loss on the two
heat-map sets, autograd backward, gradient-norm clipping, AdamW with two parameter groups.  Here the same step runs on
hand-written kernels: the forward in training mode (BatchNorm batch statistics, intermediates kept), a hand-scheduled
reverse pass over a small tape (conv data/weight gradients = egr_conv2d_nhwc_f32 transposed mode / egr_conv2d_wgrad_f32,
everything else = include/egorear_train.h), loss kernels, and a fused clip + AdamW over flat parameter ranges.

Gradient stops follow the reference's shipped pose3d configs (full_training, use_pred_heatmap_init, detach_heatmap_feat*):
  * heat-map heads and refiners see *detached* encoder features (heatmap_mvf_ex.py:273-282), so the encoders are trained
    only through the lifting head's deformable attention on feat_init (egoposeformer_mvf_ex.py:431);
  * inside a refiner, `offset_pred + frame_feat.detach()` (:715) leaves frame_feat_proj_layers without any gradient
    (torch leaves .grad = None and AdamW skips those tensors: so does this step), and conv_heatmap_layers see detached
    refined features (:717-721);
  * anchors (arg-max, reprojection) carry no gradient.

torch is used for memory, streams, and *parameter-space* algebra only (re-packing weights into the kernels' layouts,
folding W_v.W_pre of the sample-then-project attention and un-folding its gradient): activations never pass through
a torch operator.
"""
from __future__ import annotations

import contextlib
import math
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import engine, hip
from . import hip_train as T
from . import repack
from .engine import _npad, _pad_rows, _pad_vec, _rows
from .hip import ACT_NONE, ACT_RELU, RES_BEFORE_ACT, RES_NONE, Img, NMap

_WS_FLOATS = 72 << 20   # conv split-K / wgrad slab workspace (288 MB: one slab of mlp_pred.0's 2048 x 32768 gradient)


# The fp16 scheme in the training step: forward and data-gradient launches above the split threshold take the two-plane
# fp16 images of their weight operands; their inputs' abs-max records come from the producing conv launch, or from one extra read
# (hip.conv2d's amax_arena).  EGR_TRAIN_H2=0 keeps the step on the bf16 scheme.
TRAIN_H2 = os.environ.get("EGR_TRAIN_H2", "1") != "0"
# Records of the BatchNorm outputs / gradients as BOUNDS from the batch extremes (hip_train.bn_train / bn_backward) instead of one read
# of the tensor each (EGR_TRAIN_BOUNDS=0: the reads)
TRAIN_BOUNDS = os.environ.get("EGR_TRAIN_BOUNDS", "1") != "0"
# BatchNorm statistics of a conv output taken in the conv's epilogue (egr_conv_aux.bn_partials) instead of by a pass over the tensor
BN_IN_CONV = os.environ.get("EGR_TRAIN_BN_IN_CONV", "1") != "0"


# Weight gradients on a second stream (EGR_TRAIN_SIDE_STREAM=1; off by default): they depend only on a layer's input and output gradient
# and nothing in the reverse pass waits for them before the parameter gradients are flushed, so they can be forked off the main stream
# (own split workspace) and joined in finish_param_grads; a captured step records the fork / join as graph dependencies.  Measured
# SLOWER (batch 32: 34.1 against 32.1 ms per step): the large weight-gradient and data-gradient launches of the CNN stages each fill
# the chip, so running them side by side only makes them share CUs and L2 (conv 12.7 -> 16.0 ms, weight gradients 7.6 -> 12.2 ms of
# kernel time), and the small latency-bound launches of the heads sit in a different phase of the reverse pass than the large ones.
# Forking only the small launches ("small") is no better (33.7 ms): the replay of a graph with ~100 cross-stream edges costs the host
# 21 ms per step and the device follows it.
WGRAD_DIRECT = os.environ.get("EGR_TRAIN_WGRAD_DIRECT", "1") != "0"   # aligned single Linear layers: the weight gradient lands in the flat gradient buffer itself
STEM_FUSED = os.environ.get("EGR_TRAIN_STEM_FUSED", "1") != "0"   # the stem's BatchNorm + ReLU + max-pool as one forward / one reverse launch set
OVERLAP = os.environ.get("EGR_TRAIN_OVERLAP", "1") != "0"       # the leaves of the reverse pass on a second stream (Step.backward)
# which parts (bits): 1 = forward branches (own-view projection, refined heads), 2 = the detached heads' reverse pass, 4 = the refiners' reverse pass
# (measured and removed: the main stream's ~50 small weight gradients forked onto the side stream one by one - 25.4-25.7 -> 27.2-27.4 ms: every
# cross-stream edge of a hipGraph costs more than the 10-us launch it takes off the chain; the lifting head's ~30 PARKED and handed over as one
# batch at its stage marker - 26.3 ms: small launches queued on the side stream crawl beside the encoders' persistent convolutions and
# hold up the refiners' reverse pass behind them, the final join then waits for the side stream)
OVERLAP_PARTS = int(os.environ.get("EGR_TRAIN_OVERLAP_PARTS", "7"))
SIDE_WGRAD = os.environ.get("EGR_TRAIN_SIDE_STREAM", "0")          # "0" | "small" (only launches below the split threshold) | "1" (all)
SIDE_WGRAD = SIDE_WGRAD if SIDE_WGRAD in ("small", "1") else ""


def _conv2d(*a, **k):
    """hip.conv2d under the training step's split-launch rule (hip.X6_TRAIN_MIN_*)."""
    return hip.conv2d(*a, x6_min=(hip.policy().x6_train_min_rows, hip.policy().x6_train_min_flops), **k)


def _ceil32(n: int) -> int:
    return (n + 31) // 32 * 32


# --------------------------------------------------------------------------- packed parameters for one step

class TPack:
    """`groups` same-shape conv / linear modules packed for the forward kernel (w, bias) and, when a data gradient is
    needed, for the transposed launch (wt).  cin_pad / cout_pad: channel counts the kernels see (multiples of 32 on the
    K side); names: state_dict keys of the weights / biases, one per group."""
    __slots__ = ("w", "bias", "wt", "cout", "cin", "cin_pad", "cout_pad", "kh", "kw", "stride", "pad", "groups", "wmeta", "bmeta", "w6", "wt6")

    @property
    def wop(self):      # operand of the forward launch: the bf16x3 image when one is kept (hip.conv2d falls back to w for few rows)
        return self.w6 if self.w6 is not None else self.w

    @property
    def wtop(self):     # operand of the data-gradient launch
        return self.wt6 if self.wt6 is not None else self.wt


class PackCache:
    """Persistent operand buffers of every conv / linear group of the network plus the descriptor table that refreshes
    all of them from the current parameters in ONE egr_repack_f32 launch per step (a handful of very large transposes
    go through one torch copy each).  Built lazily during the first training forward; rebuilt if a parameter's storage
    moves (load_state_dict, .to(), re-homing into the optimiser's flat buffer)."""

    def __init__(self, device):
        self.dev = device
        self.packs: Dict[object, TPack] = {}
        self.table = repack.RepackTable(device)
        self.back = repack.RepackTable(device)    # BatchNorm running statistics: stacked buffers -> the modules' own, end of the forward
        self.torch_refresh: List = []        # (dst, src parameter): dst.copy_(src.t())
        self.images: List = []               # hip.W6 operands re-split from their fp32 matrices after every refresh
        self._w6_table, self._w6_key = None, None
        self._h2_table, self._h2_key = None, None
        # abs-max records of one step's activations and gradients (the graph of a captured step holds pointers into it)
        self.amax = hip.AmaxArena(device, records=1024) if (TRAIN_H2 and hip.policy().h2 and hip.policy().w_format == "f16x2") else None
        self.sources: Dict[int, tuple] = {}  # id(param) -> (param, data_ptr)
        self.ready = False

    def watch(self, p: torch.Tensor):
        self.sources[id(p)] = (p, p.data_ptr())

    def valid(self) -> bool:
        return all(p.data_ptr() == ptr for p, ptr in self.sources.values())


    def refresh(self):
        self.table.run()
        for dst, src in self.torch_refresh:
            T.transpose_into(dst, src.detach().reshape(dst.shape[1], dst.shape[0]))
        # images no launch has taken (layers whose launches are too small for the split) are left stale; the rest in one launch
        used = [w6 for w6 in self.images if w6.used]
        key = tuple(id(w6) for w6 in used)
        if self._w6_table is None or self._w6_key != key:
            self._w6_table, self._w6_key = hip.W6Table(used), key
        self._w6_table.run()
        used = [w6 for w6 in self.images if w6.h2 is not None and w6.h2_used]
        key = tuple(id(w6) for w6 in used)
        if self._h2_table is None or self._h2_key != key:
            self._h2_table, self._h2_key = hip.WH2Table(used), key
        self._h2_table.run()


class NormPack:
    """Per-group affine parameters of a BatchNorm / LayerNorm group, stacked (G, C) in persistent buffers that the cache's one repack
    launch refreshes at the start of a step; for BatchNorm also the running statistics, gathered by the same launch, updated in place
    by the kernel and written back to the modules' own buffers by ONE launch at the end of the forward (PackCache.back) - instead of
    4 stack + 2 G copy_ torch launches per BatchNorm call (20 calls per step)."""
    __slots__ = ("gamma", "beta", "rm", "rv", "tbl")


def norm_pack(cache: PackCache, mods, batchnorm: bool) -> NormPack:
    key = ("norm", tuple(id(m) for m in mods))
    hit = cache.packs.get(key)
    if hit is not None:
        if not cache.ready:
            hit.tbl.run()          # (no step-wide refresh yet: this group's own descriptors)
        return hit
    dev = cache.dev
    G, C = len(mods), mods[0].weight.numel()
    p = NormPack()
    p.gamma = torch.empty((G, C), device=dev, dtype=torch.float32)
    p.beta = torch.empty((G, C), device=dev, dtype=torch.float32)
    p.rm = torch.empty((G, C), device=dev, dtype=torch.float32) if batchnorm else None
    p.rv = torch.empty((G, C), device=dev, dtype=torch.float32) if batchnorm else None
    p.tbl = repack.RepackTable(dev)
    for g, m in enumerate(mods):
        pairs = [(m.weight, p.gamma), (m.bias, p.beta)] + ([(m.running_mean, p.rm), (m.running_var, p.rv)] if batchnorm else [])
        for src, dst in pairs:
            for t in (p.tbl, cache.table):
                t.add(repack.COPYPAD, src.detach(), dst, g * C, rows=C, total=C)
            cache.watch(src)       # (a buffer re-homed by dist.BufferSync moves its data_ptr: the cache is rebuilt)
        if batchnorm:
            cache.back.add(repack.COPYPAD, p.rm, m.running_mean, 0, rows=C, total=C, src_off=g * C)
            cache.back.add(repack.COPYPAD, p.rv, m.running_var, 0, rows=C, total=C, src_off=g * C)
    p.tbl.run()
    cache.packs[key] = p
    return p


def _get_cache(net: nn.Module, device) -> PackCache:
    c = net.__dict__.get("_egr_pack_cache")
    if c is None or c.dev != device or not c.valid():
        c = PackCache(device)
        net.__dict__["_egr_pack_cache"] = c
    return c


_BIG_TRANSPOSE = 1 << 20   # linear data-gradient operands above this many elements are refreshed by one tiled torch transpose


def make_pack(cache: PackCache, key, wparts, bparts, name_of, kh: int = 1, kw: int = 1, stride: int = 1, pad: int = 0,
              need_dx: bool = True) -> TPack:
    """One conv / linear group.  wparts[g]: list of (parameter, ci0, cin) whose rows are concatenated (ci0/cin select an
    input-channel slice of the parameter); bparts[g]: list of bias parameters or None.  The operand buffers are
    allocated once and (re)filled by repack descriptors; a plain, unpadded single linear aliases its parameter."""
    hit = cache.packs.get(key)
    if hit is not None:
        return hit
    dev = cache.dev
    p = TPack()
    G = len(wparts)
    taps = kh * kw
    p.cout = sum(int(q.shape[0]) for q, _, _ in wparts[0])
    p.cin = wparts[0][0][2]
    p.kh, p.kw, p.stride, p.pad, p.groups = kh, kw, stride, pad, G
    p.cin_pad, p.cout_pad = _ceil32(p.cin), _ceil32(p.cout)
    # where the gradient of every source goes: (parameter name, rows, first channel, channels per parameter row)
    p.wmeta = [[(name_of(q), int(q.shape[0]), ci0, q.numel() // (int(q.shape[0]) * taps)) for q, ci0, _ in parts] for parts in wparts]
    p.bmeta = [[(name_of(q), q.numel()) for q in parts] for parts in bparts] if bparts[0] is not None else None
    rows_fwd, K = _npad(p.cout), p.cin_pad * taps
    one = wparts[0][0]
    tbl = repack.RepackTable(dev)          # this pack's descriptors, run once now; also appended to the cache's table

    def add(*a, **k):
        tbl.add(*a, **k)
        cache.table.add(*a, **k)

    alias = (G == 1 and len(wparts[0]) == 1 and taps == 1 and rows_fwd == p.cout and p.cin_pad == p.cin and one[1] == 0
             and one[0].numel() == p.cout * p.cin)
    if alias:
        p.w = one[0].detach().view(p.cout, p.cin)        # the parameter already is the forward operand
    else:
        p.w = torch.empty((G, rows_fwd, K) if G > 1 else (rows_fwd, K), device=dev, dtype=torch.float32)
        for g, parts in enumerate(wparts):
            r0 = 0
            for i, (q, ci0, cin) in enumerate(parts):
                rows = int(q.shape[0])
                rows_dst = (rows_fwd - r0) if i == len(parts) - 1 else rows
                add(repack.FWD, q.detach(), p.w, (g * rows_fwd + r0) * K, rows=rows, cin=cin, cin_tot=q.numel() // (rows * taps), ci0=ci0,
                    cin_pad=p.cin_pad, taps=taps, rows_pad=rows_dst, total=rows_dst * K)
                r0 += rows
    for parts in wparts:
        for q, _, _ in parts:
            cache.watch(q)
    if bparts[0] is not None:
        p.bias = torch.empty((G, rows_fwd) if G > 1 else (rows_fwd,), device=dev, dtype=torch.float32)
        for g, parts in enumerate(bparts):
            r0 = 0
            for i, q in enumerate(parts):
                n = q.numel()
                n_dst = (rows_fwd - r0) if i == len(parts) - 1 else n
                add(repack.COPYPAD, q.detach(), p.bias, g * rows_fwd + r0, rows=n, total=n_dst)
                cache.watch(q)
                r0 += n
    else:
        p.bias = None
    p.wt = None
    if need_dx:
        rows_t, Kt = _npad(p.cin_pad), p.cout_pad * taps
        p.wt = torch.empty((G, rows_t, Kt) if G > 1 else (rows_t, Kt), device=dev, dtype=torch.float32)
        big = alias and p.cout_pad == p.cout and rows_t == p.cin and p.w.numel() > _BIG_TRANSPOSE
        if big:
            cache.torch_refresh.append((p.wt, one[0]))
            p.wt.copy_(one[0].detach().view(p.cout, p.cin).t())
        else:
            for g, parts in enumerate(wparts):
                r0 = 0
                for i, (q, ci0, cin) in enumerate(parts):
                    rows = int(q.shape[0])
                    rows_k = (p.cout_pad - r0) if i == len(parts) - 1 else rows
                    add(repack.DGRAD, q.detach(), p.wt, g * rows_t * Kt, rows=rows, cin=cin, cin_tot=q.numel() // (rows * taps), ci0=ci0,
                        cin_pad=p.cin_pad, taps=taps, rows_pad=rows_k, k_off=r0, k_tot=p.cout_pad, total=rows_t * rows_k * taps)
                    r0 += rows
    tbl.run()
    # bf16x3 images of both operands (engine.W_FORMAT / W6_MAX_ELEMS decide; they are re-split after every update)
    from .engine import _w_operand
    p.w6 = p.wt6 = None
    for attr, src in (("w6", p.w), ("wt6", p.wt)):
        if src is not None:
            op = _w_operand(src, h2=TRAIN_H2)
            if isinstance(op, hip.W6):
                setattr(p, attr, op)
                cache.images.append(op)
    cache.packs[key] = p
    return p


# --------------------------------------------------------------------------- gradient store + tape

class _Grads:
    """Gradients of forward tensors, keyed by the identity of the tensor object the forward produced."""

    def __init__(self):
        self.g: Dict[int, torch.Tensor] = {}
        self.gm: Dict[int, torch.Tensor] = {}   # contributions to post-ReLU tensors that already carry the [y > 0] mask
        # Round 6: parts of the reverse pass run on a second stream (Step.backward).  When `track` is on, every stored gradient carries
        # the stream of the launch that produced it; a closure that takes it on ANOTHER stream first makes its stream wait for what
        # the producer's stream has been given so far (and tells the allocator about the second user) - gradient flow across streams
        # is ordered by construction.  (One event per stored gradient would be the finer tool; ending a hipGraph capture that holds
        # a few hundred recorded-and-dropped events crashed inside hipStreamEndCapture, Stream.wait_stream is the proven path.)
        self.track = False
        self.src: Dict[int, object] = {}        # id(gradient tensor) -> producing stream
        self.hold: List = []

    def _produced(self, g: torch.Tensor):
        if self.track:
            self.src[id(g)] = torch.cuda.current_stream(g.device)

    def _taken(self, g: Optional[torch.Tensor], drop: bool):
        if self.track and g is not None:
            st = self.src.pop(id(g), None) if drop else self.src.get(id(g))
            if st is not None:
                cur = torch.cuda.current_stream(g.device)
                if st != cur:
                    cur.wait_stream(st)
                    # the allocator must not hand the block back to the producer's stream while this stream still reads it: the
                    # tensor lives to the end of the step (every stream is joined by then) - Tensor.record_stream inside a hipGraph
                    # capture crashed hipStreamEndCapture on this stack
                    self.hold.append(g)
        return g

    def add_masked(self, t: torch.Tensor, g: torch.Tensor):
        k = id(t)
        old = self._taken(self.gm.get(k), True)
        self.gm[k] = g if old is None else T.add(old, g)
        self._produced(self.gm[k])

    def pop_masked(self, t: torch.Tensor) -> Optional[torch.Tensor]:
        return self._taken(self.gm.pop(id(t), None), True)

    def add(self, t: torch.Tensor, g: torch.Tensor):
        if g.shape != t.shape:
            g = g.view(t.shape)
        k = id(t)
        old = self._taken(self.g.get(k), True)
        self.g[k] = g if old is None else T.add(old, g)
        self._produced(self.g[k])

    def put(self, t: torch.Tensor, g: torch.Tensor):
        """Store g as THE gradient of t (the caller has folded the previous value in)."""
        self.g[id(t)] = g
        self._produced(g)

    def pop(self, t: torch.Tensor) -> Optional[torch.Tensor]:
        return self._taken(self.g.pop(id(t), None), True)

    def peek(self, t: torch.Tensor) -> Optional[torch.Tensor]:
        return self._taken(self.g.get(id(t)), False)


class _OnSide:
    """A reverse-pass closure that runs on the step's second stream (Step.backward).  early: it depends on the loss seeds only and
    feeds nothing but parameter gradients (a detached heat-map head) - enqueued first, under the lifting head's chain of small launches."""
    __slots__ = ("fn", "early")

    def __init__(self, fn, early: bool):
        self.fn, self.early = fn, early


class Step:
    """One training forward/backward: the tape of backward closures, the activation-gradient store and the parameter
    gradients (name -> tensor in the parameter's own shape)."""

    def __init__(self, net: nn.Module, device):
        self.net = net
        self.dev = device
        self.tape: List = []
        self.G = _Grads()
        self.pgrads: Dict[str, torch.Tensor] = {}
        self.names = {id(p): k for k, p in net.named_parameters()}
        self.bufnames = {id(b): k for k, b in net.named_buffers()}
        st = net.__dict__.get("_egr_train_ws")
        if st is None or st[0].device != device:
            st = (torch.empty(_WS_FLOATS, device=device, dtype=torch.float32), T.bn_workspace(device))
            net.__dict__["_egr_train_ws"] = st
        self._ws_main, self.bnws = st
        self.keep: List = []   # forward tensors whose identity keys the gradient store
        self.stage_hook = None                                  # callable(stage) run when a gradient stage is complete (multi-process)
        self.gviews: Optional[Dict[str, torch.Tensor]] = None   # flat-buffer views to write parameter gradients into (Trainer)
        self.gtable = repack.RepackTable(device)                # packed / strided gradient pieces -> parameter layout, one launch
        self.pextra: Dict[str, torch.Tensor] = {}               # gradients computed in parameter space (torch tensors)
        self.pshapes = {k: p.shape for k, p in net.named_parameters()}
        self.cache = _get_cache(net, device)
        if self.cache.ready:
            self.cache.refresh()   # every operand buffer of the step from the current parameters: one launch
        self.amax = self.cache.amax
        if self.amax is not None:
            self.amax.begin()      # (one fill launch: every record of the step starts from zero)
        T.set_arena(self.amax)     # element-wise launches bound their outputs by their inputs' records
        self.side, self.ws_side, self._forked = None, None, False
        # ---- two-stream step (round 6, EGR_TRAIN_OVERLAP): forward_train switches it on for the config-5 graph, whose detached heat-map
        # heads and refiners are LEAVES of the reverse pass (they feed parameter gradients only): they run on the side stream under the
        # lifting head's / the encoders' launches.  One process only: the staged multi-process capture cuts the tape at stage markers.
        self.allow_overlap = False
        self._on_side = False      # launches issued now go to the side stream (its split-K workspace is ws_side)
        self.bwd_side = False      # closures recorded now run on the side stream in the reverse pass ...
        self.bwd_early = False     # ... and ahead of everything else (seed-only leaves)
        self._side_dirty = False   # the side stream holds work the main stream has not waited for
        if SIDE_WGRAD or OVERLAP:
            ss = net.__dict__.get("_egr_side")
            if ss is None or ss[1].device != device:
                ss = (torch.cuda.Stream(device=device), torch.empty(_WS_FLOATS, device=device, dtype=torch.float32))
                net.__dict__["_egr_side"] = ss
            self.side, self.ws_side = ss
        self.relu_out = set()  # ids of tensors produced by a fused ReLU: conv data gradients into them apply the mask themselves
        self.record = True     # False: evaluate without taping (constant sub-graphs)
        self.bn_dirty = False
        self.bn_mods: List = []
        self.loss_terms = None

    def flush_buffers(self):
        """End of a training forward: the running statistics the BatchNorm launches updated in the stacked buffers -> the modules' buffers."""
        if self.bn_dirty:
            self.cache.back.run()
            with torch.no_grad():
                torch._foreach_add_([b.num_batches_tracked for b in self.bn_mods], 1)
            self.bn_mods = []
            self.bn_dirty = False

    # ---- streams
    @property
    def ws(self) -> torch.Tensor:
        """Split-K workspace of the stream the launches currently go to."""
        return self.ws_side if self._on_side else self._ws_main

    def overlap(self) -> bool:
        return bool(OVERLAP and not SIDE_WGRAD and self.allow_overlap and self.stage_hook is None and self.side is not None)

    def _rec(self, fn):
        """Record a reverse-pass closure (on the side stream when the forward marked this section as a leaf: bwd_side)."""
        side = self.bwd_side and self.overlap() and (OVERLAP_PARTS & (2 if self.bwd_early else 4))
        self.tape.append(_OnSide(fn, self.bwd_early) if side else fn)

    @contextlib.contextmanager
    def side_branch(self):
        """Forward: the block's launches go to the side stream, behind everything the main stream has been given so far; whoever
        reads the results on the main stream calls join_side() first."""
        if not (self.overlap() and OVERLAP_PARTS & 1):
            yield
            return
        main = torch.cuda.current_stream(self.dev)
        self.side.wait_stream(main)
        self._on_side = True
        try:
            with torch.cuda.stream(self.side):
                yield
        finally:
            self._on_side = False
            self._side_dirty = True

    def join_side(self):
        if self._side_dirty:
            torch.cuda.current_stream(self.dev).wait_stream(self.side)
            self._side_dirty = False

    # ---- bookkeeping
    def name(self, p: torch.Tensor) -> str:
        return self.names[id(p)]

    def pacc(self, name: str, g: torch.Tensor):
        """A gradient (piece) that already has the parameter's layout; several contributions add up (feat_proj over the
        three lifting layers)."""
        old = self.pextra.get(name)
        self.pextra[name] = g if old is None else old + g

    def gdst(self, name: str) -> torch.Tensor:
        """The tensor the gradient of `name` is assembled in: the optimiser's flat-buffer view, or a fresh tensor."""
        t = self.pgrads.get(name)
        if t is None:
            t = self.gviews[name] if self.gviews is not None else torch.empty(self.pshapes[name], device=self.dev, dtype=torch.float32)
            self.pgrads[name] = t
        return t

    def finish_param_grads(self):
        self.join_side()          # the leaves of the reverse pass on the side stream: their slabs feed the table below
        if self._forked:          # the weight gradients forked onto the side stream: their slabs feed the table below
            torch.cuda.current_stream().wait_stream(self.side)
            self._forked = False
        for name, g in self.pextra.items():       # gradients assembled in parameter space: through the same table (one launch for all)
            dst = self.gdst(name)
            if g.dtype == torch.float32 and g.is_contiguous() and dst.is_contiguous() and g.numel() == dst.numel():
                self.gtable.add(repack.COPYPAD, g, dst, 0, rows=g.numel(), total=g.numel())
            else:
                dst.copy_(g.reshape(self.pshapes[name]))
        self.pextra = {}
        self.gtable.run()
        self.keep.append(self.gtable)       # a captured step replays the table's pinned upload: it must outlive this call
        self.gtable = repack.RepackTable(self.dev)

    def mark_stage(self, stage: int):
        """Record the end (in reverse-pass order) of a gradient stage: when the marker runs, the parameter gradients of the
        stage are complete; they are flushed into their destination and handed to `stage_hook` (all-reduce start)."""
        def bwd():
            if self.stage_hook is not None:
                self.finish_param_grads()
                self.stage_hook(stage)
        if self.record:
            self.tape.append(bwd)      # (a marker, never a leaf: plain entry - tests drive mark_stage / backward on a stand-in object)

    def backward(self):
        side = getattr(self, "side", None)

        def on_side(fn):
            self._on_side = True
            try:
                with torch.cuda.stream(side):
                    fn()
            finally:
                self._on_side = False
                self._side_dirty = True
        # seed-only leaves first: every one of them starts by taking a loss seed out of the gradient store, which orders the side
        # stream behind the seed's launch (and through it behind the whole forward)
        for fn in reversed(self.tape):
            if isinstance(fn, _OnSide) and fn.early:
                on_side(fn.fn)
        for fn in reversed(self.tape):
            if isinstance(fn, _OnSide):
                if not fn.early:
                    on_side(fn.fn)
            else:
                fn()
        self.tape = []

    def grad_behind_relu(self, y: torch.Tensor) -> Optional[torch.Tensor]:
        """Total gradient w.r.t. the pre-activation of a ReLU output y: the contributions that were masked at their source
        (conv data gradients, see _dgrad) plus the [y > 0]-masked rest."""
        dy, dzm = self.G.pop(y), self.G.pop_masked(y)
        if dy is None:
            return dzm
        dz = T.relu_bwd(dy, y)
        return dz if dzm is None else T.add(dz, dzm)

    def _dgrad(self, p: "TPack", x: torch.Tensor, dz: torch.Tensor, h: int, w: int):
        """Data gradient of a conv into its input x, accumulated with what x has received so far.  When x came out of a
        fused ReLU the launch applies that ReLU's mask itself (egr_conv2d_masked_f32) and the result goes to the masked store."""
        prev = self.G.pop(x)
        kw = dict(transposed_out_hw=(h, w), groups=p.groups, res=Img(prev) if prev is not None else None,
                  res_mode=RES_BEFORE_ACT if prev is not None else RES_NONE, workspace=self.ws, split_k=0,
                  amax_arena=self.amax, amax_out=self.amax.new() if self.amax is not None else None)
        if id(x) in self.relu_out:
            dx = _conv2d(Img(dz), p.wtop, p.cin_pad, p.kh, p.kw, p.stride, p.pad, mask=Img(x), **kw).t
            self.G.add_masked(x, dx)
        else:
            self.G.put(x, _conv2d(Img(dz), p.wtop, p.cin_pad, p.kh, p.kw, p.stride, p.pad, **kw).t)

    # ---- conv / linear ------------------------------------------------------------------------------------------
    def pack(self, mods: Sequence[nn.Module], need_dx=True) -> TPack:
        m0 = mods[0]
        conv = isinstance(m0, nn.Conv2d)
        stride, pad = (m0.stride[0], m0.padding[0]) if conv else (1, 0)
        kh, kw = (m0.kernel_size if conv else (1, 1))
        has_b = m0.bias is not None
        cin = m0.weight.shape[1]
        return make_pack(self.cache, (tuple(id(m) for m in mods), need_dx), [[(m.weight, 0, cin)] for m in mods],
                         [[m.bias] if has_b else None for m in mods], self.name, kh, kw, stride, pad, need_dx)

    def _wgrad(self, p: TPack, x4: torch.Tensor, dz4: torch.Tensor, gx_rows: Optional[int] = None):
        """Weight / bias gradients of every group.  x4 (G*n, h, w, cin_pad), dz4 (G*n, ho, wo, cout_pad) dense."""
        xi, di = Img(x4), Img(dz4)
        big = hip.wgrad_is_split(xi, di, p.kh, p.kw, p.groups)
        # a plain, unpadded single Linear / 1x1 conv: the packed gradient IS the parameter's layout - written straight into its place in
        # the flat gradient buffer, no unpack copy (mlp_pred.0: 268 MB read + 268 MB written by the step's repack launch)
        direct = None
        if (WGRAD_DIRECT and p.groups == 1 and len(p.wmeta[0]) == 1 and p.kh * p.kw == 1 and p.cin_pad == p.cin and p.cout_pad == p.cout
                and p.wmeta[0][0][2] == 0 and p.wmeta[0][0][3] == p.cin and not SIDE_WGRAD):
            dst = self.gdst(p.wmeta[0][0][0])
            if dst.is_contiguous() and dst.numel() == p.cout * p.cin and dst.data_ptr() % 16 == 0:
                direct = dst.view(p.cout, p.cin)
        if not SIDE_WGRAD or self.side is None or (big and SIDE_WGRAD == "small"):
            dws, dbs = hip.conv2d_wgrad(xi, di, p.kh, p.kw, p.stride, p.pad, self.ws, want_bias=p.bmeta is not None, groups=p.groups,
                                        amax_arena=self.amax, dw=direct)
        else:
            main = torch.cuda.current_stream()
            if big:
                hip.wgrad_records(xi, di, self.amax)      # on the main stream: the data-gradient launch of this layer reads dz's record there
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                dws, dbs = hip.conv2d_wgrad(xi, di, p.kh, p.kw, p.stride, p.pad, self.ws_side, want_bias=p.bmeta is not None, groups=p.groups)
            for t in (x4, dz4):
                t.record_stream(self.side)
            for t in (dws, dbs):
                if t is not None:
                    t.record_stream(main)
            self._forked = True
        taps = p.kh * p.kw
        Kp = p.cin_pad * taps
        for g in range(p.groups):       # packed (rows, cin_pad/32, taps, 32) pieces -> OIHW (column slices) of the parameters' gradients
            r0 = 0
            for name, rows, ci0, cin_tot in (p.wmeta[g] if direct is None else []):
                self.gtable.add(repack.UNPACK, dws, self.gdst(name), 0, rows=rows, cin=p.cin, cin_tot=cin_tot, ci0=ci0, cin_pad=p.cin_pad,
                                taps=taps, total=rows * p.cin * taps, src_off=(g * p.cout_pad + r0) * Kp)
                r0 += rows
            if p.bmeta is not None:
                r0 = 0
                for name, n in p.bmeta[g]:
                    self.gtable.add(repack.COPYPAD, dbs, self.gdst(name), 0, rows=n, total=n, src_off=g * p.cout_pad + r0)
                    r0 += n

    def conv(self, x: torch.Tensor, p: TPack, act: int = ACT_NONE, res: Optional[torch.Tensor] = None, need_dx: bool = True,
             out_pad: bool = False, res_up2: bool = False, stats: bool = False) -> torch.Tensor:
        """x dense NHWC (G*n, h, w, cin_pad) -> y (G*n, ho, wo, cout) [cout_pad wide, zero padded, when out_pad].
        res (same shape as y) is added before the activation.  ReLU is the only fused activation in training mode."""
        assert act in (ACT_NONE, ACT_RELU) and x.shape[-1] == p.cin_pad, (x.shape, p.cin_pad)
        n, h, w, _ = x.shape
        ho = (h + 2 * p.pad - p.kh) // p.stride + 1
        wo = (w + 2 * p.pad - p.kw) // p.stride + 1
        cw = p.cout_pad if out_pad else p.cout
        y = T.zeros((n, ho, wo, cw), self.dev) if (out_pad and cw != p.cout) else torch.empty((n, ho, wo, cw), device=self.dev)
        yo = Img(y[..., :p.cout]) if cw != p.cout else Img(y)
        rec = self.amax.new() if self.amax is not None else None
        # res_up2: `res` is a half-resolution tensor that the epilogue up-samples itself (the FPN top-down add)
        # stats: a train-mode BatchNorm consumes y next - its statistics pass is folded into this launch's epilogue where the launch
        # supports it (raw dense output, cout % 64 == 0); the BatchNorm then only finalises the slabs (Step.bn)
        slabs = None
        kw_bn = {}
        if stats and BN_IN_CONV and act == ACT_NONE and res is None and cw == p.cout and p.cout % 64 == 0 and p.bias is None:
            slabs = []
            kw_bn = dict(bn_ws=self.bnws, bn_slabs=slabs)
        try:
            _conv2d(Img(x), p.wop, p.cout, p.kh, p.kw, p.stride, p.pad, shift=p.bias, act=act, res=Img(res) if res is not None else None,
                    res_mode=(hip.RES_UP2_BEFORE_ACT if res_up2 else RES_BEFORE_ACT) if res is not None else RES_NONE, out=yo,
                    workspace=self.ws, split_k=0, groups=p.groups, amax_arena=self.amax, amax_out=rec, **kw_bn)
        except hip.LaunchError as exc:
            # only "this launch shape has no statistics epilogue" (EINVAL) and "the slabs do not fit" (EWORKSPACE) have a second way:
            # the plain launch + the statistics pass.  Anything else (a device fault, a missing pointer) is raised as it is.
            if not kw_bn or exc.code not in (hip.EINVAL, hip.EWORKSPACE):
                raise
            slabs = None
            _conv2d(Img(x), p.wop, p.cout, p.kh, p.kw, p.stride, p.pad, shift=p.bias, act=act, out=yo, workspace=self.ws, split_k=0,
                    groups=p.groups, amax_arena=self.amax, amax_out=rec)
        if slabs:
            y._egr_bn_slabs = slabs[0]
        if rec is not None:
            y._egr_amax = rec       # (the zero padding columns of an out_pad tensor do not move the maximum)

        if act == ACT_RELU and cw == p.cout:
            self.relu_out.add(id(y))

        def bwd():
            dz = self.grad_behind_relu(y) if act == ACT_RELU else self.G.pop(y)
            if dz is None:
                return
            if res is not None:
                self.G.add(res, T.upsample2x_bwd(dz, None) if res_up2 else dz)
            if dz.shape[-1] != p.cout_pad:
                raise RuntimeError("egorear_amd.train: gradient of a narrow conv output must arrive channel-padded")
            self._wgrad(p, x, dz)
            if need_dx:
                self._dgrad(p, x, dz, h, w)
        if self.record:
            self._rec(bwd)
        self.keep.append((x, y))
        return y

    def linear(self, x: torch.Tensor, p: TPack, act: int = ACT_NONE, need_dx: bool = True, out_pad: bool = False) -> torch.Tensor:
        """x (G*rows, cin_pad) -> (G*rows, cout[_pad])."""
        x4 = x.view(x.shape[0], 1, 1, x.shape[1])
        self.alias(x, x4)                       # recorded before the conv: runs after it in the reverse pass
        y4 = self.conv(x4, p, act, need_dx=need_dx, out_pad=out_pad)
        y = y4.view(y4.shape[0], y4.shape[3])
        self.alias(y4, y)                       # recorded after the conv: runs before it in the reverse pass
        return y

    def alias(self, base: torch.Tensor, view: torch.Tensor):
        """`view` is a reshape of `base`: hand the gradient of the view over to the base.  Record it between the op that
        produces `base` and the ops that consume `view` (the reverse pass runs the tape backwards)."""
        def bwd():
            g = self.G.pop(view)
            if g is not None:
                self.G.add(base, g.view(base.shape))
        if self.record:
            self._rec(bwd)
        self.keep.append((base, view))

    # ---- BatchNorm (training mode), grouped ----------------------------------------------------------------------
    def bn(self, x: torch.Tensor, bns: Sequence[nn.BatchNorm2d], res: Optional[torch.Tensor] = None, relu: bool = True,
           pool: Optional[tuple] = None) -> torch.Tensor:
        """pool = (k, stride, pad): the MaxPool2d that follows BatchNorm + ReLU evaluated in the same pass (the stem, STEM_FUSED): the
        result is the pooled tensor; neither the normalised tensor nor the max-pool's scattered gradient is ever written."""
        G = len(bns)
        npk = norm_pack(self.cache, bns, True)
        gamma, beta, rm, rv = npk.gamma, npk.beta, npk.rm, npk.rv
        b0 = bns[0]
        slabs = getattr(x, "_egr_bn_slabs", None)        # the producing conv left the statistics slabs in self.bnws (Step.conv stats=True)
        if slabs is not None:
            x._egr_bn_slabs = None
        y, ctx = T.bn_train(x, gamma, beta, rm, rv, G, self.bnws, res=res, relu=relu, momentum=b0.momentum, eps=b0.eps, slabs=slabs,
                            amax_out=self.amax.new() if (self.amax is not None and TRAIN_BOUNDS) else None,
                            want_extremes=self.amax is not None and TRAIN_BOUNDS,      # (the backward's bound needs max |xhat| even when y gets none)
                            pool=pool)
        # nn.BatchNorm2d buffer side effects of a training forward: the running statistics go back to the modules' buffers in one launch
        # at the end of the forward, the counters are incremented there in one multi-tensor launch (flush_buffers)
        self.bn_mods += list(bns)
        self.bn_dirty = True

        if relu and pool is None:
            self.relu_out.add(id(y))

        def bwd():
            dy = self.G.pop(y)
            dzm = self.G.pop_masked(y) if (relu and pool is None) else None
            if dy is None and dzm is None:
                return
            if pool is not None:     # dy: gradient of the pooled tensor; the ReLU mask is recomputed from x inside the launch
                dx, dgam, dbet, _ = T.bn_backward(ctx, dy, None, self.bnws,
                                                  amax_dx=self.amax.new() if (self.amax is not None and TRAIN_BOUNDS) else None)
                dz = None
            elif dzm is not None:      # (part of) the gradient arrived already masked: finish the sum, no mask inside the kernels
                if dy is not None:
                    dzm = T.add(dzm, T.relu_bwd(dy, y))
                dx, dgam, dbet, _ = T.bn_backward(ctx, dzm, None, self.bnws, want_dz=False,
                                                  amax_dx=self.amax.new() if (self.amax is not None and TRAIN_BOUNDS) else None)
                dz = dzm
            else:
                dx, dgam, dbet, dz = T.bn_backward(ctx, dy, y if relu else None, self.bnws, want_dz=res is not None,
                                                   amax_dx=self.amax.new() if (self.amax is not None and TRAIN_BOUNDS) else None)
            c_ = dgam.shape[1]
            for g, b in enumerate(bns):
                self.gtable.add(repack.COPYPAD, dgam, self.gdst(self.name(b.weight)), 0, rows=c_, total=c_, src_off=g * c_)
                self.gtable.add(repack.COPYPAD, dbet, self.gdst(self.name(b.bias)), 0, rows=c_, total=c_, src_off=g * c_)
            if res is not None:
                self.G.add(res, dz)
            self.G.add(x, dx)
        if self.record:
            self._rec(bwd)
        self.keep.append((x, y))
        return y

    # ---- pooling / resampling / element-wise -----------------------------------------------------------------------
    def maxpool(self, x: torch.Tensor, k: int, s: int, pd: int) -> torch.Tensor:
        yi, slot = T.maxpool_train(Img(x), k, s, pd)
        y = yi.t

        def bwd():
            dy = self.G.pop(y)
            if dy is not None:
                self.G.add(x, T.maxpool_bwd(dy, slot, (x.shape[1], x.shape[2]), k, s, pd))
        if self.record:
            self._rec(bwd)
        self.keep.append((x, y))
        return y

    def upsample(self, x: torch.Tensor, relu: bool = False) -> torch.Tensor:
        y = hip.upsample2x(Img(x), relu=relu).t
        if relu:
            self.relu_out.add(id(y))

        def bwd():
            dy = self.G.pop(y)
            dzm = self.G.pop_masked(y) if relu else None
            if dzm is None:
                if dy is not None:
                    self.G.add(x, T.upsample2x_bwd(dy, y if relu else None))   # the ReLU mask is applied inside the adjoint
                return
            if dy is not None:
                dzm = T.add(dzm, T.relu_bwd(dy, y))
            self.G.add(x, T.upsample2x_bwd(dzm, None))
        if self.record:
            self._rec(bwd)
        self.keep.append((x, y))
        return y

    def add(self, a: torch.Tensor, b: Optional[torch.Tensor], b_const: Optional[torch.Tensor] = None) -> torch.Tensor:
        """a + b with gradient to both, or a + b_const with gradient to `a` only (a detached operand)."""
        other = b if b is not None else b_const
        y = T.add(a, other)

        def bwd():
            dy = self.G.pop(y)
            if dy is None:
                return
            self.G.add(a, dy)
            if b is not None:
                self.G.add(b, dy)
        if self.record:
            self._rec(bwd)
        self.keep.append((a, other, y))
        return y

    def gelu(self, z: torch.Tensor) -> torch.Tensor:
        h = T.gelu(z)

        def bwd():
            dh = self.G.pop(h)
            if dh is not None:
                self.G.add(z, T.gelu_bwd(dh, z))
        if self.record:
            self._rec(bwd)
        self.keep.append((z, h))
        return h

    def layernorm(self, x: torch.Tensor, lns: Sequence[nn.LayerNorm], res: Optional[torch.Tensor] = None) -> torch.Tensor:
        """LayerNorm(x + res) with per-group affine parameters; gradient flows to x and res alike."""
        G = len(lns)
        npk = norm_pack(self.cache, lns, False)
        gamma, beta = npk.gamma.view(-1), npk.beta.view(-1)
        pre = T.add(x, res) if res is not None else x
        y = hip.layernorm(pre, gamma, beta, groups=G, eps=lns[0].eps)
        c = x.shape[1]

        def bwd():
            dy = self.G.pop(y)
            if dy is None:
                return
            ds, dgam, dbet = T.layernorm_bwd(dy, pre, gamma, G, lns[0].eps)
            for g, l in enumerate(lns):
                self.gtable.add(repack.COPYPAD, dgam, self.gdst(self.name(l.weight)), 0, rows=c, total=c, src_off=g * c)
                self.gtable.add(repack.COPYPAD, dbet, self.gdst(self.name(l.bias)), 0, rows=c, total=c, src_off=g * c)
            self.G.add(x, ds)
            if res is not None:
                self.G.add(res, ds)
        if self.record:
            self._rec(bwd)
        self.keep.append((x, res, pre, y))
        return y

    def mha(self, qkv: torch.Tensor, b: int, j: int, heads: int, d: int) -> torch.Tensor:
        att = hip.joint_mha(qkv, b, j, heads, d, d ** -0.5)

        def bwd():
            da = self.G.pop(att)
            if da is not None:
                self.G.add(qkv, T.joint_mha_bwd(qkv, da, b, j, heads, d, d ** -0.5))
        if self.record:
            self._rec(bwd)
        self.keep.append((qkv, att))
        return att


# --------------------------------------------------------------------------- the network in training mode

def _relu_after(seq: Sequence[nn.Module], i: int) -> bool:
    return i + 1 < len(seq) and isinstance(seq[i + 1], nn.ReLU)


def run_stack_train(S: Step, seqs: Sequence[nn.Sequential], x: torch.Tensor, need_dx_first: bool = True, out_pad_last: bool = False) -> torch.Tensor:
    """tree.stack()s of `len(seqs)` groups in training mode (same evaluation order as engine.run_stack, including the
    conv-before-upsample form, whose backward is the upsample adjoint followed by the 1x1 gradients on the coarse grid)."""
    mods = [list(s) for s in seqs]
    m0 = mods[0]
    i = 0
    first = True
    while i < len(m0):
        m = m0[i]
        if isinstance(m, nn.Conv2d):
            act = ACT_RELU if _relu_after(m0, i) else ACT_NONE
            last = (i + (2 if act else 1)) >= len(m0)
            p = S.pack([g[i] for g in mods], need_dx=(need_dx_first or not first))
            x = S.conv(x, p, act, need_dx=(need_dx_first or not first), out_pad=(last and out_pad_last))
            i += 2 if act else 1
        elif isinstance(m, nn.Upsample):
            nxt = m0[i + 1] if i + 1 < len(m0) else None
            if isinstance(nxt, nn.Conv2d) and nxt.kernel_size == (1, 1) and nxt.stride == (1, 1) and i + 2 < len(m0) and isinstance(m0[i + 2], nn.ReLU):
                p = S.pack([g[i + 1] for g in mods], need_dx=(need_dx_first or not first))
                lo = S.conv(x, p, ACT_NONE, need_dx=(need_dx_first or not first))
                x = S.upsample(lo, relu=True)
                i += 3
            else:
                x = S.upsample(x)
                i += 1
        elif isinstance(m, nn.MaxPool2d):
            k = m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
            s = m.stride if isinstance(m.stride, int) else m.stride[0]
            pd = m.padding if isinstance(m.padding, int) else m.padding[0]
            x = S.maxpool(x, k, s, pd)
            i += 1
        else:
            raise RuntimeError(f"egorear_amd.train: unexpected module {type(m).__name__}")
        first = False
    return x


def _basic_block_train(S: Step, blks, x: torch.Tensor) -> torch.Tensor:
    b0 = blks[0]
    identity = x
    if b0.downsample is not None:
        pd = S.pack([b.downsample[0] for b in blks])
        identity = S.bn(S.conv(x, pd, stats=True), [b.downsample[1] for b in blks], relu=False)
    y = S.bn(S.conv(x, S.pack([b.conv1 for b in blks]), stats=True), [b.bn1 for b in blks], relu=True)
    return S.bn(S.conv(y, S.pack([b.conv2 for b in blks]), stats=True), [b.bn2 for b in blks], res=identity, relu=True)


def backbone_train(S: Step, encs, img: torch.Tensor, view0: int, nviews: int):
    """ResNet-18 trunk + FPN of len(encs) encoders (resnet.py:43-74, 121-137) in training mode.  Returns
    (feat (G*nviews*B, 64, 64, 128), s32)."""
    G = len(encs)
    trunks, necks = [e.backbone for e in encs], [e.neck for e in encs]
    B, V, _, H, W = img.shape
    # stem: the inference kernel in raw mode (bare conv; BatchNorm on batch statistics follows); its weight gradient comes
    # from egr_stem_wgrad_f32, which re-stages the same input patches (no im2col buffer)
    w7 = [t.layer_s2[0].weight for t in trunks]
    key = ("stem", tuple(id(w) for w in w7))
    wp = S.cache.packs.get(key)
    if wp is None:
        wp = torch.empty((G, 64, 148), device=S.dev, dtype=torch.float32)
        tbl = repack.RepackTable(S.dev)
        for g, w in enumerate(w7):       # (64, 3, 7, 7) -> (64, 148): OIHW flattening + one zero column
            for t_ in (tbl, S.cache.table):
                t_.add(repack.FWD, w.detach(), wp, g * 64 * 148, rows=64, cin=147, cin_tot=147, cin_pad=148, taps=1, rows_pad=64, total=64 * 148)
            S.cache.watch(w)
        tbl.run()
        S.cache.packs[key] = wp
    # (split-bf16 launch like the other convolutions; the bank is re-split from the refreshed fp32 pack every step: one tiny launch)
    # (with TRAIN_H2 in the fp16 scheme like the inference stem: per-tile pre-scale from the patch itself, no record needed)
    if engine.STEM_X6 and H % 32 == 0 and W % 64 == 0:
        if TRAIN_H2 and hip.policy().h2 and hip.policy().w_format == "f16x2":
            bank, wds = hip.pack_stem_wh2(wp)
            x = hip.stem_x6(img, view0, nviews, bank, None, None, groups=G, w_descale=wds).t
        else:
            x = hip.stem_x6(img, view0, nviews, hip.pack_stem_w6(wp), None, None, groups=G).t
    else:
        x = hip.stem(img, view0, nviews, wp, None, None, groups=G).t
    wnames = [S.name(w) for w in w7]

    def bwd_stem(x=x):
        dy = S.G.pop(x)
        if dy is None:
            return
        dw = T.stem_wgrad(img, view0, nviews, dy, S.ws, groups=G)
        for g, nme in enumerate(wnames):
            S.gtable.add(repack.COPYPAD, dw, S.gdst(nme), 0, rows=64 * 147, total=64 * 147, src_off=g * 64 * 147)
    if S.record:
        S._rec(bwd_stem)
    S.keep.append((x,))
    mp = trunks[0].layer_s4[0]
    mk = mp.kernel_size if isinstance(mp.kernel_size, int) else mp.kernel_size[0]
    ms = mp.stride if isinstance(mp.stride, int) else mp.stride[0]
    mpad = mp.padding if isinstance(mp.padding, int) else mp.padding[0]
    if STEM_FUSED:       # BatchNorm + ReLU + MaxPool2d in one pass, their reverse pass without the full-resolution y / dy (egr_bn_relu_maxpool_f32)
        x = S.bn(x, [t.layer_s2[1] for t in trunks], relu=True, pool=(mk, ms, mpad))
    else:
        x = S.bn(x, [t.layer_s2[1] for t in trunks], relu=True)
        x = S.maxpool(x, mk, ms, mpad)
    pyramid = []
    stages = [(t.layer_s4[1], t.layer_s8, t.layer_s16, t.layer_s32) for t in trunks]
    for si in range(4):
        for bi in range(len(stages[0][si])):
            x = _basic_block_train(S, [stages[g][si][bi] for g in range(G)], x)
        pyramid.append(x)
    # FPN, same split evaluation as engine.run_backbone; the two halves of each fuse conv are separate packs whose weight
    # gradients are the two column blocks of fuse_convs[i].weight's gradient.
    n0 = necks[0]
    c = n0.out_channels
    lat = S.conv(pyramid[3], S.pack([k.lateral_convs[3][0] for k in necks]), ACT_RELU)
    for i in (3, 2, 1):
        fine = S.conv(pyramid[i - 1], S.pack([k.lateral_convs[i - 1][0] for k in necks]), ACT_RELU)
        fws = [k.fuse_convs[i - 1][0] for k in necks]
        pa = make_pack(S.cache, ("fuse_a", tuple(id(f) for f in fws)), [[(f.weight, 0, c)] for f in fws], [[f.bias] for f in fws], S.name)
        pb = make_pack(S.cache, ("fuse_b", tuple(id(f) for f in fws)), [[(f.weight, c, c)] for f in fws], [None] * G, S.name)
        coarse_lo = S.conv(lat, pb, ACT_NONE)
        fused = S.conv(fine, pa, ACT_RELU, res=coarse_lo, res_up2=True)
        lat = S.conv(fused, S.pack([k.fpn_convs[i - 1][0] for k in necks]), ACT_RELU)
    return lat, pyramid[3]


# ---- deformable-attention transformer layer ----------------------------------------------------------------------

class _LayerPack:
    pass


def _pack_layer(S: Step, layers, pres, poss) -> _LayerPack:
    """Per-step packing of `len(layers)` same-shape transformer layers (cf. engine.pack_layers): the value path is folded
    (fp64) for sample-then-project, and everything needed to un-fold the gradients is kept."""
    L = _LayerPack()
    ca0 = layers[0].cross_attn
    L.heads, L.C, L.G = ca0.n_heads, ca0.d_model, len(layers)
    L.dh = L.C // L.heads
    L.layers, L.pres, L.poss = layers, pres, poss
    cas = [l.cross_attn for l in layers]
    # value_proj as a pack: its data-gradient operand Wv^T feeds the fold below and the gradient of the positional embeddings
    L.vproj = S.pack([c.value_proj for c in cas])
    with torch.no_grad():
        Wv = [c.value_proj.weight.detach() for c in cas]
        bv = [c.value_proj.bias.detach() for c in cas]
        Wp = [w.detach().reshape(w.shape[0], -1) for w, _ in pres]
        bp = [b.detach() for _, b in pres]
        # Wfold[g] = Wv[g] . Wp[g] as a weight-gradient-shaped product sum_m WvT[m][c] Wp[m][f] on the small fp32 kernel (one launch per
        # query set; was a double-precision library GEMM + casts), cfold[g] = Wv[g] . bp[g] + bv[g]
        L.Wfold = torch.empty((L.G, L.C, Wp[0].shape[1]), device=S.dev, dtype=torch.float32)
        wt_all = L.vproj.wt if L.G > 1 else L.vproj.wt.unsqueeze(0)
        for g in range(L.G):
            hip.conv2d_wgrad(_rows(Wp[g]), _rows(wt_all[g]), 1, 1, 1, 0, S.ws, dw=L.Wfold[g], x6=False)
        L.cfold = torch.stack([torch.addmv(c, a, b) for a, b, c in zip(Wv, bp, bv)]).contiguous()  # (G, C)
        L.pos_proj = None          # (set below: pos . Wv^T for all query sets in one launch)
        cf = L.Wfold.shape[2]
        L.cf = cf
        # per-head forward operand (G, dh_pad, cf) and data-gradient operand (G, cf, dh) of the folded projection
        L.heads_grouped = L.G == 1 and L.dh % 4 == 0 and cf % 4 == 0 and L.dh % 32 == 0
        if L.dh % 32 == 0:
            # (no row padding needed) three strided copies per layer instead of one slice + pad + stack per (head, query set) - the
            # per-step parameter glue of the four layer packs was ~170 torch launches in front of the token chains
            W4 = L.Wfold.view(L.G, L.heads, L.dh, cf)
            Wh = W4.transpose(0, 1).contiguous()                          # (heads, G, dh, cf)
            Wht = W4.permute(1, 0, 3, 2).contiguous()                     # (heads, G, cf, dh)
            sh = L.cfold.view(L.G, L.heads, L.dh).transpose(0, 1).contiguous()     # (heads, G, dh)
            L.head_w, L.head_shift, L.head_wt = list(Wh.unbind(0)), list(sh.unbind(0)), list(Wht.unbind(0))
            if L.heads_grouped:      # one query set (the lifting head): the heads of a layer run as the GROUPS of one launch
                L.head_w_all, L.head_shift_all, L.head_wt_all = Wh.view(L.heads, L.dh, cf), sh.view(L.heads, L.dh), Wht.view(L.heads, cf, L.dh)
        else:
            L.head_w = [torch.stack([_pad_rows(L.Wfold[g, h * L.dh:(h + 1) * L.dh].contiguous()) for g in range(L.G)]).contiguous() for h in range(L.heads)]
            L.head_shift = [torch.stack([_pad_vec(L.cfold[g, h * L.dh:(h + 1) * L.dh], L.dh) for g in range(L.G)]).contiguous() for h in range(L.heads)]
            L.head_wt = [torch.stack([L.Wfold[g, h * L.dh:(h + 1) * L.dh].t().contiguous() for g in range(L.G)]).contiguous() for h in range(L.heads)]
    C_ = L.C
    if poss[0] is not None:
        # projected positional embeddings (G, V, hw, C): value_proj's forward operand without its bias (c_fold carries it) - was a
        # double-precision library GEMM + two casts per query set and step (78 us each)
        with torch.no_grad():
            pos_all = poss[0].detach()[0] if L.G == 1 else torch.stack([p.detach()[0] for p in poss])
        L.pos_proj = _conv2d(Img(pos_all.reshape(-1, 1, 1, C_)), L.vproj.wop, C_, 1, 1, 1, 0, groups=L.G, workspace=S.ws, split_k=0,
                             amax_arena=S.amax).t.view((L.G,) + tuple(poss[0].shape[1:]))
    L.ol = make_pack(S.cache, ("ol", tuple(id(c) for c in cas)),
                     [[(c.sampling_offsets.weight, 0, C_), (c.attention_weights.weight, 0, C_)] for c in cas],
                     [[c.sampling_offsets.bias, c.attention_weights.bias] for c in cas], S.name)
    L.out_proj = S.pack([c.output_proj for c in cas])
    L.fuse = S.pack([l.fuse_mlp for l in layers])
    sas = [l.spatial_attn for l in layers]
    L.qkv = make_pack(S.cache, ("qkv", tuple(id(s) for s in sas)),
                      [[(s.q_proj.weight, 0, C_), (s.k_proj.weight, 0, C_), (s.v_proj.weight, 0, C_)] for s in sas],
                      [[s.q_proj.bias, s.k_proj.bias, s.v_proj.bias] for s in sas], S.name)
    L.mha_out = S.pack([s.out_proj for s in sas])
    L.ffn0 = S.pack([l.ffn.layers[0][0] for l in layers])
    L.ffn1 = S.pack([l.ffn.layers[1] for l in layers])
    return L


def layer_train(S: Step, L: _LayerPack, x: torch.Tensor, memory: torch.Tensor, anchors, valid, B: int, V: int, J: int, hgt: int, wid: int,
                dmem: Optional[torch.Tensor]) -> torch.Tensor:
    """One MultiViewTransformerLayer / EgoPoseFormerTransformerLayer in training mode (cf. engine.run_layer).  memory
    (V, B, hw, cf) un-projected features; when `dmem` is given the feature gradient of the sampling is accumulated into it
    (the lifting head), otherwise the features are a constant (the refiners' detached memory)."""
    G, C, heads, dh, cf = L.G, L.C, L.heads, L.dh, L.cf
    rows = B * J * V
    ol = S.linear(x, L.ol)
    g, e, sigma, rowmask = hip.msda_gather(memory, L.pos_proj, ol, anchors, valid, B, V, J, heads, dh, hgt, wid, groups=G)
    a = torch.empty((G * rows, C), device=S.dev, dtype=torch.float32)
    g2 = g.view(G * rows, heads * cf)
    e2 = e.view(G * rows, C) if e is not None else None
    sig = sigma.view(G * heads * rows)
    if L.heads_grouped:      # the four heads as four groups of one launch: x / res / out are channel slices of the same rows
        _conv2d(_rows(g2[:rows, 0:cf]), L.head_w_all, dh, 1, 1, 1, 0, shift=L.head_shift_all, rowscale=sig, grs=rows,
                res=_rows(e2[:rows, 0:dh]) if e2 is not None else None, res_mode=hip.RES_AFTER_ACT if e2 is not None else RES_NONE,
                out=_rows(a[:rows, 0:dh]), workspace=None, split_k=1, groups=heads, gx=cf, gr=dh, gy=dh)
    for h in range(0 if L.heads_grouped else heads):
        _conv2d(_rows(g2[:rows, h * cf:(h + 1) * cf]), L.head_w[h] if G > 1 else L.head_w[h][0], dh, 1, 1, 1, 0,
                   shift=L.head_shift[h] if G > 1 else L.head_shift[h][0], rowscale=sig[h * rows:], grs=heads * rows,
                   res=_rows(e2[:rows, h * dh:(h + 1) * dh]) if e2 is not None else None,
                   res_mode=hip.RES_AFTER_ACT if e2 is not None else RES_NONE, out=_rows(a[:rows, h * dh:(h + 1) * dh]),
                   workspace=None, split_k=1, groups=G, gx=rows * heads * cf, gr=rows * C, gy=rows * C)

    def bwd_sampling():
        da = S.G.pop(a)
        if da is None:
            return
        # folded projection: dWfold_h = da_h^T g_h, dcfold_h = sum_r sigma_h da_h, dg_h = da_h Wfold_h
        dWh = torch.empty((heads, G, dh, cf), device=S.dev, dtype=torch.float32)
        dcfold = torch.empty((G, C), device=S.dev, dtype=torch.float32)
        dg = torch.empty((G, rows, heads, cf), device=S.dev, dtype=torch.float32)
        dg2 = dg.view(G * rows, heads * cf)
        if L.heads_grouped:
            hip.conv2d_wgrad(_rows(g2[:, 0:cf]), _rows(da[:, 0:dh]), 1, 1, 1, 0, S.ws, dw=dWh, groups=heads, gx=cf, gy=dh)
            _conv2d(_rows(da[:rows, 0:dh]), L.head_wt_all, cf, 1, 1, 1, 0, out=_rows(dg2[:rows, 0:cf]), workspace=None, split_k=1,
                    groups=heads, gx=dh, gy=cf)
        for h in range(0 if L.heads_grouped else heads):
            hip.conv2d_wgrad(_rows(g2[:, h * cf:(h + 1) * cf]), _rows(da[:, h * dh:(h + 1) * dh]), 1, 1, 1, 0, S.ws, dw=dWh[h], groups=G)
            _conv2d(_rows(da[:rows, h * dh:(h + 1) * dh]), L.head_wt[h] if G > 1 else L.head_wt[h][0], cf, 1, 1, 1, 0,
                       out=_rows(dg2[:rows, h * cf:(h + 1) * cf]), workspace=None, split_k=1, groups=G, gx=rows * C, gy=rows * heads * cf)
        dWfold = dWh.permute(1, 0, 2, 3).reshape(G, C, cf)
        # dcfold[g, h*dh + d] = sum_r sigma[g, h, r] * da[g, r, h*dh + d]: all heads in one launch
        T.colsum(da, C, rows, C, scale=sig, out=dcfold, groups=G, gx=rows * C, gs=heads * rows, cols_per_scale=dh, scale_stride=rows)
        dpos = T.zeros(L.pos_proj.shape, S.dev) if L.pos_proj is not None else None
        dol_v = T.msda_gather_bwd(memory, L.pos_proj, ol, anchors, valid, B, V, J, heads, dh, hgt, wid, dg, da, L.cfold, dmem, dpos, groups=G)
        S.G.add(ol, T.fold_rows(dol_v, V))
        dpos_wv = None
        if dpos is not None:       # (G, V, hw, C) . Wv per query set in one launch (was one library GEMM of 40 us each)
            pv = L.vproj
            dpos_wv = _conv2d(Img(dpos.view(-1, 1, 1, C)), pv.wtop, pv.cin_pad, 1, 1, 1, 0, transposed_out_hw=(1, 1), groups=G, workspace=S.ws,
                              split_k=0, amax_arena=S.amax).t.view(dpos.shape)
        # un-fold in parameter space (fp32 matmuls on (C, C)-sized operands)
        with torch.no_grad():
            for gi, layer in enumerate(L.layers):
                ca = layer.cross_attn
                Wv = ca.value_proj.weight.detach().float()
                pw, pb = L.pres[gi]
                Wp = pw.detach().float().reshape(pw.shape[0], -1)
                dWf = dWfold[gi].contiguous()
                # dWfold . Wp^T as a Linear over the C rows of dWfold, Wv^T . dWfold as a weight-gradient-shaped product (small fp32 kernels)
                dWv = _conv2d(_rows(dWf), Wp, Wp.shape[0], 1, 1, 1, 0, workspace=S.ws, split_k=0).t.view(Wp.shape[0], Wp.shape[0])
                dWv = torch.addr(dWv, dcfold[gi], pb.detach().float())
                if dpos is not None:
                    pos = L.poss[gi].detach()[0].float()                               # (V, hw, C)
                    # sum_p dpos[p, o] pos[p, i]: a weight gradient over V*hw rows (fp32 MFMA; the library GEMM took 61 us per query set)
                    dWv = dWv + hip.conv2d_wgrad(_rows(pos.reshape(-1, pos.shape[-1])), _rows(dpos[gi].reshape(-1, dpos.shape[-1])), 1, 1, 1, 0,
                                                 S.ws, x6=False)[0]
                    S.pacc(S.name(L.poss[gi]), dpos_wv[gi].unsqueeze(0))
                S.pacc(S.name(ca.value_proj.weight), dWv)
                S.pacc(S.name(ca.value_proj.bias), dcfold[gi].clone())
                S.pacc(S.name(pw), hip.conv2d_wgrad(_rows(dWf), _rows(Wv), 1, 1, 1, 0, S.ws, x6=False)[0].reshape(pw.shape))
                S.pacc(S.name(pb), Wv.t() @ dcfold[gi])
    S._rec(bwd_sampling)
    S.keep.append((ol, g, e, sigma, a))
    # masked_fill(~valid) after output_proj: rows of invalid anchors are zero and pass no gradient.  Forward and gradient
    # are masked in place (the gradient tensor is the fresh output of fuse_mlp's data-gradient launch, nobody else holds it).
    om = S.linear(a, L.out_proj)
    mask_all = rowmask.repeat(G) if G > 1 else rowmask
    T.rowmask_(om, mask_all)

    def bwd_mask():
        d = S.G.peek(om)
        if d is not None:
            T.rowmask_(d, mask_all)
    S._rec(bwd_mask)
    omv = om.view(G * B * J, V * C)
    S.alias(om, omv)
    f = S.linear(omv, L.fuse)
    x = S.layernorm(f, [l.norm_cross for l in L.layers], res=x)
    qkv = S.linear(x, L.qkv)
    att = S.mha(qkv, G * B, J, heads, dh)
    x = S.layernorm(S.linear(att, L.mha_out), [l.norm_spatial for l in L.layers], res=x)
    h1 = S.gelu(S.linear(x, L.ffn0))
    x = S.layernorm(S.linear(h1, L.ffn1), [l.norm_ffn for l in L.layers], res=x)
    return x


# ---- conv whose output is the (B, V, c, H, W) channel-major heat-map tensor ----------------------------------------

def conv_to_planes(S: Step, x: torch.Tensor, p: TPack, planes: torch.Tensor, B: int, V: int, need_dx: bool = True):
    """Final 1x1 conv of a heat-map head: writes planes[b, v] for the view-major images n = v*B + b held by x (all groups
    stacked).  Its gradient arrives as a tensor shaped like `planes` (S.G.add(planes, d))."""
    n, h, w, _ = x.shape
    assert n == V * B and p.kh == 1 and p.stride == 1
    plane = p.cout * h * w
    per_group = n // p.groups
    vpg = per_group // B                       # views per group
    _conv2d(Img(x), p.wop, p.cout, 1, 1, 1, 0, shift=p.bias, out_nchw=planes, ymap=NMap(B, V * plane, plane if vpg > 1 else 0),
               gy=vpg * plane, groups=p.groups, workspace=S.ws, split_k=0)

    def bwd():
        d = S.G.pop(planes)
        if d is None:
            return
        dz = T.planes_to_nhwc(d, NMap(B, V * plane, plane), n, p.cout, h * w, p.cout_pad).view(n, h, w, p.cout_pad)
        S._wgrad(p, x, dz)
        if need_dx:
            S._dgrad(p, x, dz, h, w)
    S._rec(bwd)
    S.keep.append((x, planes))


def heatmap_head_train(S: Step, seqs, x: torch.Tensor, planes: torch.Tensor, B: int, V: int, need_dx_first: bool = False):
    """A heat-map head stack (on *detached* features unless need_dx_first: heatmap_mvf_ex.py:273, :717-721): all layers but
    the last through run_stack_train, the last 1x1 conv straight into the channel-major output."""
    body = [nn.Sequential(*list(s)[:-1]) for s in seqs]
    y = run_stack_train(S, body, x, need_dx_first=need_dx_first)
    conv_to_planes(S, y, S.pack([s[-1] for s in seqs]), planes, B, V)


# ---- the four HeatmapMVF refiners ------------------------------------------------------------------------------------

def refiners_train(S: Step, rs, B: int, V: int, hm_init: torch.Tensor, feat_all: torch.Tensor, s32_all: torch.Tensor, anchors, valid,
                   hm_ref: torch.Tensor, hm_grad: bool = False, detach_heatmap_feat: bool = True) -> torch.Tensor:
    """heatmap_mvf_ex.py:652-731 for the G = V refiners in training mode.  feat_all, s32_all are constants here (detached or
    produced under no_grad in the reference); hm_init too unless hm_grad (stage-2 training: the initial heat maps feed
    heatmap_proj with gradient).  detach_heatmap_feat: the refined heat-map head sees detached refined features (:717-721).
    Returns feat_ref (V*B, 64, 64, 128), which carries gradient back from the lifting head / the refined head."""
    G = len(rs)
    r0 = rs[0]
    J, C = r0.num_heatmap, r0.embed_dims
    hgt, wid = r0.feat_shape
    hw = hgt * wid
    dev = S.dev
    S.mark_stage(1)
    # --- own-view projection: a constant in this configuration (`offset_pred + frame_feat.detach()`, :715) - evaluated without tape, and
    # (two-stream step) beside the token chain below: ~1 ms of convolutions under ~0.7 ms of small launches
    S.record = False
    with S.side_branch():
        ff = run_stack_train(S, [r.frame_feat_proj_layers for r in rs], feat_all, need_dx_first=False)
    S.record = True
    # --- JQA query: heatmap_proj.0 reads the (B, V, J, hw) heat maps in place, group g = view g
    hp0 = S.pack([r.heatmap_proj[0] for r in rs], need_dx=hm_grad)
    hm_rows = Img(hm_init.view(B * V, J, 1, hw)[0::V])
    t4 = _conv2d(hm_rows, hp0.w, C, 1, 1, 1, 0, shift=hp0.bias, act=ACT_RELU, groups=G, gx=J * hw, workspace=S.ws, split_k=0).t   # (G*B, J, 1, C)
    t = t4.view(G * B * J, C)

    def bwd_hp0():
        d = S.G.pop(t)
        if d is None:
            return
        dz = T.relu_bwd(d, t).view(G * B, J, 1, C)
        for g in range(G):
            xg = Img(hm_init.view(B, V, J, 1, hw)[:, g])                        # (B, J, 1, hw) strided over the batch
            dw, db = hip.conv2d_wgrad(xg, Img(dz[g * B:(g + 1) * B]), 1, 1, 1, 0, S.ws, want_bias=True)
            S.pacc(hp0.wmeta[g][0][0], dw)
            S.pacc(hp0.bmeta[g][0][0], db)
        if hm_grad:     # gradient w.r.t. the heat-map rows, back into the (B, V, J, H, W) layout (group g = view g)
            dxr = _conv2d(Img(dz), hp0.wt, hw, 1, 1, 1, 0, transposed_out_hw=(J, 1), groups=G, workspace=S.ws, split_k=0).t
            dhm = torch.empty_like(hm_init)
            T.nhwc_to_planes(dxr.view(G * B, 1, J * hw), dhm, NMap(B, V * J * hw, J * hw), J * hw)
            S.G.add(hm_init, dhm)
    S._rec(bwd_hp0)
    S.keep.append((t4, t))
    hm_embed = S.linear(t, S.pack([r.heatmap_proj[2] for r in rs]))
    bfb = S.linear(hip.avgpool(Img(s32_all)), S.pack([r.fc_bfb for r in rs], need_dx=False), need_dx=False)
    with torch.no_grad():
        embed = torch.stack([r.joint_query_embed.weight.detach().float() for r in rs]).contiguous()
    xs = hip.jqa_sum(hm_embed, embed, bfb, G * B, J, C, groups=G)

    def bwd_jqa():
        d = S.G.pop(xs)
        if d is None:
            return
        de, dbfb = T.jqa_sum_bwd(d, G * B, J, C, G)
        for g, r in enumerate(rs):
            S.pacc(S.name(r.joint_query_embed.weight), de[g])
        S.G.add(bfb, dbfb)
        S.G.add(hm_embed, d)
    S._rec(bwd_jqa)
    S.keep.append((xs,))
    x = S.linear(xs, S.pack([r.fc_query[0] for r in rs]), ACT_RELU)
    # --- transformer layer over the (detached) 4-view memory
    L = _pack_layer(S, [r.transformer_layers[0] for r in rs], [(r.frame_feat_multi_view_proj.weight, r.frame_feat_multi_view_proj.bias) for r in rs],
                    [r.frame_feat_multi_view_pos_embed for r in rs])
    x = layer_train(S, L, x, feat_all.view(V, B, hw, feat_all.shape[-1]), anchors, valid, B, V, J, hgt, wid, None)
    xn = S.layernorm(x, [r.post_norm[0] for r in rs])
    side = int(math.isqrt(C))
    # (G*B, J, 256) tokens -> (G*B, 16, 16, 32) image with the 15 joints as (zero-padded) channels, and back
    tok = T.planes_to_nhwc(xn, NMap(G * B, J * C, 0), G * B, J, C, 32).view(G * B, side, side, 32)

    def bwd_tok():
        d = S.G.pop(tok)
        if d is not None:
            dx = torch.empty_like(xn)
            T.nhwc_to_planes(d.view(G * B, C, 32), dx, NMap(G * B, J * C, 0), J)
            S.G.add(xn, dx)
    S._rec(bwd_tok)
    S.keep.append((tok,))
    h0 = S.conv(tok, S.pack([r.head_layers[0].head[0] for r in rs]), ACT_RELU)
    h0 = S.upsample(h0)
    off = S.conv(h0, S.pack([r.head_layers[0].head[3] for r in rs]), ACT_RELU)
    S.join_side()                                         # ff
    summed = S.add(off, None, b_const=ff)
    feat_ref = run_stack_train(S, [r.frame_feat_refined_proj_layers[0] for r in rs], summed)
    # the refined heads: forward beside whatever the caller runs next on the main stream (the lifting head's token chain; the caller
    # joins before it reads hm_ref), reverse pass as a seed-only leaf when they see detached features
    early, S.bwd_early = S.bwd_early, S.bwd_early or (S.bwd_side and detach_heatmap_feat)
    with S.side_branch():
        heatmap_head_train(S, [r.conv_heatmap_layers[0] for r in rs], feat_ref, hm_ref, B, V, need_dx_first=not detach_heatmap_feat)
    S.bwd_early = early
    return feat_ref


# ---- lifting head ------------------------------------------------------------------------------------------------------

def pose3d_train(S: Step, p3, feat_init: torch.Tensor, feat_ref: torch.Tensor, B: int, V: int, ctm):
    """EgoPoseFormerPose3D.forward (egoposeformer_mvf_ex.py:422-452) in training mode.  Returns the four predictions as
    channel-padded buffers [(B, 64) proposal, 3 x (B*16, 32)] whose first 48 / 3 columns are the coordinates."""
    dev = S.dev
    J = p3.num_joints
    hgt, wid = p3.feat_shape
    C = p3.embed_dims
    assert p3.use_pred_heatmap_init, "training path covers the shipped pose3d configs (use_pred_heatmap_init)"
    S.mark_stage(0)                     # reverse pass: everything recorded below is done when this marker runs
    memory = feat_init.view(V, B, hgt * wid, feat_init.shape[-1])
    dmem = T.zeros(memory.shape, dev)

    def bwd_mem():                      # runs after every layer's sampling backward has accumulated into dmem
        S.G.add(feat_init, dmem.view(feat_init.shape))
    S._rec(bwd_mem)
    # --- proposal: conv stack on the refined features -> per-frame vector in the reference's (v, c, h, w) order
    cf = run_stack_train(S, [p3.conv_frame_feat], feat_ref)                     # (V*B, 8, 8, 128)
    flat = torch.empty((B, V * 128 * 64), device=dev, dtype=torch.float32)
    T.nhwc_to_planes(cf.view(V * B, 64, 128), flat, NMap(B, V * 8192, 8192), 128)

    def bwd_flat():
        d = S.G.pop(flat)
        if d is not None:
            S.G.add(cf, T.planes_to_nhwc(d, NMap(B, V * 8192, 8192), V * B, 128, 64, 128).view(cf.shape))
    S._rec(bwd_flat)
    S.keep.append((flat,))
    h = S.gelu(S.linear(flat, S.pack([p3.mlp_pred[0][0]])))
    h = S.gelu(S.linear(h, S.pack([p3.mlp_pred[1][0]])))
    mlp_pad = S.linear(h, S.pack([p3.mlp_pred[2]]), out_pad=True)             # (B, 64), 48 valid
    anchors_3d = mlp_pad[:, :3 * J].clone().view(B, J, 3)                      # init_anchors_3d = mlp_pred.clone().detach(): a real copy
    # (syn mode mutates the anchors in place, F7; for B = 1 the slice is contiguous and .contiguous() would alias mlp_pad)
    ctm32 = None
    if p3.camera_model.startswith("ego4view_rw"):
        if ctm is None:
            raise RuntimeError("egorear_amd: camera_model ego4view_rw needs coord_trans_mat (B,4,4,4)")
        ctm32 = ctm.to(device=dev, dtype=torch.float32).contiguous()
    cams = p3.__dict__.get("_egr_cams")
    if cams is None or cams.device != dev:
        cams = torch.from_numpy(np.stack([c.packed() for c in p3.cameras()])).to(dev)
        p3.__dict__["_egr_cams"] = cams
    anchors_2d, valid, q4 = hip.fisheye_project(anchors_3d, ctm32, cams)       # syn: anchors_3d mutated in place (F7)
    a3_pad = T.planes_to_nhwc(anchors_3d, NMap(B * J, 3, 0), B * J, 3, 1, 32).view(B * J, 32)
    q4p = T.planes_to_nhwc(q4, NMap(B * J, 4, 0), B * J, 4, 1, 32).view(B * J, 32)
    qg = p3.query_gen_mlp
    x = S.linear(q4p, S.pack([qg[0]], need_dx=False), ACT_RELU, need_dx=False)
    x = S.linear(x, S.pack([qg[2]]), ACT_RELU)
    x = S.linear(x, S.pack([qg[4]]))
    fp = p3.feat_proj
    preds = [mlp_pad]
    for i, layer in enumerate(p3.layers):
        L = _pack_layer(S, [layer], [(fp.weight, fp.bias)], [None])
        x = layer_train(S, L, x, memory, anchors_2d, valid, B, V, J, hgt, wid, dmem)
        xn = S.layernorm(x, [p3.post_norm[i]])
        r = S.gelu(S.linear(xn, S.pack([p3.reg_mlp[i][0]])))
        y = S.linear(r, S.pack([p3.reg_mlp[i][2]]), out_pad=True)               # (B*J, 32), 3 valid
        preds.append(S.add(y, None, b_const=a3_pad))                            # offset + init_anchors_3d (detached)
    aux = {"anchors_2d": anchors_2d, "anchors_valid": valid, "anchors_3d_after": anchors_3d}
    return preds, aux


# ---- whole network -------------------------------------------------------------------------------------------------------

def forward_train(S: Step, net, img: torch.Tensor, ctm=None):
    """EgoPoseFormerMVFEX.forward in training mode.  Returns (pred buffers, [hm_init, hm_ref] as (B,V,15,64,64), aux)."""
    he, p3 = net.heatmap_estimator, net.pose3d_estimator
    if not (he.full_training and he.use_pred_heatmap_init and he.detach_heatmap_feat and not he.no_detach_feat_init and not he.use_1by1_conv):
        raise NotImplementedError("egorear_amd.train: the training path follows the shipped pose3d configs "
                                  "(full_training, use_pred_heatmap_init, detach_heatmap_feat, non-1x1 heat-map heads)")
    B, V = img.shape[:2]
    img = img.contiguous()
    H4, W4 = img.shape[3] // 4, img.shape[4] // 4
    J = he.num_heatmap
    dev = img.device
    front, back = he.heatmap_estimator_stereo_front, he.heatmap_estimator_stereo_back
    # Two streams (Step.overlap): under these flags the initial heads run on detached features, the refiners on detached heat maps and
    # memory, the refined heads on detached refined features (:273, :297, :717-721) - in the reverse pass they are leaves that produce
    # parameter gradients only.  The gradient that reaches the encoders comes from the lifting head's sampling alone.
    S.allow_overlap = True
    S.G.track = S.overlap()
    feat_all, s32_all = backbone_train(S, [front.encoder, back.encoder], img, 0, 2)
    hm_init = torch.empty((B, V, J, H4, W4), device=dev, dtype=torch.float32)
    S.mark_stage(2)
    S.bwd_side = S.bwd_early = True      # reverse pass: seed-only leaf
    heatmap_head_train(S, [he.conv_heatmap_layers_stereo_front, he.conv_heatmap_layers_stereo_back], feat_all, hm_init, B, V)
    S.bwd_side = S.bwd_early = False
    a, mv, vd, idx = hip.argmax_rows(hm_init, he.heatmap_threshold)
    anchors, valid = a.view(B, V, J, 2), vd.view(B, V, J)
    hm_ref = torch.empty_like(hm_init)
    S.bwd_side = True                    # reverse pass: the refiners are a leaf behind d(feat_ref)
    feat_ref = refiners_train(S, he.refiners(), B, V, hm_init, feat_all, s32_all, anchors, valid, hm_ref)
    S.bwd_side = False
    preds, aux_p = pose3d_train(S, p3, feat_all, feat_ref, B, V, ctm)
    S.join_side()                        # the refined heads' forward ran beside the lifting head's: hm_ref is complete from here on
    aux = {"heatmap": {"anchors_2d": anchors, "anchors_valid": valid, "argmax_idx": idx.view(B, V, J), "maxvals": mv.view(B, V, J)},
           "pose3d": aux_p}
    S.flush_buffers()
    return preds, [hm_init, hm_ref], aux


W_MPJPE, W_HEATMAP = 0.1, 10.0      # configs/*_pose3d.yaml: w_mpjpe, w_heatmap


def loss_and_seed(S: Step, preds, hms, gt_pose: torch.Tensor, gt_heatmap: torch.Tensor, w_mpjpe: float = W_MPJPE, w_heatmap: float = W_HEATMAP,
                  grad_scale: float = 1.0):
    """The wrapper's training loss (pose_3d_mvf_ex.py:133-145) and the gradients it sends into the network outputs.
    Per-term values end up in S.loss_terms (6 device float64).  grad_scale multiplies the seeded gradients only (1/world_size
    in data-parallel runs, so that a SUM all-reduce of the parameter gradients is DDP's average)."""
    w_mpjpe, w_heatmap = w_mpjpe * grad_scale, w_heatmap * grad_scale
    B, J = gt_pose.shape[:2]
    gt_pose = gt_pose.to(device=S.dev, dtype=torch.float32).contiguous()
    gt_heatmap = gt_heatmap.to(device=S.dev, dtype=torch.float32).contiguous()
    terms = torch.zeros(len(preds) + len(hms), dtype=torch.float64, device=S.dev)
    for i, p in enumerate(preds):
        if i == 0:    # (B, 64): 16 joints x 3 inside each padded row
            d = T.rownorm_loss(p, gt_pose, 3, w_mpjpe, terms[i:i + 1], rows=B * J, inner=J, ld_pred=p.shape[1], ld_gt=3 * J)
        else:         # (B*J, 32)
            d = T.rownorm_loss(p, gt_pose, 3, w_mpjpe, terms[i:i + 1], rows=B * J, inner=1, ld_pred=p.shape[1], ld_gt=3)
        S.G.add(p, d)
    V = gt_heatmap.shape[1]
    for i, h in enumerate(hms):   # sum over views of mean over (b, j, row) == V * mean over all rows
        k = len(preds) + i
        d = T.rownorm_loss(h, gt_heatmap, h.shape[-1], w_heatmap * V, terms[k:k + 1])
        S.G.add(h, d)
    S.loss_terms = terms if grad_scale == 1.0 else terms / grad_scale
    return S.loss_terms


def forward_backward(net, img, ctm, gt_pose, gt_heatmap):
    """One training forward + backward.  Returns (Step with .pgrads / .loss_terms, outputs)."""
    if not img.is_cuda:
        raise RuntimeError("egorear_amd.train: HIP device tensors only (no CPU path)")
    S = Step(net, img.device)
    S.record = True
    with torch.no_grad():
        preds, hms, aux = forward_train(S, net, img, ctm)
        loss_and_seed(S, preds, hms, gt_pose, gt_heatmap)
        _finish_backward(S)
    from .engine import invalidate
    for m in (net, net.heatmap_estimator, net.pose3d_estimator):
        invalidate(m)
    B = img.shape[0]
    J = net.pose3d_estimator.num_joints
    outs = [preds[0][:, :3 * J].reshape(B, J, 3)] + [p[:, :3].reshape(B, J, 3) for p in preds[1:]]
    return S, (outs, hms, aux)


# --------------------------------------------------------------------------- optimiser + the full step

N_GRAD_STAGES = 4


def grad_stage(name: str) -> int:
    """Order in which the reverse pass completes parameter gradients: lifting head, refiners, initial heat-map heads,
    encoders (the forward runs them the other way round)."""
    if name.startswith("pose3d_estimator."):
        return 0
    if name.startswith("heatmap_estimator.heatmap_refiner_"):
        return 1
    if name.startswith("heatmap_estimator.conv_heatmap_layers_stereo_"):
        return 2
    return 3


def is_no_decay(name: str) -> bool:
    """Parameter-group rule of the reference's configure_optimizers (pose_3d_mvf_ex.py:223)."""
    return ("norm" in name) or ("bn" in name) or ("ln" in name) or ("bias" in name)


def flat_layout(named, decay_all: bool = False):
    """Layout of the flat parameter / gradient / moment buffers for `named` = [(name, tensor-like with .numel())]: ordered by the
    stage of the reverse pass that finishes a tensor's gradient (so a stage is ONE contiguous all-reduce bucket that can start
    while earlier layers are still being differentiated), then [no-decay | decayed]; every slot starts 16-byte aligned.
    Returns (ordered list, slots [(name, offset, numel, decay)], stage_range {stage: [begin, end)}, total elements)."""
    nd = (lambda k: False) if decay_all else is_no_decay     # decay_all: one AdamW group (the heat-map stages' configure_optimizers)
    order = sorted(named, key=lambda kp: (grad_stage(kp[0]), not nd(kp[0])))
    slots, stage_range, off = [], {}, 0
    for k, p in order:
        rng = stage_range.setdefault(grad_stage(k), [off, off])
        slots.append((k, off, p.numel(), not nd(k)))
        off += (p.numel() + 3) // 4 * 4
        rng[1] = off
    return order, slots, stage_range, off


class FusedAdamW:
    """AdamW over ONE flat fp32 buffer that the module's parameters are re-homed into (their .data become views), with
    flat gradient / moment buffers of the same layout: [no-decay parameters | decayed parameters].  A step is a gradient
    sum-of-squares, then one fused clip + AdamW launch per contiguous run of updated parameters (tensors that received no
    gradient are skipped entirely, like torch's `grad is None`), and in multi-process runs ONE all-reduce of the flat
    gradient over RCCL instead of one per tensor.

    Matches torch.optim.AdamW(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay) + clip_grad_norm_(clip) with the reference's
    two parameter groups (pose_3d_mvf_ex.py:219-234) and its warm-up hook (:212-217).  The hook runs inside Lightning's
    `optimizer_step`, after `optimizer.step()` and BEFORE Lightning counts the step as completed, so during update t it sees
    `trainer.global_step == t - 1` and leaves lr * min(1, t / warmup_iters) behind for update t + 1: update 1 runs at the full
    lr, update t >= 2 at lr * min(1, (t - 1) / warmup_iters) - update 2 at 1 / warmup_iters."""

    def __init__(self, net: nn.Module, lr: float = 1e-3, weight_decay: float = 5e-4, clip: float = 5.0, warmup_iters: int = 500,
                 betas=(0.9, 0.999), eps: float = 1e-8, process_group=None, decay_all: bool = False):
        self.net, self.lr, self.wd, self.clip, self.warmup, self.betas, self.eps = net, lr, weight_decay, clip, warmup_iters, betas, eps
        self.pg = process_group
        self.force_collective = False        # issue the stage all-reduces for a one-rank group too (RCCL rehearsal on one GPU)
        named = list(net.named_parameters())
        dev = named[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("egorear_amd.train: parameters must live on the HIP device")
        order, self.slots, self.stage_range, off = flat_layout(named, decay_all)
        self.total = off
        self.pending = []                    # async all-reduce handles of this step
        self.flat_p = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(off, device=dev, dtype=torch.float32)
        self.m = torch.zeros(off, device=dev, dtype=torch.float32)
        self.v = torch.zeros(off, device=dev, dtype=torch.float32)
        self.gviews: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            for (k, o, n, _), (_, p) in zip(self.slots, order):
                self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                self.gviews[k] = self.flat_g[o:o + n].view(p.shape)
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        self.hyper = torch.zeros(4, dtype=torch.float32, device=dev)        # {lr, 1 - b1^t, sqrt(1 - b2^t)} of the coming update
        self.steps = 0
        self.lr_scale_epoch = 1.0            # MultiStepLR(lr_decay_epochs, 0.1) factor, set by the caller per epoch

    def lr_at(self, t: int) -> float:
        return self.lr * self.lr_scale_epoch * (1.0 if t <= 1 else min(1.0, float(t - 1) / float(self.warmup)))

    def _runs(self, have):
        runs, cur = [], None
        for k, o, n, decay in self.slots:
            n4 = (n + 3) // 4 * 4
            if k not in have:
                cur = None
                continue
            if cur is not None and cur[2] == decay and cur[0] + cur[1] == o:
                cur[1] += n4
            else:
                cur = [o, n4, decay]
                runs.append(cur)
        return runs

    def reduce_stage(self, stage: int):
        """Start the gradient all-reduce of one finished stage (SUM; the 1/world of DDP's average is in the loss seed).
        It runs on the communication stream of the process group while the reverse pass continues."""
        from .dist import allreduce_gradients_
        b, e = self.stage_range[stage]
        h = allreduce_gradients_(self.flat_g[b:e], self.pg, async_op=True, force=self.force_collective)
        if h is not None:
            self.pending.append(h)

    def begin_update(self):
        """Count the coming update and put its step-dependent scalars where the kernels read them (device memory, so the
        same enqueued / captured launches serve every step)."""
        self.steps += 1
        t = self.steps
        # by kernel argument, not by a pinned-buffer copy: the host may be several replays ahead of the device
        T.set4(self.hyper, self.lr_at(t), 1.0 - self.betas[0] ** t, math.sqrt(1.0 - self.betas[1] ** t))

    def enqueue_update(self, have) -> None:
        """sumsq + clip + AdamW launches for the names whose gradient views were written (every stage already reduced)."""
        for h in self.pending:
            h.wait()
        self.pending = []
        runs = self._runs(have)
        T.sumsq(self.flat_g, self.sumsq)       # one pass over the whole flat buffer: slots that never receive a gradient stay zero
        for o, n, decay in runs:
            T.adamw_dev(self.flat_p[o:o + n], self.flat_g[o:o + n], self.m[o:o + n], self.v[o:o + n], self.hyper, self.betas[0], self.betas[1],
                        self.eps, self.wd if decay else 0.0, self.sumsq, self.clip)

    def step(self, have) -> None:
        """`have`: names whose gradient views were written this step.  Every stage must have been reduced (reduce_stage)."""
        self.begin_update()
        self.enqueue_update(have)

    def grad_norm(self) -> float:
        return float(torch.sqrt(self.sumsq))


class Trainer:
    """The native optimisation step of config 5: forward (training mode) + losses + backward + (all-reduce) + clip + AdamW.

    use_graph: after two eager steps the whole step (~2600 kernel launches and copies) is captured into ONE hipGraph on
    static input buffers and replayed, which takes the host out of the loop (eager: ~45 ms of enqueue work per step, more
    than the kernels take).  With a process group the step is captured as one graph PER GRADIENT STAGE: the collectives stay
    ordinary eager calls between the replays (nothing of RCCL is captured), so each stage's all-reduce still starts as soon
    as its gradients are complete and overlaps with the replay of the stages that follow."""

    def __init__(self, net: nn.Module, lr: float = 1e-3, weight_decay: float = 5e-4, clip: float = 5.0, warmup_iters: int = 500,
                 w_mpjpe: float = W_MPJPE, w_heatmap: float = W_HEATMAP, process_group=None, use_graph: bool = False):
        self.net = net
        self.opt = FusedAdamW(net, lr, weight_decay, clip, warmup_iters, process_group=process_group)
        self.w_mpjpe, self.w_heatmap = w_mpjpe, w_heatmap
        self.use_graph = use_graph
        self.graph = None
        self._eager_done = 0
        self._static = None
        self._graph_out = None
        self._graph_step = None     # keeps the captured step's tensors / pinned tables alive
        self._cuts = None           # multi-process capture: gradient stage handed to the all-reduce after each graph segment
        # DDP side channels: rank 0's BatchNorm buffers are broadcast before every forward
        # (DDP broadcast_buffers=True), and logged scalars can be averaged over the ranks (Lightning sync_dist=True)
        from .dist import BufferSync, world_size
        self.buffers = BufferSync(net, process_group) if world_size(process_group) > 1 else None

    def _distributed(self) -> bool:
        from .dist import world_size
        return world_size(self.opt.pg) > 1 or self.opt.force_collective

    def _run(self, img, ctm, gt_pose, gt_heatmap, update: bool, hook=None):
        net = self.net
        S = Step(net, img.device)
        S.gviews = self.opt.gviews
        from .dist import grad_seed_scale
        if hook is not None:
            S.stage_hook = hook                    # capture: cuts the recording at every stage boundary
        elif self._distributed():
            S.stage_hook = self.opt.reduce_stage   # bucketed all-reduce overlapped with the rest of the reverse pass
        with torch.no_grad():
            preds, hms, aux = forward_train(S, net, img, ctm)
            loss_and_seed(S, preds, hms, gt_pose, gt_heatmap, self.w_mpjpe, self.w_heatmap, grad_scale=grad_seed_scale(self.opt.pg))
            _finish_backward(S)
            if update:
                self.opt.begin_update()
            self.opt.enqueue_update(S.pgrads.keys())
        return S, (preds, hms, aux)

    def _invalidate(self):
        from .engine import invalidate
        for m in (self.net, self.net.heatmap_estimator, self.net.pose3d_estimator):
            invalidate(m)

    def sync_buffers(self, force: bool = False):
        """Rank 0's module buffers to every rank (what DDP does at the start of each forward); a no-op for one process."""
        if self.buffers is None and force:
            from .dist import BufferSync
            self.buffers = BufferSync(self.net, self.opt.pg)
        if self.buffers is not None:
            self.buffers.broadcast(0, force=force)

    def mean_over_ranks(self, terms: torch.Tensor) -> torch.Tensor:
        """A copy of `terms` averaged over the data-parallel ranks: the value `self.log(..., sync_dist=True)` records
        (pose_3d_mvf_ex.py:208).  `step` itself returns this rank's terms, like the reference's un-synced training log."""
        from .dist import allreduce_mean_
        return allreduce_mean_(terms.clone(), self.opt.pg)

    def step(self, img, ctm, gt_pose, gt_heatmap):
        """Returns (loss terms (6,) float64 device tensor, outputs).  Parameters are updated in place."""
        from .dist import world_size
        self.sync_buffers(force=self.opt.force_collective)   # eager, in front of the (possibly replayed) forward
        if self.graph is not None:
            same = all((a is None) == (b is None) and (a is None or a.shape == b.shape) for a, b in zip(self._static, (img, ctm, gt_pose, gt_heatmap)))
            if same:
                for dst, src in zip(self._static, (img, ctm, gt_pose, gt_heatmap)):
                    if dst is not None:
                        dst.copy_(src, non_blocking=True)
                self.opt.begin_update()
                if self._cuts is None:
                    self.graph.replay()
                else:   # one segment per gradient stage; its all-reduce starts behind it and overlaps with the next segments
                    for i, g in enumerate(self.graph):
                        if i == len(self._cuts):        # the update segment: every stage must have arrived
                            for h in self.opt.pending:
                                h.wait()
                            self.opt.pending = []
                        g.replay()
                        if i < len(self._cuts):
                            self.opt.reduce_stage(self._cuts[i])
                self._invalidate()
                return self._graph_out
        if self.use_graph and self.graph is None and self._eager_done >= 2:
            try:
                self._capture(img, ctm, gt_pose, gt_heatmap)
            except Exception as exc:      # capture is an optimisation: any refusal leaves the eager path in charge - loudly
                import warnings
                warnings.warn(f"egorear_amd.train: hipGraph capture of the step failed ({type(exc).__name__}: {exc}); continuing eagerly")
                self.graph, self.use_graph = None, False
                torch.cuda.synchronize()
            else:
                return Trainer.step(self, img, ctm, gt_pose, gt_heatmap)
        S, outs = self._run(img, ctm, gt_pose, gt_heatmap, update=True)
        self._eager_done += 1
        self._invalidate()
        return S.loss_terms, outs

    def _capture(self, img, ctm, gt_pose, gt_heatmap):
        dev = img.device
        self._static = [None if t is None else t.detach().to(device=dev, dtype=torch.float32).clone().contiguous()
                        for t in (img, ctm, gt_pose, gt_heatmap)]
        torch.cuda.synchronize()
        if not self._distributed():
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="relaxed"):
                S, outs = self._run(*self._static, update=False)       # recorded, not executed: no update is counted
            self.graph, self._graph_step, self._graph_out = g, S, (S.loss_terms, outs)
            return
        # multi-process: the recording is cut wherever a gradient stage is complete (the stage hook); segment i ends with
        # stage cuts[i] flushed into the flat gradient buffer, the last segment is norm + clip + AdamW.  All segments share
        # one memory pool and are replayed in recording order.
        pool = torch.cuda.graph_pool_handle()
        segs, cuts, cur = [], [], {}

        def begin():
            cur["g"] = torch.cuda.CUDAGraph()
            cur["ctx"] = torch.cuda.graph(cur["g"], pool=pool, capture_error_mode="relaxed")
            cur["ctx"].__enter__()

        def end():
            cur["ctx"].__exit__(None, None, None)
            segs.append(cur["g"])

        def cut(stage):
            end()
            cuts.append(stage)
            begin()

        begin()
        try:
            S, outs = self._run(*self._static, update=False, hook=cut)
        except BaseException:
            cur["ctx"].__exit__(None, None, None)
            raise
        end()
        self.graph, self._cuts, self._graph_step, self._graph_out = segs, cuts, S, (S.loss_terms, outs)


class HeatmapTrainer(Trainer):
    """The native optimisation step of the reference's stages 1 and 2 (pl_wrappers/egoposeformer/heatmap.py:94-110, 144-154, 215-218 and
    heatmap_mvf_ex.py:104-127, 187-197, 258-261): training-mode forward of EgoPoseFormerHeatmap (one stereo pair's encoder + head) or
    of EgoPoseFormerHeatmapMVFEX in its stage-2 configuration (encoders frozen under no_grad, heads and refiners trained), the loss
    sum over views of w_heatmap * nn.MSELoss(mean) - per heat-map set in stage 2 (heatmap_loss_0 = initial, heatmap_loss_1 = refined) -
    backward, gradient-norm clip (trainer.gradient_clip_val 5.0) and ONE AdamW group over all parameters (lr 1e-3, weight decay
    5e-3, the warm-up hook of the wrappers).  Graph replay, the bucketed all-reduce and the BatchNorm buffer broadcast are the
    Trainer's."""

    def __init__(self, net: nn.Module, lr: float = 1e-3, weight_decay: float = 5e-3, clip: float = 5.0, warmup_iters: int = 500,
                 w_heatmap: float = 10.0, process_group=None, use_graph: bool = False):
        from .estimator import EgoPoseFormerHeatmap, EgoPoseFormerHeatmapMVFEX
        if isinstance(net, EgoPoseFormerHeatmapMVFEX):
            self.kind = "mvfex"
        elif isinstance(net, EgoPoseFormerHeatmap):
            self.kind = "heatmap"
        else:
            raise RuntimeError("egorear_amd.train.HeatmapTrainer: EgoPoseFormerHeatmap or EgoPoseFormerHeatmapMVFEX expected")
        self.net = net
        self.opt = FusedAdamW(net, lr, weight_decay, clip, warmup_iters, process_group=process_group, decay_all=True)
        self.w_mpjpe, self.w_heatmap = 0.0, w_heatmap
        self.use_graph = use_graph
        self.graph = None
        self._eager_done = 0
        self._static = self._graph_out = self._graph_step = self._cuts = None
        from .dist import BufferSync, world_size
        self.buffers = BufferSync(net, process_group) if world_size(process_group) > 1 else None

    def _run(self, img, ctm, gt_pose, gt_heatmap, update: bool, hook=None):
        net = self.net
        S = Step(net, img.device)
        S.gviews = self.opt.gviews
        from .dist import grad_seed_scale
        if hook is not None:
            S.stage_hook = hook
        elif self._distributed():
            S.stage_hook = self.opt.reduce_stage
        scale = grad_seed_scale(self.opt.pg)
        with torch.no_grad():
            if self.kind == "heatmap":
                hms = [heatmap_forward_train(S, net, img)]
                outs = hms[0]
            else:
                hms, feat_all, feat_ref = mvfex_heatmap_forward_train(S, net, img)
                outs = (hms, feat_all, feat_ref)
            gt = gt_heatmap.to(device=S.dev, dtype=torch.float32).contiguous()
            terms = torch.zeros(len(hms), dtype=torch.float64, device=S.dev)
            for i, h in enumerate(hms):      # sum over views of the per-view mean == V * the mean over everything (equal-sized views)
                V = h.shape[1]               # (the stage-1 model sees one stereo pair: views [0, V) of the ground truth)
                g = gt if gt.shape[1] == V else gt[:, :V].contiguous()
                S.G.add(h, T.mse_loss(h, g, self.w_heatmap * V * scale, terms[i:i + 1]))
            S.loss_terms = terms if scale == 1.0 else terms / scale
            _finish_backward(S)
            if update:
                self.opt.begin_update()
            self.opt.enqueue_update(S.pgrads.keys())
        return S, outs

    def _invalidate(self):
        from .engine import invalidate
        mods = [self.net]
        if self.kind == "mvfex":
            mods += [self.net.heatmap_estimator_stereo_front, self.net.heatmap_estimator_stereo_back]
        for m in mods:
            invalidate(m)

    def step(self, img, gt_heatmap):      # noqa: D401 - (loss terms: (1,) in stage 1, (2,) in stage 2; outputs)
        """Returns (loss terms float64 device tensor, outputs).  Parameters are updated in place."""
        return Trainer.step(self, img, None, None, gt_heatmap)


# --------------------------------------------------------------------------- autograd bridge (drop-in for the wrapper)

def _finish_backward(S: Step):
    S.cache.ready = True       # every pack of the network exists now: later steps refresh them with one launch
    S.backward()
    S.finish_param_grads()
    if S.stage_hook is not None:
        S.stage_hook(N_GRAD_STAGES - 1)    # the encoders finish last


def _mvfex_grad_free(name: str) -> bool:
    """Parameters of EgoPoseFormerMVFEX that the reference's graph never reaches under the shipped pose3d flags (_FLAGS_POSE3D): the stereo
    estimators' own conv_heatmap (the MVFEx model only takes their features, heatmap_mvf_ex.py:212-234) and every refiner's
    frame_feat_proj_layers (`offset_pred + frame_feat.detach()`, :715).  torch leaves their .grad None, AdamW skips them, and
    DistributedDataParallel(find_unused_parameters=True) - the strategy the wrapper's Trainer uses - must find them UNUSED: a tensor
    that is an input of the autograd node but receives no gradient would be reduced as zeros and then decayed by AdamW.
    The rule is tied to those flags: forward_train refuses every other flag combination with NotImplementedError BEFORE anything runs
    (so a configuration the rule does not describe never reaches backward), and backward re-checks it against the tape."""
    return (".conv_heatmap." in name and "heatmap_estimator_stereo_" in name) or ".frame_feat_proj_layers." in name


class _MVFEXTrainFn(torch.autograd.Function):
    """One autograd node for the whole network: forward = training-mode forward on the HIP kernels, backward = the taped
    reverse pass.  Inputs after `ctm` are the module's parameters that the reverse pass reaches (so autograd routes .grad to
    them); the others are not part of the graph, exactly like the reference's untouched parameters."""

    @staticmethod
    def forward(ctx, net, img, ctm, *params):
        S = Step(net, img.device)
        preds, hms, aux = forward_train(S, net, img, ctm)
        B, J = img.shape[0], net.pose3d_estimator.num_joints
        ctx.S, ctx.preds, ctx.hms, ctx.names, ctx.BJ = S, preds, hms, [S.name(p) for p in params], (B, J)
        ctx.set_materialize_grads(False)
        net.__dict__["_egr_last_aux"] = aux
        outs = [preds[0][:, :3 * J].reshape(B, J, 3)] + [p[:, :3].reshape(B, J, 3) for p in preds[1:]]
        return tuple(outs) + tuple(hms)

    @staticmethod
    def backward(ctx, *grads):
        S, (B, J) = ctx.S, ctx.BJ
        n = len(ctx.preds)
        for i, (buf, g) in enumerate(zip(ctx.preds, grads[:n])):
            if g is None:
                continue
            g = g.to(torch.float32).contiguous()
            c = 3 * J if i == 0 else 3
            rows = B if i == 0 else B * J
            S.G.add(buf, T.planes_to_nhwc(g, NMap(rows, c, 0), rows, c, 1, buf.shape[1]).view(buf.shape))
        for h, g in zip(ctx.hms, grads[n:]):
            if g is not None:
                S.G.add(h, g.to(torch.float32).contiguous())
        _finish_backward(S)
        pg = S.pgrads
        ctx.S = None
        if set(pg) != set(ctx.names):      # the static rule (_mvfex_grad_free) and the tape disagree: refuse rather than mis-report to DDP
            raise RuntimeError(f"egorear_amd.train: the reverse pass reached {len(pg)} parameters, the autograd node declares {len(ctx.names)}: "
                               f"{sorted(set(pg) ^ set(ctx.names))[:4]}")
        return (None, None, None) + tuple(pg.get(k) for k in ctx.names)


def mvfex_training_forward(net, img, ctm=None):
    """EgoPoseFormerMVFEX.forward in training mode -> (list_pred_pose3d, list_pred_heatmap) like the reference."""
    from .engine import _check_input, invalidate
    _check_input(img, net)
    from .engine import as_rgb
    img = as_rgb(img)                      # (B, V, H, W) grayscale frames: resnet.py:44-46
    for m in (net, net.heatmap_estimator, net.pose3d_estimator):
        invalidate(m)                      # parameters are about to change: drop the inference packs
    params = [p for p in net.parameters()]
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        if not all(p.requires_grad for p in params):
            raise NotImplementedError("egorear_amd.train: partially frozen parameter sets are not supported (the reference trains all of them)")
        outs = _MVFEXTrainFn.apply(net, img, ctm, *[p for k, p in net.named_parameters() if not _mvfex_grad_free(k)])
        return list(outs[:4]), list(outs[4:])
    with torch.no_grad():
        S = Step(net, img.device)
        S.record = False
        preds, hms, aux = forward_train(S, net, img, ctm)
        net.__dict__["_egr_last_aux"] = aux
        B, J = img.shape[0], net.pose3d_estimator.num_joints
        return [preds[0][:, :3 * J].reshape(B, J, 3)] + [p[:, :3].reshape(B, J, 3) for p in preds[1:]], hms


# --------------------------------------------------------------------------- the two heat-map training stages

_FLAGS_POSE3D = (True, True, True, False)     # full_training, use_pred_heatmap_init, detach_heatmap_feat, no_detach_feat_init
_FLAGS_STAGE2 = (False, False, False, False)  # constructor defaults = configs/*_heatmap_mvfex-n1_jqa.yaml


def heatmap_forward_train(S: Step, net, img: torch.Tensor) -> torch.Tensor:
    """EgoPoseFormerHeatmap.forward in training mode (egoposeformer_heatmap.py:29-44; stage 1 of the reference's schedule)."""
    B, V = img.shape[:2]
    img = img.contiguous()
    feat, _ = backbone_train(S, [net.encoder], img, 0, V)
    hm = torch.empty((B, V, net.num_heatmap, feat.shape[1], feat.shape[2]), device=img.device, dtype=torch.float32)
    conv_to_planes(S, feat, S.pack([net.conv_heatmap], need_dx=not net.detach_heatmap_feat_init), hm, B, V,
                   need_dx=not net.detach_heatmap_feat_init)
    S.flush_buffers()
    return hm


def mvfex_heatmap_forward_train(S: Step, he, img: torch.Tensor):
    """EgoPoseFormerHeatmapMVFEX.forward in training mode for the stage-2 configuration (heatmap_mvf_ex.py:236-352 with the
    constructor defaults): encoders under no_grad but with train()-mode BatchNorm, initial heads and refiners trained, the
    initial heat maps feed the refiners' heatmap_proj WITH gradient, refined heads see un-detached refined features.
    Returns ([hm_init, hm_ref], feat_all, feat_ref)."""
    flags = (he.full_training, he.use_pred_heatmap_init, he.detach_heatmap_feat, he.no_detach_feat_init)
    if flags != _FLAGS_STAGE2 or he.use_1by1_conv:
        raise NotImplementedError("egorear_amd.train: EgoPoseFormerHeatmapMVFEX trains standalone with the shipped stage-2 flags only "
                                  "(full_training / use_pred_heatmap_init / detach_heatmap_feat / no_detach_feat_init all False)")
    B, V = img.shape[:2]
    img = img.contiguous()
    H4, W4 = img.shape[3] // 4, img.shape[4] // 4
    J = he.num_heatmap
    front, back = he.heatmap_estimator_stereo_front, he.heatmap_estimator_stereo_back
    rec = S.record
    S.record = False                     # `with torch.no_grad():` around the encoders (:267-268); BatchNorm still updates its buffers
    feat_all, s32_all = backbone_train(S, [front.encoder, back.encoder], img, 0, 2)
    S.record = rec
    hm_init = torch.empty((B, V, J, H4, W4), device=img.device, dtype=torch.float32)
    heatmap_head_train(S, [he.conv_heatmap_layers_stereo_front, he.conv_heatmap_layers_stereo_back], feat_all, hm_init, B, V)
    a, mv, vd, idx = hip.argmax_rows(hm_init, he.heatmap_threshold)
    hm_ref = torch.empty_like(hm_init)
    feat_ref = refiners_train(S, he.refiners(), B, V, hm_init, feat_all, s32_all, a.view(B, V, J, 2), vd.view(B, V, J), hm_ref,
                              hm_grad=True, detach_heatmap_feat=False)
    he.__dict__["_egr_last_aux"] = {"anchors_2d": a.view(B, V, J, 2), "anchors_valid": vd.view(B, V, J), "argmax_idx": idx.view(B, V, J),
                                    "maxvals": mv.view(B, V, J)}
    S.flush_buffers()
    return [hm_init, hm_ref], feat_all, feat_ref


class _HeatmapTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, img, *params):
        S = Step(net, img.device)
        hm = heatmap_forward_train(S, net, img)
        ctx.S, ctx.hm, ctx.names = S, hm, [S.name(p) for p in params]
        return hm

    @staticmethod
    def backward(ctx, g):
        S = ctx.S
        S.G.add(ctx.hm, g.to(torch.float32).contiguous())
        _finish_backward(S)
        pg = S.pgrads
        ctx.S = None
        return (None, None) + tuple(pg.get(k) for k in ctx.names)


class _HeatmapMVFEXTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, he, img, *params):
        from .engine import _vb_view
        S = Step(he, img.device)
        hms, feat_all, feat_ref = mvfex_heatmap_forward_train(S, he, img)
        B, V = img.shape[:2]
        fa, fr = _vb_view(feat_all, V, B), _vb_view(feat_ref, V, B)
        ctx.S, ctx.hms, ctx.names = S, hms, [S.name(p) for p in params]
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(fa, fr)      # the stage-2 wrapper trains on the heat maps only
        return hms[0], hms[1], fa, fr

    @staticmethod
    def backward(ctx, g0, g1, *_):
        S = ctx.S
        for h, g in zip(ctx.hms, (g0, g1)):
            if g is not None:
                S.G.add(h, g.to(torch.float32).contiguous())
        _finish_backward(S)
        pg = S.pgrads
        ctx.S = None
        return (None, None) + tuple(pg.get(k) for k in ctx.names)


def _training_entry(net, img, fn, eager):
    from .engine import _check_input, invalidate
    _check_input(img, net)
    from .engine import as_rgb
    img = as_rgb(img)                      # (B, V, H, W) grayscale frames: resnet.py:44-46
    invalidate(net)
    params = [p for p in net.parameters()]
    if torch.is_grad_enabled() and any(p.requires_grad for p in params):
        if not all(p.requires_grad for p in params):
            raise NotImplementedError("egorear_amd.train: partially frozen parameter sets are not supported")
        return fn.apply(net, img, *params)
    with torch.no_grad():
        S = Step(net, img.device)
        S.record = False
        return eager(S)


def heatmap_training_forward(net, img, return_feat=False):
    """EgoPoseFormerHeatmap.forward in train() mode -> heat maps (B, V, 15, 64, 64)."""
    if return_feat:
        raise NotImplementedError("egorear_amd.train: return_feat is an inference-time option")
    return _training_entry(net, img, _HeatmapTrainFn, lambda S: heatmap_forward_train(S, net, img))


def heatmap_mvfex_training_forward(he, img, heatmap_for_anchor=None):
    """EgoPoseFormerHeatmapMVFEX.forward in train() mode -> ([hm_init, hm_refined], [feat_init, feat_refined])."""
    if heatmap_for_anchor is not None:
        raise NotImplementedError("egorear_amd.train: heatmap_for_anchor is an inference-time option")
    from .engine import _vb_view
    B, V = img.shape[:2]

    def eager(S):
        hms, fa, fr = mvfex_heatmap_forward_train(S, he, img)
        return hms[0], hms[1], _vb_view(fa, V, B), _vb_view(fr, V, B)
    h0, h1, fa, fr = _training_entry(he, img, _HeatmapMVFEXTrainFn, eager)
    return [h0, h1], [fa, fr]
