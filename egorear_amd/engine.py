"""Execution of the hot path on the HIP kernels.

Layout in HBM: every activation is fp32 channels-last, *view-major*:
(V, B, H, W, C).  The four camera views of a batch are four contiguous image
batches, so a stereo estimator (views 0-1 / 2-3), one refiner's own view, and the
4-view memory the deformable attention samples are all plain slices — no gather,
transpose or torch.cat anywhere on the path.  The reference's batch-major NCHW only
exists at the boundary: the stem kernel reads the (B,V,3,H,W) input through an image
map, the 15-channel heatmap convs write (B,V,15,64,64) planes through one, and feature
tensors handed back to callers are permuted *views* of the view-major buffers.

Weights are re-packed once per module/device (conv OIHW -> [cout_pad][(kh,kw,ci)],
BatchNorm as per-channel scale/shift applied in the conv epilogue, q/k/v and
offset/logit projections concatenated, the deformable-attention value path folded for
the sample-then-project form) and cached; `invalidate(module)` drops the cache.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from . import hip
from .hip import ACT_GELU, ACT_NONE, ACT_RELU, RES_AFTER_ACT, RES_BEFORE_ACT, RES_NONE, Img, NMap

_WORKSPACE_FLOATS = 16 << 20  # split-K partial slabs (64 MB)


# --------------------------------------------------------------------------- packed parameters

class PConv:
    __slots__ = ("w", "scale", "shift", "cout", "cin", "kh", "kw", "stride", "pad")


def _pad_rows(w2d: torch.Tensor) -> torch.Tensor:
    cout = w2d.shape[0]
    npad = (cout + 31) // 32 * 32
    if npad == cout:
        return w2d.contiguous()
    out = torch.zeros((npad, w2d.shape[1]), device=w2d.device, dtype=w2d.dtype)
    out[:cout] = w2d
    return out


def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    """OIHW -> [cout][cin/32][kh*kw][32] (the kernel's K order: channel chunk, tap, channel-in-chunk)."""
    co, ci, kh, kw = w.shape
    return w.float().permute(0, 2, 3, 1).reshape(co, kh * kw, ci // 32, 32).permute(0, 2, 1, 3).reshape(co, -1).contiguous()


def pack_conv(conv: nn.Conv2d, bn: Optional[nn.BatchNorm2d] = None) -> PConv:
    p = PConv()
    w = conv.weight.detach()
    p.cout, p.cin, p.kh, p.kw = w.shape
    p.stride, p.pad = conv.stride[0], conv.padding[0]
    p.w = _pad_rows(pack_conv_weight(w))
    bias = conv.bias.detach().float() if conv.bias is not None else None
    if bn is not None:
        scale = (bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps))
        shift = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
        if bias is not None:
            shift = shift + bias.double() * scale
        p.scale, p.shift = scale.float().contiguous(), shift.float().contiguous()
    else:
        p.scale, p.shift = None, (bias.contiguous() if bias is not None else None)
    return p


def pack_linear_w(weight: torch.Tensor, bias: Optional[torch.Tensor]) -> PConv:
    p = PConv()
    w = weight.detach().float()
    p.cout, p.cin = w.shape
    p.kh = p.kw = p.stride = 1
    p.pad = 0
    p.w = _pad_rows(w)
    p.scale, p.shift = None, (bias.detach().float().contiguous() if bias is not None else None)
    return p


def pack_linear(lin: nn.Linear) -> PConv:
    return pack_linear_w(lin.weight, lin.bias)


def _rows(t: torch.Tensor) -> Img:
    """(rows, c) matrix as a batch of 1x1 images."""
    r, c = t.shape
    return Img(t.view(r, 1, 1, c))


class State:
    """Per-module cache of packed weights + scratch."""

    def __init__(self, device):
        self.device = device
        self.packs: Dict[int, object] = {}
        self.workspace = torch.empty(_WORKSPACE_FLOATS, device=device, dtype=torch.float32)

    def get(self, key, builder):
        k = id(key) if not isinstance(key, (str, tuple)) else key
        v = self.packs.get(k)
        if v is None:
            with torch.no_grad():
                v = builder()
            self.packs[k] = v
        return v


def _state(mod: nn.Module, device) -> State:
    st = mod.__dict__.get("_egr_state")
    if st is None or st.device != device:
        st = State(device)
        mod.__dict__["_egr_state"] = st
        if "_egr_hook" not in mod.__dict__:
            mod.__dict__["_egr_hook"] = mod.register_load_state_dict_post_hook(lambda m, _k: invalidate(m))
    return st


def invalidate(mod: nn.Module):
    """Drop packed weights (call after mutating parameters in place; load_state_dict does it itself)."""
    mod.__dict__.pop("_egr_state", None)


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
    return hip.conv2d(x, p.w, p.cout, p.kh, p.kw, p.stride, p.pad, scale=p.scale, shift=p.shift, act=act,
                      workspace=st.workspace, split_k=kw.pop("split_k", 0), **kw)


def linear(st: State, x: torch.Tensor, p: PConv, act=ACT_NONE, **kw) -> torch.Tensor:
    out = conv(st, _rows(x), p, act, **kw)
    return out.t.view(x.shape[0], p.cout)


def run_stack(st: State, seq: nn.Sequential, x: Img, out: Optional[Img] = None, last_kw: Optional[dict] = None) -> Optional[Img]:
    """Execute a tree.stack(): Conv2d(+ReLU) fuse into one launch; Upsample / MaxPool2d are their own kernels.
    `out` / `last_kw` apply to the final conv."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d):
            act = ACT_NONE
            if i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU):
                act = ACT_RELU
            last = (i + (2 if act else 1)) >= len(mods)
            p = st.get(m, lambda m=m: pack_conv(m))
            kw = dict(last_kw or {}) if last else {}
            if last and out is not None:
                kw["out"] = out
            x = conv(st, x, p, act, **kw)
            i += 2 if act else 1
        elif isinstance(m, nn.Upsample):
            # Upsample -> Conv2d(1x1) -> ReLU is evaluated as Conv2d(1x1) -> Upsample(+ReLU): the bias-carrying
            # 1x1 conv commutes with bilinear interpolation, and runs on a quarter of the pixels this way.
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if (isinstance(nxt, nn.Conv2d) and nxt.kernel_size == (1, 1) and nxt.stride == (1, 1) and i + 2 < len(mods)
                    and isinstance(mods[i + 2], nn.ReLU)):
                p = st.get(nxt, lambda m=nxt: pack_conv(m))
                lo = conv(st, x, p, ACT_NONE)
                last = (i + 3) >= len(mods)
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

def _pack_stem(conv1: nn.Conv2d, bn: nn.BatchNorm2d):
    w = conv1.weight.detach().float().reshape(64, 147)
    wp = torch.zeros((64, 148), device=w.device, dtype=torch.float32)
    wp[:, :147] = w
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    shift = bn.bias.detach().double() - bn.running_mean.detach().double() * scale
    return wp.contiguous(), scale.float().contiguous(), shift.float().contiguous()


def _basic_block(st: State, blk, x: Img, out: Optional[Img] = None) -> Img:
    identity = x
    if blk.downsample is not None:
        pd = st.get(blk.downsample, lambda: pack_conv(blk.downsample[0], blk.downsample[1]))
        identity = conv(st, x, pd, ACT_NONE)
    p1 = st.get(blk.conv1, lambda: pack_conv(blk.conv1, blk.bn1))
    p2 = st.get(blk.conv2, lambda: pack_conv(blk.conv2, blk.bn2))
    y = conv(st, x, p1, ACT_RELU)
    return conv(st, y, p2, ACT_RELU, res=identity, res_mode=RES_BEFORE_ACT, out=out)


def run_backbone(st: State, enc, img: torch.Tensor, view0: int, nviews: int, feat_out: Img, s32_out: Optional[Img] = None):
    """ResNet-18 trunk + FPN for views [view0, view0+nviews) (resnet.py:43-74,121-137).
    Writes the stride-4 feature into `feat_out` (nviews*B, 64, 64, 128) and, if given, the stride-32 feature
    into `s32_out`; returns the pyramid [s4, s8, s16, s32]."""
    trunk, neck = enc.backbone, enc.neck
    wp, sc, sh = st.get(trunk.layer_s2, lambda: _pack_stem(trunk.layer_s2[0], trunk.layer_s2[1]))
    x = hip.stem(img, view0, nviews, wp, sc, sh)
    x = hip.maxpool(x, 3, 2, 1)
    pyramid = []
    stages = (trunk.layer_s4[1], trunk.layer_s8, trunk.layer_s16, trunk.layer_s32)
    for si, stage in enumerate(stages):
        for bi, blk in enumerate(stage):
            last = si == len(stages) - 1 and bi == len(stage) - 1
            x = _basic_block(st, blk, x, out=s32_out if last else None)
        pyramid.append(x)
    # FPN top-down: lateral conv writes the left half of a 256-wide buffer, the x2 upsample the right half
    n = x.n
    lat = conv(st, pyramid[3], st.get(neck.lateral_convs[3], lambda: pack_conv(neck.lateral_convs[3][0])), ACT_RELU)
    for i in (3, 2, 1):
        lo = pyramid[i - 1]
        cat = torch.empty((n, lo.h, lo.w, 2 * neck.out_channels), device=img.device, dtype=torch.float32)
        c = neck.out_channels
        conv(st, lo, st.get(neck.lateral_convs[i - 1], lambda i=i: pack_conv(neck.lateral_convs[i - 1][0])), ACT_RELU,
             out=Img(cat[..., :c]))
        hip.upsample2x(lat, out=Img(cat[..., c:]))
        fused = conv(st, Img(cat), st.get(neck.fuse_convs[i - 1], lambda i=i: pack_conv(neck.fuse_convs[i - 1][0])), ACT_RELU)
        lat = conv(st, fused, st.get(neck.fpn_convs[i - 1], lambda i=i: pack_conv(neck.fpn_convs[i - 1][0])), ACT_RELU,
                   out=feat_out if i == 1 else None)
    return pyramid


def _vb_view(t: torch.Tensor, V: int, B: int) -> torch.Tensor:
    """(V*B, H, W, C) view-major NHWC buffer -> logical (B, V, C, H, W) view (no copy)."""
    n, h, w, c = t.shape
    return t.view(V, B, h, w, c).permute(1, 0, 4, 2, 3)


# --------------------------------------------------------------------------- EgoPoseFormerHeatmap API

def heatmap_backbone_api(mod, img):
    _check_input(img, mod)
    st = _state(mod, img.device)
    B, V = img.shape[:2]
    feat = torch.empty((V * B, img.shape[3] // 4, img.shape[4] // 4, mod.encoder.neck.out_channels), device=img.device,
                       dtype=torch.float32)
    pyr = run_backbone(st, mod.encoder, img.contiguous(), 0, V, Img(feat))
    return _vb_view(feat, V, B), [_vb_view(p.t, V, B) for p in pyr]


def heatmap_forward_api(mod, img, return_feat=False):
    """EgoPoseFormerHeatmap.forward (egoposeformer_heatmap.py:29-44)."""
    _check_input(img, mod)
    st = _state(mod, img.device)
    B, V = img.shape[:2]
    H4, W4 = img.shape[3] // 4, img.shape[4] // 4
    feat = torch.empty((V * B, H4, W4, mod.encoder.neck.out_channels), device=img.device, dtype=torch.float32)
    pyr = run_backbone(st, mod.encoder, img.contiguous(), 0, V, Img(feat))
    hm = torch.empty((B, V, mod.num_heatmap, H4, W4), device=img.device, dtype=torch.float32)
    plane = mod.num_heatmap * H4 * W4
    conv(st, Img(feat), st.get(mod.conv_heatmap, lambda: pack_conv(mod.conv_heatmap)), ACT_NONE, out_nchw=hm,
         ymap=NMap(B, V * plane, plane))
    if return_feat:
        return hm, _vb_view(feat, V, B), [_vb_view(p.t, V, B) for p in pyr]
    return hm


# --------------------------------------------------------------------------- deformable-attention layer (a12-a15, a24)

class PLayer:
    __slots__ = ("offs_logits", "head_w", "head_shift", "pos_proj", "out_proj", "fuse", "ln_cross", "qkv", "mha_out",
                 "ln_spatial", "ffn0", "ffn1", "ln_ffn", "heads", "dh", "C")


def pack_layer(layer, pre_w: torch.Tensor, pre_b: torch.Tensor, pos: Optional[torch.Tensor]) -> PLayer:
    """Fold the linear chain in front of the sampling so that sampling can come first:
        value = W_v (W_pre f + b_pre [+ pos]) + b_v
      sampled & weighted:  W_v W_pre g + (W_v b_pre + b_v) sigma + W_v e
    (g, e, sigma from egr_msda_gather_f32).  W_pre/b_pre is the 1x1 conv in front of the attention
    (frame_feat_multi_view_proj for MVFEx, feat_proj for the lifting head).  Folded in fp64, stored fp32."""
    ca = layer.cross_attn
    P = PLayer()
    P.heads, P.C = ca.n_heads, ca.d_model
    P.dh = P.C // P.heads
    dev = ca.value_proj.weight.device
    Wv, bv = ca.value_proj.weight.detach().double(), ca.value_proj.bias.detach().double()
    Wp, bp = pre_w.detach().double().reshape(pre_w.shape[0], -1), pre_b.detach().double()
    Wfold = (Wv @ Wp).float()                      # (C, cf)
    cfold = (Wv @ bp + bv).float()                 # (C,)
    P.head_w = [_pad_rows(Wfold[h * P.dh:(h + 1) * P.dh].contiguous()) for h in range(P.heads)]
    P.head_shift = [cfold[h * P.dh:(h + 1) * P.dh].contiguous() for h in range(P.heads)]
    P.pos_proj = None
    if pos is not None:  # (1, V, HW, C) -> (V, HW, C) projected by W_v
        pp = pos.detach()[0].double() @ Wv.t()
        P.pos_proj = pp.float().contiguous()
    ol_w = torch.cat([ca.sampling_offsets.weight.detach(), ca.attention_weights.weight.detach()], 0)
    ol_b = torch.cat([ca.sampling_offsets.bias.detach(), ca.attention_weights.bias.detach()], 0)
    P.offs_logits = pack_linear_w(ol_w, ol_b)
    P.out_proj = pack_linear(ca.output_proj)
    P.fuse = pack_linear(layer.fuse_mlp)
    sa = layer.spatial_attn
    P.qkv = pack_linear_w(torch.cat([sa.q_proj.weight.detach(), sa.k_proj.weight.detach(), sa.v_proj.weight.detach()], 0),
                          torch.cat([sa.q_proj.bias.detach(), sa.k_proj.bias.detach(), sa.v_proj.bias.detach()], 0))
    P.mha_out = pack_linear(sa.out_proj)
    P.ffn0 = pack_linear(layer.ffn.layers[0][0])
    P.ffn1 = pack_linear(layer.ffn.layers[1])
    f32 = lambda t: t.detach().float().contiguous()
    P.ln_cross = (f32(layer.norm_cross.weight), f32(layer.norm_cross.bias))
    P.ln_spatial = (f32(layer.norm_spatial.weight), f32(layer.norm_spatial.bias))
    P.ln_ffn = (f32(layer.norm_ffn.weight), f32(layer.norm_ffn.bias))
    assert dev == P.head_w[0].device
    return P


def run_layer(st: State, P: PLayer, x: torch.Tensor, memory: torch.Tensor, anchors: torch.Tensor, valid: torch.Tensor,
              B: int, V: int, J: int, hgt: int, wid: int) -> torch.Tensor:
    """One MultiViewTransformerLayer / EgoPoseFormerTransformerLayer (heatmap_mvf_ex.py:874-935,
    egoposeformer_mvf_ex.py:546-588).  x (B*J, C); memory (V, B, hgt*wid, cf) *un-projected* features."""
    C, heads, dh = P.C, P.heads, P.dh
    ol = linear(st, x, P.offs_logits)                                   # (B*J, heads*16*3); shared by all views
    g, e, sigma, rowmask = hip.msda_gather(memory, P.pos_proj, ol, anchors, valid, B, V, J, heads, dh, hgt, wid)
    rows = B * J * V
    cf = g.shape[-1]
    a = torch.empty((rows, C), device=x.device, dtype=torch.float32)
    g2 = g.view(rows, heads * cf)
    for h in range(heads):                                              # per-head folded value projection
        hip.conv2d(_rows(g2[:, h * cf:(h + 1) * cf]), P.head_w[h], dh, 1, 1, 1, 0, shift=P.head_shift[h],
                   rowscale=sigma[h], res=_rows(e[:, h * dh:(h + 1) * dh]) if e is not None else None,
                   res_mode=RES_AFTER_ACT if e is not None else RES_NONE, out=_rows(a[:, h * dh:(h + 1) * dh]),
                   workspace=None, split_k=1)
    o = linear(st, a, P.out_proj, rowmask=rowmask)                      # masked_fill(~valid) after output_proj
    f = linear(st, o.view(B * J, V * C), P.fuse)                        # cat over views is the row layout already
    x = hip.layernorm(f, P.ln_cross[0], P.ln_cross[1], res=x)
    qkv = linear(st, x, P.qkv)
    att = hip.joint_mha(qkv, B, J, heads, dh, dh ** -0.5)
    x = hip.layernorm(linear(st, att, P.mha_out), P.ln_spatial[0], P.ln_spatial[1], res=x)
    h1 = linear(st, x, P.ffn0, ACT_GELU)
    x = hip.layernorm(linear(st, h1, P.ffn1), P.ln_ffn[0], P.ln_ffn[1], res=x)
    return x


# --------------------------------------------------------------------------- EgoPoseFormerHeatmapMVFEX (a5-a18)

class PRefiner:
    __slots__ = ("hp0", "hp2", "fc_bfb", "fc_query", "embed", "layer", "post_norm", "head0_w", "head0_b", "head3")


def _pack_refiner(r) -> PRefiner:
    P = PRefiner()
    P.hp0, P.hp2 = pack_linear(r.heatmap_proj[0]), pack_linear(r.heatmap_proj[2])
    P.fc_bfb, P.fc_query = pack_linear(r.fc_bfb), pack_linear(r.fc_query[0])
    P.embed = r.joint_query_embed.weight.detach().float().contiguous()
    mv = r.frame_feat_multi_view_proj
    P.layer = pack_layer(r.transformer_layers[0], mv.weight, mv.bias, r.frame_feat_multi_view_pos_embed)
    P.post_norm = (r.post_norm[0].weight.detach().float().contiguous(), r.post_norm[0].bias.detach().float().contiguous())
    head = r.head_layers[0].head
    P.head0_w = head[0].weight.detach().float().reshape(head[0].weight.shape[0], -1).contiguous()   # (64, 15)
    P.head0_b = head[0].bias.detach().float().contiguous()
    P.head3 = pack_conv(head[3])
    return P


def _run_refiner(st: State, r, v: int, B: int, V: int, hm_init: torch.Tensor, feat_all: torch.Tensor, s32_all: torch.Tensor,
                 anchors, valid, feat_refined_out: Img, hm_refined: torch.Tensor):
    """HeatmapMVF.forward for view v (heatmap_mvf_ex.py:652-731)."""
    P: PRefiner = st.get(r, lambda: _pack_refiner(r))
    J, C = r.num_heatmap, r.embed_dims
    hgt, wid = r.feat_shape
    hw = hgt * wid
    dev = feat_all.device
    # --- joint queries (JQA): heatmap_proj(hm) + fc_bfb(avgpool s32) + embedding -> fc_query
    hm_rows = Img(hm_init.view(B * V, J, 1, hw)[v::V])                       # (B, J, 1, hw): rows of this view
    t = conv(st, hm_rows, P.hp0, ACT_RELU)                                   # (B, J, 1, C)
    hm_embed = linear(st, t.t.view(B * J, C), P.hp2)
    bfb = linear(st, hip.avgpool(Img(s32_all[v * B:(v + 1) * B])), P.fc_bfb)
    x = linear(st, hip.jqa_sum(hm_embed, P.embed, bfb, B, J, C), P.fc_query, ACT_RELU)
    # --- own-view feature projection
    own = Img(feat_all[v * B:(v + 1) * B])
    ff = run_stack(st, r.frame_feat_proj_layers, own)                        # (B, 32, 32, 128)
    # --- transformer layer over the 4-view memory (sampled un-projected, see pack_layer)
    x = run_layer(st, P.layer, x, feat_all.view(V, B, hw, feat_all.shape[-1]), anchors, valid, B, V, J, hgt, wid)
    # --- head: LN -> (B, J, 16, 16) image with joints as channels -> 1x1 15->64, up x2, 1x1 64->128 (+ frame_feat)
    xn = hip.layernorm(x, P.post_norm[0], P.post_norm[1])
    side = int(math.isqrt(C))
    tok = hip.tokens_to_nhwc(xn, B, J, C)                                    # (B, 256, J)
    h0 = hip.linear_smallk(tok, J, 1, P.head0_w, P.head0_b, B * C, P.head0_w.shape[0], J, ACT_RELU)
    h0 = hip.upsample2x(Img(h0.view(B, side, side, -1)))
    summed = conv(st, h0, P.head3, ACT_RELU, res=ff, res_mode=RES_AFTER_ACT)  # offset_pred + frame_feat
    run_stack(st, r.frame_feat_refined_proj_layers[0], summed, out=feat_refined_out)
    # --- refined heatmap, written as (B, V, 15, 64, 64) planes of view v
    plane = J * hw
    run_stack(st, r.conv_heatmap_layers[0], feat_refined_out,
              last_kw={"out_nchw": hm_refined.view(-1)[v * plane:], "ymap": NMap(B, V * plane, 0)})


def _anchors(hm: torch.Tensor, thr: float):
    B, V, J = hm.shape[:3]
    anchors, maxvals, valid, index = hip.argmax_rows(hm, thr)
    return anchors.view(B, V, J, 2), maxvals.view(B, V, J), valid.view(B, V, J), index.view(B, V, J)


def anchors_from_heatmap_api(mod, heatmap):
    """get_anchors_2d_from_hm (heatmap_mvf_ex.py:128-143) -> (pts2d, maxvals, mask_valid[bool])."""
    a, m, v, _ = _anchors(heatmap.contiguous(), mod.heatmap_threshold)
    return a, m, v.bool()


def _mvfex(mod, img: torch.Tensor, heatmap_for_anchor=None):
    st = _state(mod, img.device)
    B, V = img.shape[:2]
    img = img.contiguous()
    H4, W4 = img.shape[3] // 4, img.shape[4] // 4
    J = mod.num_heatmap
    dev = img.device
    feat_all = torch.empty((V * B, H4, W4, 128), device=dev, dtype=torch.float32)
    front, back = mod.heatmap_estimator_stereo_front, mod.heatmap_estimator_stereo_back
    s32_all = torch.empty((V * B, H4 // 8, W4 // 8, 512), device=dev, dtype=torch.float32)
    run_backbone(st, front.encoder, img, 0, 2, Img(feat_all[:2 * B]), Img(s32_all[:2 * B]))
    run_backbone(st, back.encoder, img, 2, 2, Img(feat_all[2 * B:]), Img(s32_all[2 * B:]))
    # --- initial heatmaps per stereo pair, written straight into (B, V, J, H4, W4)
    hm_init = torch.empty((B, V, J, H4, W4), device=dev, dtype=torch.float32)
    plane = J * H4 * W4
    for seq, v0 in ((mod.conv_heatmap_layers_stereo_front, 0), (mod.conv_heatmap_layers_stereo_back, 2)):
        run_stack(st, seq, Img(feat_all[v0 * B:(v0 + 2) * B]),
                  last_kw={"out_nchw": hm_init.view(-1)[v0 * plane:], "ymap": NMap(B, V * plane, plane)})
    src = heatmap_for_anchor.contiguous() if isinstance(heatmap_for_anchor, torch.Tensor) else hm_init
    anchors, maxvals, valid, index = _anchors(src, mod.heatmap_threshold)
    # --- four refiners (own weights each), every one attending to all four views
    feat_ref = torch.empty_like(feat_all)
    hm_ref = torch.empty_like(hm_init)
    for v, r in enumerate(mod.refiners()):
        _run_refiner(st, r, v, B, V, hm_init, feat_all, s32_all, anchors, valid, Img(feat_ref[v * B:(v + 1) * B]), hm_ref)
    aux = {"anchors_2d": anchors, "maxvals": maxvals, "anchors_valid": valid, "argmax_idx": index}
    return hm_init, hm_ref, feat_all, feat_ref, aux


def heatmap_mvfex_forward_api(mod, img, heatmap_for_anchor=None):
    """EgoPoseFormerHeatmapMVFEX.forward -> ([hm_init, hm_refined], [feat_init, feat_refined])."""
    _check_input(img, mod)
    B, V = img.shape[:2]
    hm_init, hm_ref, feat_all, feat_ref, aux = _mvfex(mod, img, heatmap_for_anchor)
    mod.__dict__["_egr_last_aux"] = aux
    return [hm_init, hm_ref], [_vb_view(feat_all, V, B), _vb_view(feat_ref, V, B)]


# --------------------------------------------------------------------------- EgoPoseFormerPose3D (a20-a24)

class PPose:
    __slots__ = ("mlp0", "mlp1", "mlp2", "qg0_w", "qg0_b", "qg2", "qg4", "layers", "post", "reg0", "reg2", "cams")


def _pack_pose3d(p3) -> PPose:
    P = PPose()
    V = p3.num_views
    w0 = p3.mlp_pred[0][0].weight.detach()
    n_out = w0.shape[0]
    # reference flattens "(b v) c h w -> b (v c h w)" (egoposeformer_mvf_ex.py:317); ours is (v, h, w, c)
    w0p = w0.view(n_out, V, 128, 8, 8).permute(0, 1, 3, 4, 2).reshape(n_out, -1)
    P.mlp0 = pack_linear_w(w0p, p3.mlp_pred[0][0].bias)
    P.mlp1 = pack_linear(p3.mlp_pred[1][0])
    P.mlp2 = pack_linear(p3.mlp_pred[2])
    qg = p3.query_gen_mlp
    P.qg0_w, P.qg0_b = qg[0].weight.detach().float().contiguous(), qg[0].bias.detach().float().contiguous()
    P.qg2, P.qg4 = pack_linear(qg[2]), pack_linear(qg[4])
    fp = p3.feat_proj
    P.layers = [pack_layer(l, fp.weight, fp.bias, None) for l in p3.layers]
    f32 = lambda t: t.detach().float().contiguous()
    P.post = [(f32(n.weight), f32(n.bias)) for n in p3.post_norm]
    P.reg0 = [pack_linear(r[0]) for r in p3.reg_mlp]
    P.reg2 = [pack_linear(r[2]) for r in p3.reg_mlp]
    rec = np.stack([c.packed() for c in p3.cameras()])
    P.cams = torch.from_numpy(rec).to(w0.device)
    return P


def _pose3d(p3, st: State, feat_init: torch.Tensor, feat_final: torch.Tensor, B: int, V: int, ctm):
    """EgoPoseFormerPose3D.forward (egoposeformer_mvf_ex.py:422-452).  feat_*: (V*B, 64, 64, 128) view-major NHWC."""
    P: PPose = st.get(p3, lambda: _pack_pose3d(p3))
    dev = feat_init.device
    J = p3.num_joints
    hgt, wid = p3.feat_shape
    src = feat_init if p3.use_pred_heatmap_init else feat_final
    # --- proposal: conv stack on the refined features, flattened per frame, 3-layer MLP (_forward_mlp_conv)
    flat = torch.empty((B, V * 8 * 8 * 128), device=dev, dtype=torch.float32)
    run_stack(st, p3.conv_frame_feat, Img(feat_final),
              out=Img(flat.view(B * V, 8, 8, 128)), last_kw={"ymap": NMap(B, V * 8192, 8192)})  # (v,b) -> (b,v)
    h = linear(st, flat, P.mlp0, ACT_GELU)
    h = linear(st, h, P.mlp1, ACT_GELU)
    mlp_pred = linear(st, h, P.mlp2).view(B, J, 3)
    anchors_3d = mlp_pred.clone()                                            # init_anchors_3d = mlp_pred.clone().detach()
    ctm32 = None
    if p3.camera_model.startswith("ego4view_rw"):
        if ctm is None:
            raise RuntimeError("egorear_amd: camera_model ego4view_rw needs coord_trans_mat (B,4,4,4)")
        ctm32 = ctm.to(device=dev, dtype=torch.float32).contiguous()         # any float dtype accepted
    anchors_2d, valid, q4 = hip.fisheye_project(anchors_3d, ctm32, P.cams)   # syn: anchors_3d mutated in place (F7)
    # --- decoder
    C = p3.embed_dims
    x = hip.linear_smallk(q4, 4, 1, P.qg0_w, P.qg0_b, B * J, C, 4, ACT_RELU)
    x = linear(st, x, P.qg2, ACT_RELU)
    x = linear(st, x, P.qg4)
    memory = src.view(V, B, hgt * wid, src.shape[-1])
    preds = [mlp_pred]
    a3 = anchors_3d.view(B * J, 3)
    for i, L in enumerate(P.layers):
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


def pose3d_forward_api(p3, feat_init, feat_final, ctm=None):
    _check_input(feat_init, p3)
    st = _state(p3, feat_init.device)
    B, V = feat_init.shape[:2]
    preds, aux = _pose3d(p3, st, _to_view_major(feat_init.float()), _to_view_major(feat_final.float()), B, V, ctm)
    p3.__dict__["_egr_last_aux"] = aux
    return preds


def mvfex_forward_api(mod, img, ctm=None):
    """EgoPoseFormerMVFEX.forward (egoposeformer_mvf_ex.py:50-59) -> (list_pred_pose3d, list_pred_heatmap)."""
    _check_input(img, mod)
    B, V = img.shape[:2]
    he = mod.heatmap_estimator
    hm_init, hm_ref, feat_all, feat_ref, aux_h = _mvfex(he, img)
    st = _state(mod.pose3d_estimator, img.device)
    preds, aux_p = _pose3d(mod.pose3d_estimator, st, feat_all, feat_ref, B, V, ctm)
    mod.__dict__["_egr_last_aux"] = {"heatmap": aux_h, "pose3d": aux_p}
    return preds, [hm_init, hm_ref]
