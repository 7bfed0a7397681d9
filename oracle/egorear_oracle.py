"""CPU oracle for the EgoRear multi-view inference hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain PyTorch-CPU functional form over a flat `state_dict`,
the algorithm of the reference's hot path.  It is the checker the
HIP path is compared against; it is never the thing shipped or measured.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
it.  The product package `egorear_amd` does not import it and has no CPU fallback.

Pinning: the reference holds no tests or golden vectors, so the
oracle is pinned by outputs of the reference itself, run in the build container
from /root/reference under the shims in oracle/ref_shims.py by
oracle/make_golden.py; the resulting vectors are committed under tests/golden/
and tests/test_oracle_golden.py checks this file against them.

Third-party arithmetic that is absent from /root/reference and restated here from
its published algorithm:
  * mmcv==2.2.0 `MultiScaleDeformableAttnFunction` forward (pin README.md:134;
    call site models/utils/deform_attn.py:155-162)  -> `msda_core`
  * torchvision==0.19.0 `resnet18` BasicBlock trunk (pin README.md:132;
    call site models/backbones/resnet.py:33-39)        -> `resnet_trunk`

Every function cites the reference file:line it follows (paths relative to
/root/reference/pose_estimation/).
"""
from __future__ import annotations

import json
import math
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
CAMERAS = ("camera_front_left", "camera_front_right", "camera_back_left", "camera_back_right")
REFINERS = ("front_left", "front_right", "back_left", "back_right")


# --------------------------------------------------------------------------- small helpers

def _conv(sd: SD, p: str, x, stride=1, padding=0):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def _lin(sd: SD, p: str, x):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


# BatchNorm mode.  eval (default): running statistics.  train (`with bn_train() as updates:`): batch statistics, and the
# buffers nn.BatchNorm2d would have written (momentum 0.1, unbiased running_var, num_batches_tracked + 1) are collected
# in `updates` instead of mutating `sd` (the wrappers call network.train(): pose_3d_mvf_ex.py:115).
_BN_TRAIN: Optional[dict] = None


class bn_train:
    def __enter__(self):
        global _BN_TRAIN
        _BN_TRAIN = {}
        return _BN_TRAIN

    def __exit__(self, *exc):
        global _BN_TRAIN
        _BN_TRAIN = None
        return False


def _bn(sd: SD, p: str, x):
    # BatchNorm2d, eps 1e-5
    if _BN_TRAIN is None:
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.1, 1e-5)
    rm, rv = sd[p + ".running_mean"].detach().clone(), sd[p + ".running_var"].detach().clone()
    y = F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, 1e-5)
    _BN_TRAIN[p + ".running_mean"], _BN_TRAIN[p + ".running_var"] = rm, rv
    _BN_TRAIN[p + ".num_batches_tracked"] = sd[p + ".num_batches_tracked"] + 1
    return y


def _ln(sd: SD, p: str, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _up2(x):
    # nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) everywhere (App. B-10)
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)


# --------------------------------------------------------------------------- backbone (a1-a3)

def _basic_block(sd: SD, p: str, x, stride: int):
    """torchvision BasicBlock.forward: conv3x3-bn-relu-conv3x3-bn, (+downsample), add, relu."""
    out = F.relu(_bn(sd, p + ".bn1", _conv(sd, p + ".conv1", x, stride, 1)))
    out = _bn(sd, p + ".bn2", _conv(sd, p + ".conv2", out, 1, 1))
    if (p + ".downsample.0.weight") in sd:
        x = _bn(sd, p + ".downsample.1", _conv(sd, p + ".downsample.0", x, stride, 0))
    return F.relu(out + x)


def resnet_trunk(sd: SD, p: str, img: torch.Tensor) -> List[torch.Tensor]:
    """models/backbones/resnet.py:43-74 (ResNetTorchvision.forward), out_stride 4."""
    if img.dim() == 4:  # resnet.py:44-46 grayscale branch
        B, V, H, W = img.shape
        x = img.reshape(B * V, 1, H, W).repeat(1, 3, 1, 1)
    else:
        B, V, C, H, W = img.shape
        x = img.reshape(B * V, C, H, W)
    s2 = F.relu(_bn(sd, p + ".layer_s2.1", _conv(sd, p + ".layer_s2.0", x, 2, 3)))
    x = F.max_pool2d(s2, 3, 2, 1)
    x = _basic_block(sd, p + ".layer_s4.1.0", x, 1)
    s4 = _basic_block(sd, p + ".layer_s4.1.1", x, 1)
    x = _basic_block(sd, p + ".layer_s8.0", s4, 2)
    s8 = _basic_block(sd, p + ".layer_s8.1", x, 1)
    x = _basic_block(sd, p + ".layer_s16.0", s8, 2)
    s16 = _basic_block(sd, p + ".layer_s16.1", x, 1)
    x = _basic_block(sd, p + ".layer_s32.0", s16, 2)
    s32 = _basic_block(sd, p + ".layer_s32.1", x, 1)
    return [t.reshape(B, V, *t.shape[1:]) for t in (s4, s8, s16, s32)]


def fpn_neck(sd: SD, p: str, feats: Sequence[torch.Tensor]) -> torch.Tensor:
    """models/backbones/resnet.py:121-137 (EfficientFPN.forward)."""
    B, V = feats[0].shape[:2]
    xs = [f.flatten(0, 1) for f in feats]
    lat = [F.relu(_conv(sd, f"{p}.lateral_convs.{i}.0", x)) for i, x in enumerate(xs)]
    for i in range(len(lat) - 1, 0, -1):
        cat = torch.cat((lat[i - 1], _up2(lat[i])), dim=1)
        fused = F.relu(_conv(sd, f"{p}.fuse_convs.{i - 1}.0", cat))
        lat[i - 1] = F.relu(_conv(sd, f"{p}.fpn_convs.{i - 1}.0", fused, 1, 1))
    return lat[0].reshape(B, V, *lat[0].shape[1:])


def heatmap_backbone(sd: SD, p: str, img):
    """models/estimator/egoposeformer_heatmap.py:25-27 + resnet.py:149-152."""
    pyramid = resnet_trunk(sd, p + ".encoder.backbone", img)
    return fpn_neck(sd, p + ".encoder.neck", pyramid), pyramid


def heatmap_forward(sd: SD, p: str, img, return_feat: bool = False):
    """models/estimator/egoposeformer_heatmap.py:29-44 (EgoPoseFormerHeatmap.forward)."""
    B, V = img.shape[:2]
    feats, pyramid = heatmap_backbone(sd, p, img)
    hm = _conv(sd, p + ".conv_heatmap", feats.view(B * V, *feats.shape[2:]))
    hm = hm.view(B, V, *hm.shape[1:])
    return (hm, feats, pyramid) if return_feat else hm


# --------------------------------------------------------------------------- argmax (a7)

def get_max_preds(heatmaps: torch.Tensor, threshold: float = 0.5, normalize: bool = True):
    """utils/loss.py:122-142.  Returns preds (N,J,2) f32, maxvals (N,J), valid (N,J) and,
    additionally for the parity tests, the raw int64 flat index (N,J)."""
    N, J, H, W = heatmaps.shape
    maxvals, idx = torch.max(heatmaps.reshape(N, J, -1), dim=2, keepdim=True)
    preds = torch.tile(idx, (1, 1, 2)).float()
    preds[:, :, 0] = preds[:, :, 0] % W
    preds[:, :, 1] = preds[:, :, 1] // W
    if normalize:
        preds[:, :, 0] = preds[:, :, 0] / W
        preds[:, :, 1] = preds[:, :, 1] / H
    valid = maxvals >= threshold
    return preds, maxvals.reshape(N, J), valid.reshape(N, J), idx.reshape(N, J)


# --------------------------------------------------------------------------- deformable attention (a13, N1)

def msda_core(value: torch.Tensor, H: int, W: int, loc: torch.Tensor, attn: torch.Tensor) -> torch.Tensor:
    """mmcv 2.2.0 ms_deform_attn forward, single level (published algorithm:
    ms_deform_attn_im2col_bilinear).  value (N, H*W, nh, D); loc (N, Lq, nh, P, 2) in
    normalised (x, y); attn (N, Lq, nh, P).  -> (N, Lq, nh*D).
    pixel = loc*size - 0.5; a point contributes iff -1 < h,w and h < H, w < W; each of the
    four corners contributes iff it lies inside the map (zero padding)."""
    N, L, nh, D = value.shape
    Lq, P = loc.shape[1], loc.shape[3]
    w_im = loc[..., 0] * W - 0.5
    h_im = loc[..., 1] * H - 0.5
    inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
    h_low = torch.floor(h_im)
    w_low = torch.floor(w_im)
    lh, lw = h_im - h_low, w_im - w_low
    hh, hw = 1 - lh, 1 - lw
    h_low, w_low = h_low.long(), w_low.long()
    h_high, w_high = h_low + 1, w_low + 1
    vt = value.permute(0, 2, 1, 3)  # (N, nh, L, D)

    def corner(hi, wi, ok):
        ok = ok & inside
        flat = (hi.clamp(0, H - 1) * W + wi.clamp(0, W - 1)).permute(0, 2, 1, 3).reshape(N, nh, Lq * P)
        g = torch.gather(vt, 2, flat[..., None].expand(N, nh, Lq * P, D)).reshape(N, nh, Lq, P, D)
        return g * ok.permute(0, 2, 1, 3)[..., None].to(g.dtype)

    v1 = corner(h_low, w_low, (h_low >= 0) & (w_low >= 0))
    v2 = corner(h_low, w_high, (h_low >= 0) & (w_high <= W - 1))
    v3 = corner(h_high, w_low, (h_high <= H - 1) & (w_low >= 0))
    v4 = corner(h_high, w_high, (h_high <= H - 1) & (w_high <= W - 1))

    def pw(t):
        return t.permute(0, 2, 1, 3)[..., None]

    val = pw(hh * hw) * v1 + pw(hh * lw) * v2 + pw(lh * hw) * v3 + pw(lh * lw) * v4
    out = (val * pw(attn)).sum(dim=3)  # (N, nh, Lq, D)
    return out.permute(0, 2, 1, 3).reshape(N, Lq, nh * D)


def msda_levels(value: torch.Tensor, spatial_shapes, level_start_index, loc: torch.Tensor, attn: torch.Tensor) -> torch.Tensor:
    """mmcv 2.2.0 MultiScaleDeformableAttnFunction.apply in its own operand layout (call site
    models/utils/deform_attn.py:155-162): value (N, Lin, nh, D); spatial_shapes (L, 2) = (H, W); level_start_index (L,);
    loc (N, Lq, nh, L, P, 2); attn (N, Lq, nh, L, P) -> (N, Lq, nh*D).  The sum over levels of msda_core on each level's
    token range; differentiable by autograd in value, loc and attn (floor and the inside tests carry no gradient, as in
    mmcv's col2im kernels)."""
    out = None
    for lvl, ((H, W), st) in enumerate(zip(torch.as_tensor(spatial_shapes).tolist(), torch.as_tensor(level_start_index).tolist())):
        o = msda_core(value[:, st : st + H * W], int(H), int(W), loc[:, :, :, lvl], attn[:, :, :, lvl])
        out = o if out is None else out + o
    return out


def ms_deform_attn(sd: SD, p: str, query, ref_pts, memory, H: int, W: int, n_heads: int = 4, n_points: int = 16):
    """models/utils/deform_attn.py:90-168 (MSDeformAttn.forward, one level, 2-d reference points)."""
    N, Lq, C = query.shape
    Lin = memory.shape[1]
    assert H * W == Lin
    value = _lin(sd, p + ".value_proj", memory).view(N, Lin, n_heads, C // n_heads)
    off = _lin(sd, p + ".sampling_offsets", query).view(N, Lq, n_heads, 1, n_points, 2)
    aw = _lin(sd, p + ".attention_weights", query).view(N, Lq, n_heads, n_points)
    aw = F.softmax(aw, -1).view(N, Lq, n_heads, 1, n_points)
    normalizer = torch.tensor([[W, H]], dtype=torch.long)  # (W, H) order: deform_attn.py:131-133
    loc = ref_pts[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
    # (deform_attn.py:155 casts to float for the mmcv op; a float64 run of this oracle - the referee of oracle/census.py - stays float64)
    out = msda_core(value if value.dtype == torch.float64 else value.float(), H, W, loc[:, :, :, 0], aw[:, :, :, 0])
    return _lin(sd, p + ".output_proj", out)


def joint_mha(sd: SD, p: str, x, num_heads: int = 4):
    """SpatialMHA / EgoformerSpatialMHA forward (heatmap_mvf_ex.py:799-817,
    egoposeformer_mvf_ex.py:481-498) over models/utils/transformer.py:36-81."""
    B, J, C = x.shape
    d = C // num_heads
    q = _lin(sd, p + ".q_proj", x).reshape(B, J, num_heads, d).permute(0, 2, 1, 3)
    k = _lin(sd, p + ".k_proj", x).reshape(B, J, num_heads, d).permute(0, 2, 1, 3)
    v = _lin(sd, p + ".v_proj", x).reshape(B, J, num_heads, d).permute(0, 2, 1, 3)
    attn = ((q @ k.transpose(-2, -1)) * (d ** -0.5)).softmax(dim=-1)
    out = (attn @ v).permute(0, 2, 1, 3).reshape(B, J, C)
    return _lin(sd, p + ".out_proj", out)


def ffn(sd: SD, p: str, x):
    """models/utils/transformer.py:8-33: Linear, exact-erf GELU, Linear (dropout p=0)."""
    return _lin(sd, p + ".layers.1", F.gelu(_lin(sd, p + ".layers.0.0", x)))


def joint_transformer_layer(sd: SD, p: str, x, memory, anchors_2d, anchors_valid, H: int, W: int, cap: Optional[dict] = None):
    """MultiViewTransformerLayer.forward (heatmap_mvf_ex.py:874-935) ==
    EgoPoseFormerTransformerLayer.forward (egoposeformer_mvf_ex.py:546-588).
    memory (B,V,HW,C); anchors_2d (B,V,J,2); anchors_valid (B,V,J) bool."""
    B, V = memory.shape[:2]
    per_view = []
    for v in range(V):
        ref = anchors_2d[:, v].reshape(B, -1, 1, 2)
        a = ms_deform_attn(sd, p + ".cross_attn", x, ref, memory[:, v].reshape(B, -1, memory.shape[-1]), H, W)
        if cap is not None:
            cap.setdefault("msda", []).append(a)          # the deformable attention's own output (before the validity mask): a13
        a = a.masked_fill(~anchors_valid[:, v][..., None].expand_as(a), 0.0)  # after output_proj (App. B-2)
        per_view.append(a)
    x = _ln(sd, p + ".norm_cross", x + _lin(sd, p + ".fuse_mlp", torch.cat(per_view, dim=-1)))
    x = _ln(sd, p + ".norm_spatial", x + joint_mha(sd, p + ".spatial_attn", x))
    x = _ln(sd, p + ".norm_ffn", x + ffn(sd, p + ".ffn", x))
    return x


# --------------------------------------------------------------------------- heatmap heads / refiner (a6, a9-a18)

def init_heatmap_head(sd: SD, p: str, x):
    """conv_heatmap_layers_stereo_{front,back} (heatmap_mvf_ex.py:101-126)."""
    x = F.relu(_conv(sd, p + ".0", x))
    x = F.relu(_conv(sd, p + ".2", x, 2, 1))
    x = F.relu(_conv(sd, p + ".4", x))
    x = _up2(x)
    x = F.relu(_conv(sd, p + ".7", x))
    return _conv(sd, p + ".9", x)


def heatmap_mvf(sd: SD, p: str, heatmap, frame_feat, feat_mv, anchors_2d, anchors_valid, s32_own, detach_heatmap_feat: bool = True,
                cap: Optional[dict] = None):
    """HeatmapMVF.forward, JQA branch, one transformer layer, non-1x1 heatmap head
    (heatmap_mvf_ex.py:652-731).  Returns (heatmap_refined, frame_feat_refined).
    Gradient stops follow the source: `frame_feat.detach()` (:715, unconditional) and, with detach_heatmap_feat
    (all shipped configs), `frame_feat_refined.detach()` in front of the heat-map convs (:717-721)."""
    B, V, C, H, W = feat_mv.shape
    hm_embed = _lin(sd, p + ".heatmap_proj.2", F.relu(_lin(sd, p + ".heatmap_proj.0", heatmap.reshape(B, heatmap.shape[1], H * W))))
    bfb = F.adaptive_avg_pool2d(s32_own, (1, 1)).view(B, -1)
    bfb = _lin(sd, p + ".fc_bfb", bfb).unsqueeze(1)
    embed = sd[p + ".joint_query_embed.weight"].unsqueeze(0).repeat(B, 1, 1)
    x = F.relu(_lin(sd, p + ".fc_query.0", embed + bfb + hm_embed))
    # memory (:689-693)
    mem = _conv(sd, p + ".frame_feat_multi_view_proj", feat_mv.reshape(B * V, C, H, W))
    mem = mem.reshape(B, V, -1, H * W).permute(0, 1, 3, 2)
    mem = mem + sd[p + ".frame_feat_multi_view_pos_embed"]
    # own-view feature projection (:525-532, :695)
    ff = F.relu(_conv(sd, p + ".frame_feat_proj_layers.0", frame_feat))
    ff = F.relu(_conv(sd, p + ".frame_feat_proj_layers.2", ff, 2, 1))
    ff = F.relu(_conv(sd, p + ".frame_feat_proj_layers.4", ff))
    if cap is not None:
        cap["query"] = x                                   # JQA query behind fc_query: a9
    x = joint_transformer_layer(sd, p + ".transformer_layers.0", x, mem, anchors_2d, anchors_valid, H, W, cap)
    _x = _ln(sd, p + ".post_norm.0", x)
    if cap is not None:
        cap["post_norm"] = _x                              # the transformer layer + post_norm: a12
    side = int(math.sqrt(_x.shape[-1]))
    _x = _x.reshape(B, -1, side, side)  # joints as channels (App. B-3)
    hp = p + ".head_layers.0.head"
    off = F.relu(_conv(sd, hp + ".3", _up2(F.relu(_conv(sd, hp + ".0", _x)))))
    rp = p + ".frame_feat_refined_proj_layers.0"
    if cap is not None:
        cap["head_sum"] = off + ff                         # TransformerHeadLayer output + frame_feat_proj_layers output: a16 + a11
    refined = F.relu(_conv(sd, rp + ".3", _up2(F.relu(_conv(sd, rp + ".0", off + ff.detach())))))
    cp = p + ".conv_heatmap_layers.0"
    h = F.relu(_conv(sd, cp + ".0", refined.detach() if detach_heatmap_feat else refined, 2, 1))
    h = F.relu(_conv(sd, cp + ".2", h))
    h = F.relu(_conv(sd, cp + ".5", _up2(h)))
    h = _conv(sd, cp + ".7", h)
    return h, refined


def heatmap_mvfex_forward(sd: SD, p: str, img, heatmap_threshold: float = 0.5, heatmap_for_anchor=None,
                          full_training: bool = True, use_pred_heatmap_init: bool = True, no_detach_feat_init: bool = False,
                          detach_heatmap_feat: bool = True, capture: bool = False):
    """EgoPoseFormerHeatmapMVFEX.forward, num_views==4, use_1by1_conv False
    (heatmap_mvf_ex.py:236-352).  In eval/no_grad the detach/clone branches are value-neutral; under autograd they are
    the gradient stops of :262-288 (flag defaults = the shipped pose3d configs).
    Returns ([hm_init, hm_refined], [feat_init, feat_refined], aux) where aux carries the
    argmax intermediates the parity tests pin."""
    pre = (p + ".") if p else ""
    B = img.shape[0]
    with torch.enable_grad() if (full_training and torch.is_grad_enabled()) else torch.no_grad():   # :262-266
        f_front, pyr_front = heatmap_backbone(sd, pre + "heatmap_estimator_stereo_front", img[:, 0:2])
        f_back, pyr_back = heatmap_backbone(sd, pre + "heatmap_estimator_stereo_back", img[:, 2:])
    feat_init = torch.cat((f_front, f_back), dim=1)  # (B,4,128,64,64)
    s32 = torch.cat((pyr_front[-1], pyr_back[-1]), dim=1)  # (B,4,512,8,8)
    _, V, C, H, W = feat_init.shape
    head_in = feat_init.detach() if use_pred_heatmap_init else feat_init  # :273 / :284
    hm_front = init_heatmap_head(sd, pre + "conv_heatmap_layers_stereo_front", head_in[:, 0:2].reshape(B * 2, C, H, W))
    hm_back = init_heatmap_head(sd, pre + "conv_heatmap_layers_stereo_back", head_in[:, 2:].reshape(B * 2, C, H, W))
    hm_init_out = torch.cat((hm_front.view(B, 2, -1, H, W), hm_back.view(B, 2, -1, H, W)), dim=1)
    if use_pred_heatmap_init:                                             # :275-282
        hm_init = hm_init_out.detach().clone()
        feat, s32 = (feat_init, s32) if no_detach_feat_init else (feat_init.detach().clone(), s32.detach().clone())
    else:
        hm_init, feat = hm_init_out, feat_init
    src = heatmap_for_anchor if isinstance(heatmap_for_anchor, torch.Tensor) else hm_init_out
    J = src.shape[2]
    with torch.no_grad():                                                 # :128 (decorated @torch.no_grad()) + :297
        pts, maxvals, valid, idx = get_max_preds(src.detach().reshape(B * V, J, H, W), heatmap_threshold, True)
    anchors_2d, anchors_valid = pts.view(B, V, J, 2), valid.view(B, V, J)
    hms, feats = [], []
    mid = {}
    for v, name in enumerate(REFINERS):
        cap = {} if capture else None
        h, f = heatmap_mvf(sd, pre + "heatmap_refiner_" + name, hm_init[:, v], feat[:, v], feat, anchors_2d, anchors_valid, s32[:, v],
                           detach_heatmap_feat, cap)
        hms.append(h)
        feats.append(f)
        if capture:
            mid[name] = cap
    aux = {"argmax_idx": idx.view(B, V, J), "maxvals": maxvals.view(B, V, J), "anchors_valid": anchors_valid, "anchors_2d": anchors_2d, "s32": s32}
    if capture:   # per refiner: the intermediates tests/golden/mvfex_mid_s*.npz pin (queries, MSDA outputs, head offsets)
        aux["mid"] = mid
    return [hm_init_out, torch.stack(hms, dim=1)], [feat_init, torch.stack(feats, dim=1)], aux


# --------------------------------------------------------------------------- fisheye camera (a22)

_SYN_OFFSETS = {  # utils/camera_models.py:29-40
    "camera_front_left": (6.0, 0.0, 0.0),
    "camera_front_right": (-6.0, 0.0, 0.0),
    "camera_back_left": (-6.0, 37.0, 0.0),
    "camera_back_right": (6.0, 37.0, 0.0),
}


class FishEye:
    """utils/camera_models.py:14-104 on CPU (the reference hard-codes device="cuda", F8)."""

    def __init__(self, camera_model: str, calib_dir: str, name: str):
        with open(os.path.join(calib_dir, name + ".json")) as f:
            d = json.load(f)
        self.camera_model, self.name = camera_model, name
        self.image_size = torch.tensor(d["size"])  # int64
        self.image_center = torch.tensor(d["image_center"])  # fp32
        self.poly = torch.tensor(d["polynomialW2C"])  # fp32 rounding at construction (App. B-6)
        self.offset = torch.tensor(_SYN_OFFSETS[name])

    def camera_relative(self, pts3d, coord_trans_mat=None):
        """:53-68.  syn mode mutates its argument in place (F7); rw mode allocates."""
        if self.camera_model.startswith("ego4view_rw"):
            return apply_batch_transformation_matrix(pts3d * 0.01, coord_trans_mat) * 100.0
        if self.name in ("camera_back_left", "camera_back_right"):
            pts3d[..., 0:2] *= -1
        pts3d += self.offset
        return pts3d

    def world2camera(self, pts3d_original, coord_trans_mat=None):
        """:70-104."""
        pts3d = self.camera_relative(pts3d_original, coord_trans_mat)
        pts3d = pts3d[:, None].repeat(1, 1, 1, 1)
        x, y, z = pts3d[..., 0], pts3d[..., 1], pts3d[..., 2]
        norm = torch.sqrt(x * x + y * y)
        theta = torch.atan(-z / norm)
        rho = sum(a * theta ** i for i, a in enumerate(self.poly))  # left-to-right power sum (App. B-6)
        u = x / norm * rho + self.image_center[0]
        v = y / norm * rho + self.image_center[1]
        u = u / self.image_size[1]
        v = v / self.image_size[0]
        pt = torch.stack((u, v), dim=-1)
        in_fov = (pt[..., 0] > 0) & (pt[..., 1] > 0) & (pt[..., 0] < 1) & (pt[..., 1] < 1)
        return pt.clamp(min=0.0, max=1.0), in_fov


def apply_batch_transformation_matrix(pts3d, mats):
    """utils/camera_models.py:187-215; accepts any float dtype for `mats` and casts to the
    points' dtype (the reference raises on f64 x f32, F9)."""
    B, J = pts3d.shape[:2]
    ones = torch.ones((B, J, 1), dtype=pts3d.dtype)
    hom = torch.cat([pts3d, ones], dim=2)
    m = mats.to(pts3d.dtype).unsqueeze(1).expand(-1, J, -1, -1)
    return torch.matmul(m, hom.unsqueeze(3)).squeeze(3)[:, :, :3]


def reproject_3d_to_2d(cams: Sequence[FishEye], anchors_3d, coord_trans_mat=None):
    """egoposeformer_mvf_ex.py:340-382 (4-view branches).  NOTE: in syn mode this mutates
    `anchors_3d` through the four chained in-place camera offsets (F7)."""
    pts, valid = [], []
    for i, cam in enumerate(cams):
        m = coord_trans_mat[:, i] if coord_trans_mat is not None and cam.camera_model.startswith("ego4view_rw") else None
        p2, ok = cam.world2camera(anchors_3d, m)
        pts.append(p2)
        valid.append(ok)
    return torch.cat(pts, dim=1), torch.cat(valid, dim=1)


# --------------------------------------------------------------------------- 3-D lifting head (a20-a24)

def pose3d_forward(sd: SD, p: str, cams, feat_init, feat_final, coord_trans_mat=None, num_layers: int = 3, use_pred_heatmap_init: bool = True):
    """EgoPoseFormerPose3D.forward (egoposeformer_mvf_ex.py:422-452) with
    _forward_mlp_conv (:309-322) and _forward_transformer (:384-420)."""
    src = feat_init if use_pred_heatmap_init else feat_final
    B, V, C, H, W = src.shape
    ff = _conv(sd, p + ".feat_proj", src.reshape(B * V, C, H, W)).reshape(B, V, -1, H, W)
    # proposal
    cp = p + ".conv_frame_feat"
    x = F.relu(_conv(sd, cp + ".0", feat_final.reshape(B * V, C, H, W)))
    x = F.relu(_conv(sd, cp + ".2", x, 2, 1))
    x = F.max_pool2d(x, 2)
    x = F.relu(_conv(sd, cp + ".5", x))
    x = F.relu(_conv(sd, cp + ".7", x, 2, 1))
    x = x.reshape(B, V, *x.shape[1:]).reshape(B, -1)  # "(b v) c h w -> b (v c h w)" (:317)
    x = F.gelu(_lin(sd, p + ".mlp_pred.0.0", x))
    x = F.gelu(_lin(sd, p + ".mlp_pred.1.0", x))
    mlp_pred = _lin(sd, p + ".mlp_pred.2", x).reshape(B, -1, 3)
    J = mlp_pred.shape[1]
    anchors_3d = mlp_pred.clone().detach()  # (:441)
    # decoder
    mem = ff.permute(0, 1, 3, 4, 2).reshape(B, V, H * W, -1)
    with torch.no_grad():                                                # _reproject_3d_to_2d is @torch.no_grad() (:339)
        anchors_2d, anchors_valid = reproject_3d_to_2d(cams, anchors_3d, coord_trans_mat)  # mutates anchors_3d in syn mode
    anchors_2d = anchors_2d.to(mem.dtype)
    joint_inds = (torch.arange(1, J + 1).to(mem.dtype).reshape(1, J, 1).repeat(B, 1, 1)) / float(J)
    q = torch.cat((joint_inds, anchors_3d), dim=-1)
    q = F.relu(_lin(sd, p + ".query_gen_mlp.0", q))
    q = F.relu(_lin(sd, p + ".query_gen_mlp.2", q))
    x = _lin(sd, p + ".query_gen_mlp.4", q)
    preds = [mlp_pred]
    for i in range(num_layers):
        x = joint_transformer_layer(sd, f"{p}.layers.{i}", x, mem, anchors_2d, anchors_valid, H, W)
        _x = _ln(sd, f"{p}.post_norm.{i}", x)
        off = _lin(sd, f"{p}.reg_mlp.{i}.2", F.gelu(_lin(sd, f"{p}.reg_mlp.{i}.0", _x)))
        preds.append(off + anchors_3d)
    aux = {"anchors_2d": anchors_2d, "anchors_valid": anchors_valid, "anchors_3d_after": anchors_3d}
    return preds, aux


def make_cameras(camera_model: str, calib_dir: str):
    return [FishEye(camera_model, calib_dir, n) for n in CAMERAS]


def mvfex_forward(sd: SD, cams, img, coord_trans_mat=None, heatmap_threshold: float = 0.5, num_layers: int = 3):
    """EgoPoseFormerMVFEX.forward (egoposeformer_mvf_ex.py:50-59), shipped pose3d configs
    (use_pred_heatmap_init True -> decoder on init feats, proposal on refined feats)."""
    hms, feats, aux_h = heatmap_mvfex_forward(sd, "heatmap_estimator", img, heatmap_threshold)
    preds, aux_p = pose3d_forward(sd, "pose3d_estimator", cams, feats[0], feats[-1], coord_trans_mat, num_layers, True)
    return preds, hms, {"heatmap": aux_h, "pose3d": aux_p, "feats": feats}


def compute_mpjpe_batch(pred, gt):
    """utils/loss.py:9-12."""
    return torch.linalg.norm(pred - gt, dim=-1, ord=2).mean(dim=1)
