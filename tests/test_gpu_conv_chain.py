"""egr_conv1x1_chain_f32 (round 5): two 1x1 convolutions back to back in one launch of the fp16 scheme, the 128-channel
intermediate kept in the accumulator registers of the lane that owns the pixel (split under a per-pixel power-of-two scale and fed
back to the matrix cores).  Against fp64, against the two single launches it replaces (egr_conv2d_nhwc_ex_f32 twice: the bar is
theirs - error <= 1.5x + 4e-7 of the result's magnitude, both within 2e-5), in every epilogue mode of the chains on the path:
FPN lateral -> fuse (ReLU, up-sampled residual, ReLU; resnet.py:96-110, 127-133) and the refiners' frame_feat_refined_proj_layers
(ReLU, none; egoposeformer_heatmap_mvf_ex.py:553-563), grouped like the launches of the forward."""
import math

import pytest
import torch
import torch.nn.functional as F

from test_gpu_conv_x6 import pack_w, rnd
from test_gpu_conv_h2 import record_of, record_value

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ops(hip, cin, cmid, cout, G, seed):
    w1 = [rnd(cmid, cin, 1, 1, seed=seed + g, scale=1.0 / math.sqrt(cin)) for g in range(G)]
    w2 = [rnd(cout, cmid, 1, 1, seed=seed + 10 + g, scale=1.0 / math.sqrt(cmid)) for g in range(G)]
    b1, b2 = rnd(G, cmid, seed=seed + 20, scale=0.3), rnd(G, 128, seed=seed + 21, scale=0.3)
    st1 = torch.stack([pack_w(w) for w in w1]) if G > 1 else pack_w(w1[0])
    st2 = torch.stack([pack_w(w) for w in w2]) if G > 1 else pack_w(w2[0])
    p1, p2 = hip.add_wh2(hip.pack_w6(st1.to(DEV))), hip.add_wh2(hip.pack_w6(st2.to(DEV)))
    return w1, w2, b1, b2, p1, p2


def _reference(x, w1, w2, b1, b2, G, act1, act2, res, res_mode, cout):
    n = x.shape[0] // G
    outs = []
    for g in range(G):
        xs = x[g * n:(g + 1) * n].permute(0, 3, 1, 2).double()
        y1 = F.conv2d(xs, w1[g].double(), b1[g].double())
        if act1:
            y1 = F.relu(y1)
        y2 = F.conv2d(y1, w2[g].double(), b2[g][:cout].double())
        if res_mode == 3:
            y2 = y2 + F.interpolate(res[g * n:(g + 1) * n, ..., :cout].permute(0, 3, 1, 2).double(), scale_factor=2, mode="bilinear", align_corners=True)
        elif res_mode == 1:
            y2 = y2 + res[g * n:(g + 1) * n, ..., :cout].permute(0, 3, 1, 2).double()
        if act2:
            y2 = F.relu(y2)
        if res_mode == 2:
            y2 = y2 + res[g * n:(g + 1) * n, ..., :cout].permute(0, 3, 1, 2).double()
        outs.append(y2.permute(0, 2, 3, 1))
    return torch.cat(outs)


CASES = [
    # n (per group), h, w, cin, cout, G, act1, act2, res_mode            what on the path
    (3, 32, 32, 64, 128, 2, 1, 1, 3),        # FPN lateral 0 -> fuse 0 (64 -> 128 -> 128, up-sampled residual), two encoders
    (2, 16, 16, 128, 128, 2, 1, 1, 3),       # FPN lateral 1 -> fuse 1
    (2, 32, 32, 128, 128, 4, 1, 0, 0),       # refiners' frame_feat_refined_proj_layers (conv, ReLU, conv [, up x2 + ReLU behind])
    (1, 20, 12, 64, 124, 1, 1, 1, 1),        # ragged: 240 pixels (7.5 tiles), 124 output channels, residual before the ReLU
    (1, 8, 24, 128, 128, 1, 0, 1, 2),        # no activation between the convs, residual behind the second ReLU
    (5, 8, 8, 64, 128, 3, 1, 0, 0),          # three groups, 320 pixels each
]


@pytest.mark.parametrize("case", CASES, ids=[f"n{c[0]}x{c[1]}x{c[2]}c{c[3]}-{c[4]}G{c[5]}a{c[6]}{c[7]}r{c[8]}" for c in CASES])
def test_chain_matches_fp64_and_the_two_launches(case):
    from egorear_amd import hip
    n, h, w, cin, cout, G, act1, act2, res_mode = case
    cmid = 128
    x = F.relu(rnd(G * n, h, w, cin, seed=7)) * 3.0
    w1, w2, b1, b2, p1, p2 = _ops(hip, cin, cmid, cout, G, 40)
    res = None
    if res_mode == 3:
        res = rnd(G * n, h // 2, w // 2, 128, seed=9)
    elif res_mode:
        res = rnd(G * n, h, w, 128, seed=9)
    ref = _reference(x, w1, w2, b1, b2, G, act1, act2, res, res_mode, cout)
    saved = hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.CHAIN_MIN_ROWS
    hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.CHAIN_MIN_ROWS = 0, 0.0, 0
    try:
        xd = x.to(DEV)
        xin = hip.Img(xd, amax=record_of(xd))
        rimg = hip.Img(res.to(DEV)) if res is not None else None
        b1d, b2d = b1.to(DEV).contiguous(), b2.to(DEV).contiguous()
        assert hip.chain_eligible(xin, p1, p2, cmid, cout, G)
        rec = torch.zeros(64, dtype=torch.int32, device=DEV)
        y = hip.conv1x1_chain(xin, p1, p2, cmid, cout, shift1=b1d, shift2=b2d, act1=act1, act2=act2, res=rimg, res_mode=res_mode,
                              groups=G, amax_out=rec)
        y2 = hip.conv1x1_chain(xin, p1, p2, cmid, cout, shift1=b1d, shift2=b2d, act1=act1, act2=act2, res=rimg, res_mode=res_mode, groups=G)
        # the two launches it replaces, fp16 scheme both (the first leaves the record the second needs)
        r1 = torch.zeros(64, dtype=torch.int32, device=DEV)
        mid = hip.conv2d(xin, p1, cmid, 1, 1, 1, 0, shift=b1d, act=act1, groups=G, amax_out=r1)
        two = hip.conv2d(mid, p2, cout, 1, 1, 1, 0, shift=b2d, act=act2, res=rimg, res_mode=res_mode, groups=G)
    finally:
        hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.CHAIN_MIN_ROWS = saved
    assert torch.equal(y.t, y2.t), "deterministic"
    scale = max(float(ref.abs().max()), 1e-6)
    e_chain = float((y.t.cpu().double() - ref).abs().max()) / scale
    e_two = float((two.t.cpu().double() - ref).abs().max()) / scale
    assert e_chain <= 2e-5 and e_two <= 2e-5, (e_chain, e_two)
    assert e_chain <= 1.5 * e_two + 4e-7, ("the chain is less accurate than the two launches", e_chain, e_two)
    assert record_value(rec) == float(y.t.abs().max()), "the record is the maximum over exactly what the launch stored"
    assert y.amax is rec


def test_every_pixel_gets_its_own_scale():
    """The intermediate is scaled per pixel: with pixels whose activations are 1e-3 / 1e+3 times their neighbours', every pixel's
    result stays within 2e-5 of ITS OWN magnitude (the first product still runs under the per-tensor scale of x - its record -, where
    values 2^-18 below the tensor's maximum start to lose bits; the second product adds nothing on top), and the chain is no worse per
    pixel than the two launches with their per-tensor scale of the intermediate."""
    from egorear_amd import hip
    n, h, w, cin, cout, G = 1, 16, 16, 64, 128, 1
    x = F.relu(rnd(n, h, w, cin, seed=3)) + 0.1
    mult = torch.ones(n, h, w, 1)
    mult[0, ::3, ::2] = 1e-3
    mult[0, 1::5, 1::4] = 1e3
    x = x * mult
    w1, w2, _, _, p1, p2 = _ops(hip, cin, 128, cout, G, 80)
    zero = torch.zeros(1, 128)
    ref = _reference(x, w1, w2, zero, zero, G, 1, 0, None, 0, cout)      # no bias: the output scales with the pixel
    saved = hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.CHAIN_MIN_ROWS
    hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.CHAIN_MIN_ROWS = 0, 0.0, 0
    try:
        xd = x.to(DEV)
        xin = hip.Img(xd, amax=record_of(xd))
        y = hip.conv1x1_chain(xin, p1, p2, 128, cout, act1=1, act2=0, groups=G)
        r1 = torch.zeros(64, dtype=torch.int32, device=DEV)
        mid = hip.conv2d(xin, p1, 128, 1, 1, 1, 0, act=1, groups=G, amax_out=r1)
        two = hip.conv2d(mid, p2, cout, 1, 1, 1, 0, groups=G)
    finally:
        hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.CHAIN_MIN_ROWS = saved
    pix_scale = ref.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    rel = (y.t.cpu().double() - ref).abs() / pix_scale
    rel2 = (two.t.cpu().double() - ref).abs() / pix_scale
    big = (mult.squeeze(-1) >= 1.0)
    assert float(rel[big].max()) <= 2e-6, float(rel[big].max())
    assert float(rel.max()) <= 2e-5, float(rel.max())
    assert float(rel.max()) <= 1.5 * float(rel2.max()) + 4e-7, (float(rel.max()), float(rel2.max()))


def test_refusals():
    from egorear_amd import hip
    w1, w2, b1, b2, p1, p2 = _ops(hip, 64, 128, 128, 1, 90)
    x = rnd(1, 8, 8, 64, seed=1).to(DEV)
    assert not hip.chain_eligible(hip.Img(x), p1, p2, 128, 128, 1)                                  # no abs-max record
    xin = hip.Img(x, amax=record_of(x))
    assert not hip.chain_eligible(xin, p1, p2, 128, 128, 1)                                         # below the pixel threshold
    assert not hip.chain_eligible(xin, p1, p2, 128, 128, 1, scale1=b1)                              # BatchNorm-scaled convs stay single
    with pytest.raises(hip.LaunchError):
        hip.conv1x1_chain(xin, p1, p2, 128, 128)
    x3 = rnd(1, 8, 8, 96, seed=1).to(DEV)
    assert not hip.chain_eligible(hip.Img(x3, amax=record_of(x3)), p1, p2, 128, 128, 1)


BIG = [
    # n (per group), h, w, cout, G, act1, act2, scale of the second half of the intermediate's channels
    (2, 32, 32, 128, 2, 1, 0, 1.0),          # the stereo heads' 256 -> 256 (+ReLU) -> 128 (the ReLU sits behind the up-sampling)
    (2, 16, 16, 128, 4, 1, 1, 1.0),          # four groups (the refiners' heads), ReLU behind the second conv
    (1, 20, 12, 124, 1, 1, 1, 1.0),          # ragged: 240 pixels (7.5 tiles of 32), 124 output channels
    (3, 8, 8, 128, 3, 0, 0, 1.0),            # no activations, three groups of 192 pixels (most waves idle in the last round)
    (1, 16, 16, 128, 1, 1, 0, 1e-5),         # the second half of the intermediate far below the first: its own (clamped) scale
    (1, 16, 16, 128, 1, 1, 0, 3e4),          # ... and far above
]


@pytest.mark.parametrize("case", BIG, ids=[f"n{c[0]}x{c[1]}x{c[2]}-{c[3]}G{c[4]}a{c[5]}{c[6]}s{c[7]:g}" for c in BIG])
def test_streamed_chain_256_256_128_matches_fp64_and_the_two_launches(case):
    """conv_pw2_kernel (round 5): 256 -> 256 -> <= 128 with both weight matrices streamed through an LDS ring and the intermediate
    produced / consumed in two halves of 128 channels, each under its own per-pixel scale."""
    from egorear_amd import hip
    n, h, w, cout, G, act1, act2, half2 = case
    cin = cmid = 256
    x = F.relu(rnd(G * n, h, w, cin, seed=17)) * 2.0
    w1, w2, b1, b2, _, _ = _ops(hip, cin, cmid, cout, G, 140)
    if half2 != 1.0:
        w1 = [torch.cat([wg[:128], wg[128:] * half2]) for wg in w1]
        b1 = torch.cat([b1[:, :128], b1[:, 128:] * half2], 1)
    st1 = torch.stack([pack_w(wg) for wg in w1]) if G > 1 else pack_w(w1[0])
    st2 = torch.stack([pack_w(wg) for wg in w2]) if G > 1 else pack_w(w2[0])
    p1, p2 = hip.add_wh2(hip.pack_w6(st1.to(DEV))), hip.add_wh2(hip.pack_w6(st2.to(DEV)))
    ref = _reference(x, w1, w2, b1, b2, G, act1, act2, None, 0, cout)
    saved = hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.CHAIN_BIG_MIN_ROWS
    hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.CHAIN_BIG_MIN_ROWS = 0, 0.0, 0
    try:
        xd = x.to(DEV)
        xin = hip.Img(xd, amax=record_of(xd))
        b1d, b2d = b1.to(DEV).contiguous(), b2.to(DEV).contiguous()
        assert hip.chain_eligible(xin, p1, p2, cmid, cout, G)
        assert not hip.chain_eligible(xin, p1, p2, cmid, cout, G, res_mode=1)            # no residual in the streamed form
        rec = torch.zeros(64, dtype=torch.int32, device=DEV)
        y = hip.conv1x1_chain(xin, p1, p2, cmid, cout, shift1=b1d, shift2=b2d, act1=act1, act2=act2, groups=G, amax_out=rec)
        y2 = hip.conv1x1_chain(xin, p1, p2, cmid, cout, shift1=b1d, shift2=b2d, act1=act1, act2=act2, groups=G)
        r1 = torch.zeros(64, dtype=torch.int32, device=DEV)
        mid = hip.conv2d(xin, p1, cmid, 1, 1, 1, 0, shift=b1d, act=act1, groups=G, amax_out=r1)
        two = hip.conv2d(mid, p2, cout, 1, 1, 1, 0, shift=b2d, act=act2, groups=G)
    finally:
        hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.CHAIN_BIG_MIN_ROWS = saved
    torch.cuda.synchronize()
    assert torch.equal(y.t, y2.t), "deterministic"
    scale = max(float(ref.abs().max()), 1e-6)
    e_chain = float((y.t.cpu().double() - ref).abs().max()) / scale
    e_two = float((two.t.cpu().double() - ref).abs().max()) / scale
    assert e_chain <= 2e-5 and e_two <= 2e-5, (e_chain, e_two)
    assert e_chain <= 1.5 * e_two + 4e-7, ("the chain is less accurate than the two launches", e_chain, e_two)
    assert record_value(rec) == float(y.t.abs().max()), "the record is the maximum over exactly what the launch stored"


@pytest.mark.parametrize("dead", ["first-all", "first-some", "second-all", "second-some"])
def test_streamed_chain_with_a_dead_half_of_the_intermediate(dead):
    """a pixel whose first 128 post-ReLU intermediate channels are all zero has k2_prev = 60; the second half's scale
    used to be clamped UP to k2_prev - 40 = 20, y1 * 2^20 overflowed fp16 and the pixel's outputs were NaN.  Whole tensors and single
    pixels with a dead first / second half: finite, and at the two launches' accuracy."""
    from egorear_amd import hip
    n, h, w, cout, G = 2, 16, 16, 128, 2
    cin = cmid = 256
    x = F.relu(rnd(G * n, h, w, cin, seed=27)) * 2.0
    w1, w2, b1, b2, _, _ = _ops(hip, cin, cmid, cout, G, 240)
    lo, hi = (0, 128) if dead.startswith("first") else (128, 256)
    b1 = b1.clone()
    b1[:, lo:hi] = -b1[:, lo:hi].abs()                          # the dead half's bias <= 0
    if dead.endswith("all"):
        w1 = [wg.clone() for wg in w1]
        for wg in w1:
            wg[lo:hi] = 0                                         # ... and no input reaches it: dead for every pixel
    else:
        x = x.clone()
        x.view(-1, cin)[::7] = 0                                  # every seventh pixel has no input: its dead half is exactly zero,
        b1[:, hi % 256:(hi % 256) + 128] = b1[:, hi % 256:(hi % 256) + 128].abs() + 0.1      # the other half is its bias (> 0)
    st1, st2 = torch.stack([pack_w(wg) for wg in w1]), torch.stack([pack_w(wg) for wg in w2])
    p1, p2 = hip.add_wh2(hip.pack_w6(st1.to(DEV))), hip.add_wh2(hip.pack_w6(st2.to(DEV)))
    ref = _reference(x, w1, w2, b1, b2, G, 1, 0, None, 0, cout)
    saved = hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.CHAIN_BIG_MIN_ROWS
    hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.CHAIN_BIG_MIN_ROWS = 0, 0.0, 0
    try:
        xd = x.to(DEV)
        xin = hip.Img(xd, amax=record_of(xd))
        b1d, b2d = b1.to(DEV).contiguous(), b2.to(DEV).contiguous()
        assert hip.chain_eligible(xin, p1, p2, cmid, cout, G)
        y = hip.conv1x1_chain(xin, p1, p2, cmid, cout, shift1=b1d, shift2=b2d, act1=1, act2=0, groups=G)
        r1 = torch.zeros(64, dtype=torch.int32, device=DEV)
        mid = hip.conv2d(xin, p1, cmid, 1, 1, 1, 0, shift=b1d, act=1, groups=G, amax_out=r1)
        two = hip.conv2d(mid, p2, cout, 1, 1, 1, 0, shift=b2d, act=0, groups=G)
    finally:
        hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.CHAIN_BIG_MIN_ROWS = saved
    torch.cuda.synchronize()
    assert float(mid.t[..., lo:hi].abs().max()) == 0.0 or dead.endswith("some")
    if dead.endswith("some"):
        assert float(mid.t.view(-1, cmid)[::7, lo:hi].abs().max()) == 0.0 and float(mid.t.view(-1, cmid)[::7].abs().max()) > 0.0
    assert bool(torch.isfinite(y.t).all()), "a dead half of the intermediate produced Inf / NaN"
    scale = max(float(ref.abs().max()), 1e-6)
    e_chain = float((y.t.cpu().double() - ref).abs().max()) / scale
    e_two = float((two.t.cpu().double() - ref).abs().max()) / scale
    assert e_chain <= 2e-5 and e_two <= 2e-5, (e_chain, e_two)
    assert e_chain <= 1.5 * e_two + 4e-7, (e_chain, e_two)
