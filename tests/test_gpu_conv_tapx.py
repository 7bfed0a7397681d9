"""conv_tapx_kernel (egorear_amd/csrc/egr_conv_tapx.hip): the fp16 scheme's 3x3 / pad 1 forward launches on role-split persistent
workgroups - four multiplying waves, four waves that load / split the activations and run the PREVIOUS tile's epilogue under the
current tile's K loop (accumulators handed over through a 64-KB LDS staging area, half a tile at a time).  Forced here from one
tile up and with fewer workgroups than tiles, so that a workgroup walks several tiles, groups and column tiles, with the
XCD-contiguous tile order (8 workgroups) and the plain one (5).  Against fp64, the fp32 launch, the bf16 scheme - and bit for bit
against the tap-sharing kernels it replaces (same products in the same order)."""
import math

import pytest
import torch
import torch.nn.functional as F

from test_gpu_conv_h2 import DEV, _epilogue_kw, _reference, check, record_of, record_value, three
from test_gpu_conv_x6 import pack_w, rnd

pytestmark = pytest.mark.gpu

S1_CASES = [
    # n, h(=w), cin, cout, groups, extras
    (2, 64, 64, 64, 1, "res_before"),          # 512 x 64 tiles (eight image rows), residual + ReLU, four chunks = the epilogue slices fill the K loop
    (8, 32, 128, 128, 2, "scale_relu"),        # 256 x 128 tiles, grouped, BatchNorm affine + ReLU
    (4, 64, 96, 256, 1, "res_after"),          # six chunks, two column tiles per row tile, residual behind the activation
    (32, 64, 64, 128, 1, "scale_relu"),        # 512 tiles
    (16, 64, 64, 64, 1, "res_before"),         # 128 tiles of 512 x 64
    (256, 16, 64, 192, 1, "scale_relu"),       # 512 x 64 tiles = two whole 16 x 16 images, three column tiles
    (128, 8, 128, 128, 1, "plain"),            # 8 x 8 images: a 256-pixel tile is four whole images
    (24, 16, 256, 512, 2, "res_before"),       # layer3 / layer4 like: sixteen chunks, four column tiles, grouped, tile count not a multiple of 8
    (6, 32, 64, 128, 1, "plain"),              # 24 tiles: the last round of the walk is ragged
]
S2_CASES = [
    # n, h(=w, input), cin, cout, groups, extras
    (16, 32, 128, 256, 2, "scale_relu"),       # layer3 entry, grouped: 16 x 16 outputs, a tile is half an image
    (64, 16, 256, 512, 1, "plain"),            # layer4 entry: 8 x 8 outputs, a tile is two whole images, two column tiles
    (4, 64, 64, 256, 1, "scale_relu"),         # 32-pixel output rows: four rows per tile
    (8, 64, 128, 512, 2, "res_after"),         # the refiners' 256 -> 512 geometry (narrower), residual behind the ReLU
    (6, 32, 96, 256, 1, "plain"),              # six chunks, 12 tiles
]


MODE = {"on": 1}      # egr_conv_set_tapx's first argument in force: 2 = the 128 x 32 wave tile (whole tile through the staging area), 3 = 128 x 64 (two halves)


@pytest.fixture(params=[(5, 2), (8, 2), (5, 3), (8, 3)], ids=["5wg-w32", "8wg-w32", "5wg-w64", "8wg-w64"])
def tapx(request):
    from egorear_amd import hip
    blocks, MODE["on"] = request.param
    hip.lib.egr_conv_set_tapx(MODE["on"], 1, blocks)
    yield hip
    hip.lib.egr_conv_set_tapx(1, 256, 256)


def _run(hip, case, stride, seed):
    n, hw, cin, cout, G, extra = case
    ho = hw // stride
    x = rnd(G * n, hw, hw, cin, seed=seed)
    wts = [rnd(cout, cin, 3, 3, seed=seed + 1 + g, scale=1.0 / math.sqrt(9 * cin)) for g in range(G)]
    wp = torch.stack([pack_w(t) for t in wts]) if G > 1 else pack_w(wts[0])
    kw, res, sc, sh = _epilogue_kw(hip, extra, G, n, ho, ho, cout, wp.shape[-2], seed + 5)
    a, b, c, kern, rec = three(hip, hip.Img(x.to(DEV)), wp, cout, 3, 3, stride, 1, **kw)
    assert kern == 6, "the role-split kernel must carry this launch"
    ref = _reference(x, wts, G, n, stride, 1, extra, res, sc, sh, cout)
    check(a.t.permute(0, 3, 1, 2), b.t.permute(0, 3, 1, 2), c.t.permute(0, 3, 1, 2), ref, rec, f"tapx s{stride} {case}")
    # the kernels it replaces: same products, same order
    hip.lib.egr_conv_set_tapx(0, -1, -1)
    try:
        _, _, c0, kern0, rec0 = three(hip, hip.Img(x.to(DEV)), wp, cout, 3, 3, stride, 1, **kw)
    finally:
        hip.lib.egr_conv_set_tapx(MODE["on"], -1, -1)
    assert kern0 in (1, 2, 3)
    if kern0 == 1:        # (too few rows for the tap-sharing kernels: the generic split kernel walks the taps in another order)
        assert float((c.t - c0.t).abs().max()) <= 2e-6 * float(c0.t.abs().max()), case
    else:
        assert torch.equal(c.t, c0.t), (case, float((c.t - c0.t).abs().max()))
        assert record_value(rec) == record_value(rec0)


@pytest.mark.parametrize("case", S1_CASES)
def test_role_split_stride1(tapx, case):
    _run(tapx, case, 1, 70)


@pytest.mark.parametrize("case", S2_CASES)
def test_role_split_stride2(tapx, case):
    _run(tapx, case, 2, 170)


@pytest.mark.parametrize("case", [(8, 64, 64, 128, 1, "scale_relu"),      # layer2 entry: 128 output channels exist as 128 x 128 tiles only
                                  (16, 32, 128, 128, 2, "res_after")])
def test_role_split_stride2_128_channels(case):
    from egorear_amd import hip
    MODE["on"] = 1
    hip.lib.egr_conv_set_tapx(1, 1, 8)
    try:
        _run(hip, case, 2, 270)
    finally:
        hip.lib.egr_conv_set_tapx(1, 256, 256)


PW_CASES = [
    # n, h(=w), cin, cout, groups, extras        1x1 convolutions with >= 256 input channels: chunks of 64 channels, four k16 steps each
    (8, 32, 256, 256, 2, "scale_relu"),        # the heads' 256 -> 256 (egoposeformer_heatmap_mvf_ex.py:101-126), grouped
    (16, 16, 256, 128, 1, "plain"),            # 256 -> 128: 128 x 128 tiles only
    (4, 32, 512, 128, 4, "scale_relu"),        # the refiners' 512 -> 128 (:525-532), four groups, eight chunks
    (2, 64, 320, 256, 1, "res_after"),         # five chunks' worth of channels is not a multiple of ... (320 = 5 x 64: odd chunk count -> stays on the tiled kernel)
    (6, 32, 384, 512, 1, "res_before"),        # six chunks, two column tiles, residual
]


@pytest.mark.parametrize("case", PW_CASES)
def test_role_split_1x1(tapx, case):
    hip = tapx
    n, hw, cin, cout, G, extra = case
    x = rnd(G * n, hw, hw, cin, seed=370)
    wts = [rnd(cout, cin, 1, 1, seed=371 + g, scale=1.0 / math.sqrt(cin)) for g in range(G)]
    wp = torch.stack([pack_w(t) for t in wts]) if G > 1 else pack_w(wts[0])
    kw, res, sc, sh = _epilogue_kw(hip, extra, G, n, hw, hw, cout, wp.shape[-2], 375)
    a, b, c, kern, rec = three(hip, hip.Img(x.to(DEV)), wp, cout, 1, 1, 1, 0, **kw)
    wide_only = MODE["on"] == 3 and cout % 256 != 0         # the 128 x 64 wave tile forced, but only 128 x 128 tiles exist for this width
    assert kern == (1 if (cin // 64) % 2 or wide_only else 6), (case, kern)
    ref = _reference(x, wts, G, n, 1, 0, extra, res, sc, sh, cout)
    check(a.t.permute(0, 3, 1, 2), b.t.permute(0, 3, 1, 2), c.t.permute(0, 3, 1, 2), ref, rec, f"tapx 1x1 {case}")


def test_launches_outside_its_cover_stay_on_the_tap_kernels(tapx):
    hip = tapx
    for (n, hw, cin, cout, stride, want) in [(32, 16, 32, 192, 1, 2),      # two chunks only
                                             (8, 64, 64, 192, 2, 3),      # stride 2 with 192 output channels
                                             (9, 32, 64, 62, 1, 2)]:      # channel count not a multiple of four
        x = rnd(n, hw, hw, cin, seed=1)
        wt = rnd(cout, cin, 3, 3, seed=2, scale=1.0 / math.sqrt(9 * cin))
        a, b, c, kern, rec = three(hip, hip.Img(x.to(DEV)), pack_w(wt), cout, 3, 3, stride, 1)
        assert kern == want, (n, hw, cin, cout, stride, kern)
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), None, stride, 1)
        check(a.t.permute(0, 3, 1, 2), b.t.permute(0, 3, 1, 2), c.t.permute(0, 3, 1, 2), ref, rec, "fallback")


@pytest.mark.parametrize("stride", [1, 2])
def test_operand_placement_nan_and_scales(tapx, stride):
    """Input as a channel slice behind an image map, output into a channel slice of a wider tensor (its neighbours untouched); a NaN
    reaches exactly the outputs that see it; operands far outside fp16's range."""
    hip = tapx
    V, B, h, cin, cout = 4, 8, 32, 64, 256
    n, ho = V * B, h // stride
    x = rnd(n, h, h, cin, seed=411) * 3e-5
    stored = torch.zeros(B, V, h, h, 2 * cin)
    stored[..., cin:] = x.view(V, B, h, h, cin).permute(1, 0, 2, 3, 4)
    stored[..., :cin] = 9.0
    stored = stored.to(DEV)
    wt = rnd(cout, cin, 3, 3, seed=412, scale=1.0 / math.sqrt(9 * cin)) * 2e3
    ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), None, stride, 1))
    img = h * h * 2 * cin
    saved = hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS
    hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = 0, 0.0
    try:
        w6 = hip.add_wh2(hip.pack_w6(pack_w(wt).to(DEV)))
        cat = torch.full((n, ho, ho, 2 * cout), 5.0, device=DEV)
        rec = torch.zeros(64, dtype=torch.int32, device=DEV)
        out = hip.Img(cat[..., cout:])
        hip.conv2d(hip.Img(stored.view(n, h, h, 2 * cin)[..., cin:], amax=record_of(x)), w6, cout, 3, 3, stride, 1, act=hip.ACT_RELU,
                   xmap=hip.NMap(B, V * img, img), out=out, amax_out=rec)
        assert hip.lib.egr_conv_last_kernel() == 6
        got = cat[..., cout:].permute(0, 3, 1, 2).double().cpu()
        assert float((got - ref).abs().max()) / float(ref.abs().max()) <= 2e-5
        assert float((cat[..., :cout] - 5.0).abs().max()) == 0
        assert record_value(rec) == float(got.abs().max())
        xn = x.clone()
        xn[3, 10, 10, 5] = float("nan")
        y = hip.conv2d(hip.Img(xn.to(DEV), amax=record_of(x)), w6, cout, 3, 3, stride, 1).t.cpu()
        assert hip.lib.egr_conv_last_kernel() == 6
        bad = torch.isnan(y).any(-1)
        want = torch.zeros_like(bad)
        for oy in range(ho):
            for ox in range(ho):
                if abs(oy * stride - 10) <= 1 and abs(ox * stride - 10) <= 1:
                    want[3, oy, ox] = True
        assert torch.equal(bad, want)
    finally:
        hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = saved


# ---- training launches on the role-split kernel (template parameter TR): the data gradient of a 3x3 / stride-1 conv (mirrored taps),
# with the fused ReLU mask + accumulated gradient; the train-mode BatchNorm statistics of a raw conv output in the epilogue
@pytest.fixture(params=[(5, 2), (8, 3), (8, 2), (5, 3)], ids=["5wg-w32", "8wg-w64", "8wg-w32", "5wg-w64"])
def tapx_train(request):
    from egorear_amd import hip
    blocks, MODE["on"] = request.param          # (the wide wave tile exists for the statistics epilogue at stride 1 only: other training launches take the narrow one)
    hip.lib.egr_conv_set_tapx(MODE["on"], 1, blocks)
    yield hip
    hip.lib.egr_conv_set_tapx(1, 256, 256)


DGRAD_CASES = [
    # groups, n, h(=w), forward cin, forward cout, mode
    (2, 8, 32, 128, 128, "masked_res"),        # 128 x 128 tiles, grouped, mask + the gradient accumulated so far
    (1, 4, 64, 64, 64, "masked_res"),          # 256 x 64 tiles (layer1)
    (1, 4, 64, 64, 64, "masked"),              # mask only
    (1, 32, 16, 256, 256, "masked_res"),       # eight chunks, two column tiles
    (2, 32, 8, 512, 512, "plain"),             # plain data gradient, 8 x 8 images (two per tile), four column tiles
    (1, 16, 64, 128, 64, "res"),               # gradient into a 64-channel tensor from 128 channels, accumulated, no mask
    (1, 6, 32, 64, 128, "masked_res"),         # 24 tiles of 256 x 64: ragged last round
]


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_role_split_data_gradient(tapx_train, case):
    from test_gpu_conv_x6 import pack_w_dgrad
    hip = tapx_train
    G, n, hw, cin, cout, mode = case           # forward conv cin -> cout; the gradient maps dy (cout channels) to dx (cin)
    dy = rnd(G * n, hw, hw, cout, seed=481)
    xs, prev = rnd(G * n, hw, hw, cin, seed=482), rnd(G * n, hw, hw, cin, seed=483)
    wts = [rnd(cout, cin, 3, 3, seed=484 + g, scale=1.0 / math.sqrt(9 * cin)) for g in range(G)]
    wt = torch.stack([pack_w_dgrad(w) for w in wts]) if G > 1 else pack_w_dgrad(wts[0])
    kw = dict(transposed_out_hw=(hw, hw), groups=G)
    if "res" in mode:
        kw.update(res=hip.Img(prev.to(DEV)), res_mode=hip.RES_BEFORE_ACT)
    if "masked" in mode:
        kw.update(mask=hip.Img(xs.to(DEV)))
    a, b, c, kern, rec = three(hip, hip.Img(dy.to(DEV)), wt, cin, 3, 3, 1, 1, **kw)
    assert kern == 6, "the role-split kernel must carry this launch"
    refs = []
    for g in range(G):
        xr = torch.zeros(n, cin, hw, hw, dtype=torch.float64, requires_grad=True)
        (dx_ref,) = torch.autograd.grad(F.conv2d(xr, wts[g].double(), None, 1, 1), xr, dy[g * n:(g + 1) * n].permute(0, 3, 1, 2).double())
        r = dx_ref.permute(0, 2, 3, 1)
        if "res" in mode:
            r = r + prev[g * n:(g + 1) * n].double()
        if "masked" in mode:
            r = r * (xs[g * n:(g + 1) * n] > 0)
        refs.append(r)
    check(a.t, b.t, c.t, torch.cat(refs), rec, f"tapx dgrad {case}")
    hip.lib.egr_conv_set_tapx(0, -1, -1)
    try:
        _, _, c0, kern0, rec0 = three(hip, hip.Img(dy.to(DEV)), wt, cin, 3, 3, 1, 1, **kw)
    finally:
        hip.lib.egr_conv_set_tapx(MODE["on"], -1, -1)
    assert kern0 == 2
    assert torch.equal(c.t, c0.t), (case, float((c.t - c0.t).abs().max()))
    assert record_value(rec) == record_value(rec0)


BN_CASES = [
    # n, h(=w), cin, cout, stride, groups
    (8, 32, 64, 64, 1, 2),                     # 256 x 64 tiles, grouped
    (16, 64, 64, 64, 1, 1),                    # layer1 geometry
    (4, 32, 128, 128, 1, 1),                   # 128 x 128 tiles
    (32, 8, 256, 512, 1, 2),                   # 8 x 8 images: two per tile, four column tiles
    (8, 32, 64, 128, 2, 2),                    # stride 2 (layer2 entry): four chunks - the slab goes out right in front of the next hand-over
    (8, 32, 128, 256, 2, 1),                   # stride 2, two column tiles
]


@pytest.mark.parametrize("case", BN_CASES)
def test_role_split_batchnorm_statistics(tapx_train, case):
    """egr_conv_aux.bn_partials on the role-split kernel: the same output tile, and per-tile slabs that finalise to the statistics,
    running buffers, normalised output and bounds of the tap-sharing kernels' epilogue and of the pass over the tensor."""
    from egorear_amd import hip_train as T
    hip = tapx_train
    n, hw, cin, cout, s, G = case
    x = rnd(G * n, hw, hw, cin, seed=531).to(DEV)
    x[:, :, :, 3] += 7.0                       # (a channel far from zero mean on the input side)
    wts = [rnd(cout, cin, 3, 3, seed=532 + g, scale=1.0 / math.sqrt(9 * cin)) for g in range(G)]
    wp = (torch.stack([pack_w(t) for t in wts]) if G > 1 else pack_w(wts[0])).to(DEV)
    gamma, beta = (rnd(G, cout, seed=535) + 1.5).to(DEV), rnd(G, cout, seed=536).to(DEV)
    ws = T.bn_workspace(DEV)
    saved = hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS
    hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = 0, 0.0
    outs = []
    try:
        w = hip.add_wh2(hip.pack_w6(wp))
        xin = hip.Img(x, amax=record_of(x))
        for route in ("tapx", "tap", "pass"):
            hip.lib.egr_conv_set_tapx(MODE["on"] if route == "tapx" else 0, -1, -1)
            slabs = []
            rec = torch.zeros(64, dtype=torch.int32, device=DEV)
            kw = dict(bn_ws=ws, bn_slabs=slabs) if route != "pass" else {}
            y = hip.conv2d(xin, w, cout, 3, 3, s, 1, groups=G, amax_out=rec, **kw).t
            kern = hip.lib.egr_conv_last_kernel()
            assert kern == (6 if route == "tapx" else (2 if s == 1 else 3)), (route, kern)
            rm, rv = torch.zeros(G, cout, device=DEV), torch.ones(G, cout, device=DEV)
            r = torch.zeros(64, dtype=torch.int32, device=DEV)
            o, ctx = T.bn_train(y, gamma, beta, rm, rv, G, ws, relu=True, amax_out=r, slabs=slabs[0] if slabs else None)
            torch.cuda.synchronize()
            outs.append((y.clone(), o.clone(), ctx.mean.clone(), ctx.invstd.clone(), rm, rv, ctx.xhat_max.clone(), r, record_value(rec)))
    finally:
        hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = saved
        hip.lib.egr_conv_set_tapx(MODE["on"], -1, -1)
    t = outs[0]
    for u, what in ((outs[1], "tap-sharing epilogue"), (outs[2], "statistics pass")):
        assert torch.equal(t[0], u[0]), what                                     # the conv output itself
        assert t[8] == u[8]
        for i, name in ((2, "mean"), (3, "invstd"), (4, "running_mean"), (5, "running_var"), (1, "y")):
            scale = float(u[i].abs().max())
            assert float((t[i] - u[i]).abs().max()) <= 2e-6 * scale, (what, name, float((t[i] - u[i]).abs().max()), scale)
        assert torch.equal(t[6], u[6]) and torch.equal(t[7], u[7]), what         # extremes are exact on every route
    # against fp64 statistics of the stored tensor
    yv = t[0].double().view(G, -1, cout)
    assert float((t[2].double().view(G, cout) - yv.mean(1)).abs().max()) <= 1e-6 * float(yv.abs().max())


def test_training_launch_refusals_and_the_forced_wave_tile(tapx_train):
    """A launch with both a statistics epilogue request and an activation is refused (EINVAL) before any kernel is chosen, slabs that
    do not fit with EWORKSPACE; the statistics epilogue runs on either wave tile."""
    hip = tapx_train
    from egorear_amd import hip_train as T
    x = rnd(8, 32, 32, 64, seed=601).to(DEV)
    wp = pack_w(rnd(128, 64, 3, 3, seed=602, scale=0.05)).to(DEV)
    saved = hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS
    hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = 0, 0.0
    try:
        w = hip.add_wh2(hip.pack_w6(wp))
        xin = hip.Img(x, amax=record_of(x))
        ws = T.bn_workspace(DEV)
        with pytest.raises(hip.LaunchError) as ei:
            hip.conv2d(xin, w, 128, 3, 3, 1, 1, act=hip.ACT_RELU, bn_ws=ws, bn_slabs=[])
        assert ei.value.code == hip.EINVAL
        with pytest.raises(hip.LaunchError) as ei:
            hip.conv2d(xin, w, 128, 3, 3, 1, 1, bn_ws=ws[:64], bn_slabs=[])
        assert ei.value.code == hip.EWORKSPACE
        for on, bm in ((2, 128), (3, 256)):                 # slabs: one per M tile of the wave tile in force
            hip.lib.egr_conv_set_tapx(on, -1, -1)
            slabs = []
            hip.conv2d(xin, w, 128, 3, 3, 1, 1, bn_ws=ws, bn_slabs=slabs)
            assert hip.lib.egr_conv_last_kernel() == 6 and slabs[0] == 8 * 32 * 32 // bm
    finally:
        hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = saved
        hip.lib.egr_conv_set_tapx(1, -1, -1)
