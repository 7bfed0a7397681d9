"""The fp16-scheme launch of egr_conv2d_nhwc_ex_f32 (EGR_W_F16X2: both operands as two fp16 planes of the value times an exact
power of two, three products on v_mfma_f32_32x32x16_f16, fp32 accumulate) against the fp32-matrix-core launch, the
split-bf16 launch and an fp64 reference, in every forward mode and on every kernel that carries it (generic tiled, persistent,
tap-sharing stride 1 / 2, streaming 1x1), plus what is new with it: the weight image and its per-channel descale, the abs-max
record a launch leaves behind (bit-exact) and the pre-scale a launch derives from its input's record (tiny / huge / wide-range
operands, zeros, Inf / NaN).
The bar is the split-bf16 launch's: as close to fp64 as the fp32 launch (error <= 1.5x + 1e-7 of the result's magnitude), both
within 2e-5."""
import math

import pytest
import torch
import torch.nn.functional as F

from test_gpu_conv_x6 import CONV_CASES, DGRAD_CASES, PW_CASES, TAP2_CASES, TAP_CASES, WGRAD_CASES, WGRAD_KERNEL, judge, pack_w, pack_w_dgrad, rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True, scope="module")
def _tap_kernels_only():
    """This module pins the tap-sharing kernels (ids 2 / 3); the role-split kernel that takes their large launches has its own
    module (test_gpu_conv_tapx.py)."""
    from egorear_amd import hip
    hip.lib.egr_conv_set_tapx(0, -1, -1)
    yield
    hip.lib.egr_conv_set_tapx(1, -1, -1)


def record_of(t: torch.Tensor, slot: int = 0) -> torch.Tensor:
    """An abs-max record (64 int32 slots of float bits) holding max |t| in one slot, as a producing launch would leave it."""
    rec = torch.zeros(64, dtype=torch.int32)
    rec[slot] = torch.tensor([float(t.abs().max())], dtype=torch.float32).view(torch.int32)[0]
    return rec.to(DEV)


def record_value(rec: torch.Tensor) -> float:
    return float(rec.cpu().view(torch.float32).max())


def three(hip, x, wp, *args, **kw):
    """The same launch on the fp32 kernel, the split-bf16 kernel and the fp16-scheme kernel (size rule off).  Returns the three
    outputs, the kernel id of the fp16 launch and the abs-max record it left."""
    wp = wp.to(DEV)
    a = hip.conv2d(x, wp, *args, **kw)
    saved = hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.H2
    hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = 0, 0.0
    try:
        w6 = hip.add_wh2(hip.pack_w6(wp))
        hip.H2 = False
        b = hip.conv2d(x, w6, *args, **kw)
        hip.H2 = True
        xin = hip.Img(x.t, amax=record_of(x.t))
        rec = torch.zeros(64, dtype=torch.int32, device=DEV)
        prof, hip.PROFILE = hip.PROFILE, []
        try:
            c = hip.conv2d(xin, w6, *args, amax_out=rec, **kw)
            tags = [t for name, *_, t in hip.PROFILE if name == "egr_conv2d_nhwc_f32"]
        finally:
            hip.PROFILE = prof
        assert tags and "h2 " in tags[-1], "the third launch must be the fp16-scheme one"
        kern = hip.lib.egr_conv_last_kernel()
    finally:
        hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.H2 = saved
    return a, b, c, kern, rec


def check(a, b, c, ref, rec, what, out_is_planes=False):
    ea, ec = judge(a, c, ref, what)                      # fp16 scheme vs the fp32 launch, both vs fp64
    judge(a, b, ref, what + " [bf16x3]")
    scale = float(ref.abs().max())
    assert float((b - c).abs().max()) <= 4e-6 * scale, (what, "fp16 and bf16 schemes disagree")
    if rec is not None:   # the record is the maximum over exactly what the launch stored
        assert record_value(rec) == float(c.abs().max()), (what, record_value(rec), float(c.abs().max()))
    return ea, ec


def test_pack_wh2_image_and_descale():
    from egorear_amd import hip
    npad, K = 96, 9 * 64
    w = rnd(2, npad, K, seed=3, scale=0.3)
    w[0, 5, 7] = 1.0
    w[0, 6] *= 1e-6                    # a row of tiny weights gets its own scale
    w[1, 0] *= 3.0e4
    w[1, 2] = 0.0                      # an all-zero row: descale clamps, planes are zero
    w6 = hip.add_wh2(hip.pack_w6(w.to(DEV)))
    assert w6.h2_gstride == 4 * (K // 32) * 2048 and w6.h2.numel() == 2 * w6.h2_gstride
    ds = w6.h2_ds.cpu().view(2, npad).double()
    m, e = torch.frexp(ds)
    assert torch.all(m == 0.5), "descales are exact powers of two"
    amax = w.abs().amax(-1).double()
    scaled = amax / ds
    nz = amax > 0
    assert torch.all(scaled[nz] >= 2.0 ** 14) and torch.all(scaled[nz] < 2.0 ** 15)
    assert torch.all(ds[~nz] == 2.0 ** -60)
    img = w6.h2.float().cpu().view(2, 4, K // 32, 2, 2, 64, 8).double()   # [g][frag][chunk][step][plane][lane][j]
    lanes = torch.arange(64)
    for g in range(2):
        for cf in range(4):
            cols = cf * 32 + (lanes & 31)
            for step in range(2):
                for ch in (0, K // 32 - 1, 7):
                    kk = (ch * 32 + step * 16 + 8 * (lanes >> 5))[:, None] + torch.arange(8)[None]
                    h, l = img[g, cf, ch, step, 0], img[g, cf, ch, step, 1]
                    if cf == 3:
                        assert float(h.abs().max()) == 0 and float(l.abs().max()) == 0
                        continue
                    ws = w[g][cols[:, None].expand(64, 8), kk].double() / ds[g][cols][:, None]
                    assert torch.equal(h, ws.to(torch.float16).double()), "h = RNE fp16 of the scaled weight"
                    assert torch.equal(l, (ws - h).to(torch.float16).double()), "l = RNE fp16 of the residual"
                    big = ws.abs() >= 0.125
                    assert float(((ws - h - l).abs() / ws.abs().clamp_min(1e-30))[big].max()) <= 2.0 ** -22


@pytest.mark.parametrize("case", CONV_CASES)
def test_forward_modes(case):
    from egorear_amd import hip
    n, h, w, cin, cout, k, stride, act, res_mode, bn, cfg = case
    pad = k // 2
    x = rnd(n, h, w, cin, seed=1)
    wt = rnd(cout, cin, k, k, seed=2, scale=1.0 / math.sqrt(cin * k * k))
    scale = (rnd(cout, seed=3) * 0.4 + 1.0) if bn else None
    shift = rnd(cout, seed=4)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = rnd(n, ho, wo, cout, seed=5) if res_mode else None
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), None, stride, pad)
    if scale is not None:
        ref = ref * scale.double().view(1, -1, 1, 1)
    ref = ref + shift.double().view(1, -1, 1, 1)
    if res_mode == 1:
        ref = ref + res.permute(0, 3, 1, 2).double()
    ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    if res_mode == 2:
        ref = ref + res.permute(0, 3, 1, 2).double()
    hip.conv_force_config(cfg)
    try:
        a, b, c, _, rec = three(hip, hip.Img(x.to(DEV)), pack_w(wt), cout, k, k, stride, pad, scale=scale.to(DEV) if scale is not None else None,
                                shift=shift.to(DEV), act=act, res=hip.Img(res.to(DEV)) if res is not None else None, res_mode=res_mode)
    finally:
        hip.conv_force_config(-1)
    check(a.t.permute(0, 3, 1, 2), b.t.permute(0, 3, 1, 2), c.t.permute(0, 3, 1, 2), ref, rec, str(case))


def _epilogue_kw(hip, extra, G, n, ho, wo, cout, npad, seed):
    kw, res, sc, sh = dict(groups=G), None, None, None
    if extra in ("res_before", "res_after"):
        res = rnd(G * n, ho, wo, cout, seed=seed)
        kw.update(res=hip.Img(res.to(DEV)), res_mode=hip.RES_BEFORE_ACT if extra == "res_before" else hip.RES_AFTER_ACT, act=hip.ACT_RELU)
    if extra == "up2":
        res = rnd(G * n, ho // 2, wo // 2, cout, seed=seed)
        kw.update(res=hip.Img(res.to(DEV)), res_mode=hip.RES_UP2_BEFORE_ACT, act=hip.ACT_RELU)
    if extra in ("scale_relu", "stride2_scale"):
        sc, sh = rnd(G, npad, seed=seed + 1) * 0.2 + 1.0, rnd(G, npad, seed=seed + 2)
        if G == 1:
            sc, sh = sc[0], sh[0]
        kw.update(scale=sc.to(DEV), shift=sh.to(DEV), act=hip.ACT_RELU if extra == "scale_relu" else hip.ACT_NONE)
    return kw, res, sc, sh


def _reference(x, wts, G, n, stride, pad, extra, res, sc, sh, cout):
    refs = []
    for g in range(G):
        r = F.conv2d(x[g * n:(g + 1) * n].permute(0, 3, 1, 2).double(), wts[g].double(), None, stride, pad)
        rg = res[g * n:(g + 1) * n].permute(0, 3, 1, 2).double() if res is not None else None
        if extra in ("scale_relu", "stride2_scale"):
            s_, b_ = (sc[g], sh[g]) if G > 1 else (sc, sh)
            r = r * s_[:cout].double().view(1, -1, 1, 1) + b_[:cout].double().view(1, -1, 1, 1)
            if extra == "scale_relu":
                r = F.relu(r)
        if extra == "res_before":
            r = F.relu(r + rg)
        if extra == "res_after":
            r = F.relu(r) + rg
        if extra == "up2":
            r = F.relu(r + F.interpolate(rg, scale_factor=2, mode="bilinear", align_corners=True))
        refs.append(r)
    return torch.cat(refs)


@pytest.mark.parametrize("case", [c for c in TAP_CASES if c[5] != "nchw"])
def test_tap_sharing_stride1(case):
    from egorear_amd import hip
    n, hw, cin, cout, G, extra = case
    x = rnd(G * n, hw, hw, cin, seed=70)
    wts = [rnd(cout, cin, 3, 3, seed=71 + g, scale=1.0 / math.sqrt(9 * cin)) for g in range(G)]
    wp = torch.stack([pack_w(t) for t in wts]) if G > 1 else pack_w(wts[0])
    kw, res, sc, sh = _epilogue_kw(hip, extra, G, n, hw, hw, cout, wp.shape[-2], 75)
    a, b, c, kern, rec = three(hip, hip.Img(x.to(DEV)), wp, cout, 3, 3, 1, 1, **kw)
    assert kern == 2
    ref = _reference(x, wts, G, n, 1, 1, extra, res, sc, sh, cout)
    check(a.t.permute(0, 3, 1, 2), b.t.permute(0, 3, 1, 2), c.t.permute(0, 3, 1, 2), ref, rec, f"tap {case}")


@pytest.mark.parametrize("case", TAP2_CASES)
def test_tap_sharing_stride2(case):
    from egorear_amd import hip
    n, hw, cin, cout, G, extra = case
    x = rnd(G * n, hw, hw, cin, seed=170)
    wts = [rnd(cout, cin, 3, 3, seed=171 + g, scale=1.0 / math.sqrt(9 * cin)) for g in range(G)]
    wp = torch.stack([pack_w(t) for t in wts]) if G > 1 else pack_w(wts[0])
    kw, res, sc, sh = _epilogue_kw(hip, extra, G, n, hw // 2, hw // 2, cout, wp.shape[-2], 175)
    a, b, c, kern, rec = three(hip, hip.Img(x.to(DEV)), wp, cout, 3, 3, 2, 1, **kw)
    assert kern == 3
    ref = _reference(x, wts, G, n, 2, 1, extra, res, sc, sh, cout)
    check(a.t.permute(0, 3, 1, 2), b.t.permute(0, 3, 1, 2), c.t.permute(0, 3, 1, 2), ref, rec, f"tap2 {case}")


@pytest.mark.parametrize("case", PW_CASES)
def test_streaming_1x1(case):
    from egorear_amd import hip
    n, h, w, cin, cout, G, extra = case
    stride = 2 if extra.startswith("stride2") else 1
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    x = rnd(G * n, h, w, cin, seed=270)
    wts = [rnd(cout, cin, 1, 1, seed=271 + g, scale=1.0 / math.sqrt(cin)) for g in range(G)]
    wp = torch.stack([pack_w(t) for t in wts]) if G > 1 else pack_w(wts[0])
    kw, res, sc, sh = _epilogue_kw(hip, extra, G, n, ho, wo, cout, wp.shape[-2], 275)
    a, b, c, kern, rec = three(hip, hip.Img(x.to(DEV)), wp, cout, 1, 1, stride, 0, **kw)
    assert kern == 4
    ref = _reference(x, wts, G, n, stride, 0, extra, res, sc, sh, cout)
    check(a.t.permute(0, 3, 1, 2), b.t.permute(0, 3, 1, 2), c.t.permute(0, 3, 1, 2), ref, rec, f"1x1 {case}")


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_data_gradient_modes(case):
    """The training step's data-gradient launches (transposed mode incl. the stride-2 parity classes, plain and with the fused ReLU
    mask + accumulated gradient) on the fp16 scheme."""
    from egorear_amd import hip
    n, h, w, cin, cout, k, s = case
    pad = k // 2
    x = rnd(n, cin, h, w, seed=1).double().requires_grad_(True)
    wt = rnd(cout, cin, k, k, seed=2, scale=1.0 / math.sqrt(cin * k * k))
    y = F.conv2d(x, wt.double(), None, s, pad)
    dy = rnd(*y.shape, seed=3)
    (dx_ref,) = torch.autograd.grad(y, x, dy.double())
    dy_nhwc = hip.Img(dy.permute(0, 2, 3, 1).contiguous().to(DEV))
    a, b, c, kern, rec = three(hip, dy_nhwc, pack_w_dgrad(wt), cin, k, k, s, pad, transposed_out_hw=(h, w))
    check(a.t.permute(0, 3, 1, 2), b.t.permute(0, 3, 1, 2), c.t.permute(0, 3, 1, 2), dx_ref, rec, f"dgrad {case}")
    if cin % 4 == 0:
        act_fwd, prev = rnd(n, h, w, cin, seed=5), rnd(n, h, w, cin, seed=6)
        a, b, c, kern, rec = three(hip, dy_nhwc, pack_w_dgrad(wt), cin, k, k, s, pad, transposed_out_hw=(h, w), res=hip.Img(prev.to(DEV)),
                                   res_mode=hip.RES_BEFORE_ACT, mask=hip.Img(act_fwd.to(DEV)))
        ref = (dx_ref.permute(0, 2, 3, 1) + prev.double()) * (act_fwd > 0)
        check(a.t, b.t, c.t, ref, rec, f"masked dgrad {case}")


@pytest.mark.parametrize("case", [(1, 2, 64, 64, 64, False, 3), (2, 8, 32, 128, 128, True, 3), (1, 32, 16, 256, 96, True, 3),
                                  (1, 16, 64, 128, 128, True, 1), (2, 8, 64, 64, 128, True, 1), (1, 16, 64, 128, 64, False, 1)])
def test_tap_sharing_and_streaming_data_gradient(case):
    """The tap-sharing (3x3) and streaming (1x1) kernels in data-gradient mode, grouped, plain and masked."""
    from egorear_amd import hip
    G, n, hw, cin, cout, masked, k = case          # forward conv cin -> cout; the gradient maps dy (cout) to dx (cin)
    dy = rnd(G * n, hw, hw, cout, seed=81)
    xs, prev = rnd(G * n, hw, hw, cin, seed=82), rnd(G * n, hw, hw, cin, seed=83)
    wts = [rnd(cout, cin, k, k, seed=84 + g, scale=1.0 / math.sqrt(k * k * cin)) for g in range(G)]
    wt = torch.stack([pack_w_dgrad(w) for w in wts]) if G > 1 else pack_w_dgrad(wts[0])
    kw = dict(transposed_out_hw=(hw, hw), groups=G)
    if masked:
        kw.update(res=hip.Img(prev.to(DEV)), res_mode=hip.RES_BEFORE_ACT, mask=hip.Img(xs.to(DEV)))
    a, b, c, kern, rec = three(hip, hip.Img(dy.to(DEV)), wt, cin, k, k, 1, k // 2, **kw)
    assert kern == (2 if k == 3 else 4)
    refs = []
    for g in range(G):
        xr = torch.zeros(n, cin, hw, hw, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(xr, wts[g].double(), None, 1, k // 2)
        (dx_ref,) = torch.autograd.grad(y, xr, dy[g * n:(g + 1) * n].permute(0, 3, 1, 2).double())
        r = dx_ref.permute(0, 2, 3, 1)
        if masked:
            r = (r + prev[g * n:(g + 1) * n].double()) * (xs[g * n:(g + 1) * n] > 0)
        refs.append(r)
    check(a.t, b.t, c.t, torch.cat(refs), rec, f"dgrad kernels {case}")


@pytest.mark.parametrize("case", WGRAD_CASES)
@pytest.mark.parametrize("xs,ds", [(1.0, 1.0), (3e3, 2e-5)])
def test_weight_gradient_fp16_scheme(case, xs, ds):
    """The split weight-gradient launches (generic and tap-sharing kernels) in the fp16 scheme: both operands are activations, each
    pre-scaled from its own abs-max record; against the fp32 launch, the bf16 launch and fp64 autograd."""
    from egorear_amd import hip
    from egorear_amd.engine import unpack_conv_weight
    n, h, w, cin, cout, k, s, G = case
    pad = k // 2
    x = rnd(G * n, cin, h, w, seed=21) * xs
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    dy = rnd(G * n, cout, ho, wo, seed=22) * ds
    ws = torch.empty(1 << 25, device=DEV)
    xt, dyt = x.permute(0, 2, 3, 1).contiguous().to(DEV), dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = []
    saved = hip.H2
    try:
        for fmt, h2 in ((False, False), ("force", False), ("force", True)):
            hip.H2 = h2
            xi, dyi = hip.Img(xt), hip.Img(dyt)
            if h2:
                xi.amax, dyi.amax = record_of(xt), record_of(dyt)
            dw, db = hip.conv2d_wgrad(xi, dyi, k, k, s, pad, ws, want_bias=True, groups=G, x6=fmt)
            out.append(dw.clone())
            assert hip.lib.egr_wgrad_last_kernel() == (WGRAD_KERNEL[WGRAD_CASES.index(case)] if fmt else 0)
            assert hip.lib.egr_wgrad_last_h2() == (1 if h2 else 0)
    finally:
        hip.H2 = saved
    for g in range(G):
        wt = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(x[g * n:(g + 1) * n].double(), wt, None, s, pad)
        (dw_ref,) = torch.autograd.grad(y, wt, dy[g * n:(g + 1) * n].double())
        a, b, c = (unpack_conv_weight(o[g] if G > 1 else o, cin, k, k) for o in out)
        judge(a, c, dw_ref, f"dw fp16 {case} g{g}")
        assert float((b - c).abs().max()) <= 4e-6 * float(dw_ref.abs().max()), ("fp16 and bf16 schemes disagree", case, g)


def test_weight_gradient_arena_makes_the_records():
    """Without records the launch stays on the bf16 scheme; with an arena it reads both operands once and takes the fp16 scheme."""
    from egorear_amd import hip
    x, dy = rnd(8, 32, 32, 64, seed=31).to(DEV), rnd(8, 32, 32, 64, seed=32).to(DEV)
    ws = torch.empty(1 << 24, device=DEV)
    dw0, _ = hip.conv2d_wgrad(hip.Img(x), hip.Img(dy), 3, 3, 1, 1, ws, x6="force")
    assert hip.lib.egr_wgrad_last_h2() == 0
    arena = hip.AmaxArena(torch.device(DEV), records=4)
    arena.begin()
    dw1, _ = hip.conv2d_wgrad(hip.Img(x), hip.Img(dy), 3, 3, 1, 1, ws, x6="force", amax_arena=arena)
    assert hip.lib.egr_wgrad_last_h2() == 1 and arena.k == 2
    assert getattr(x, "_egr_amax", None) is not None and getattr(dy, "_egr_amax", None) is not None
    assert float((dw0 - dw1).abs().max()) <= 4e-6 * float(dw0.abs().max())
    d = hip.ConvDesc()
    d.w_format = 7
    assert hip.lib.egr_conv2d_wgrad_f32(None, None, None, None, None, None, 0, 0, None) != 0
    import ctypes as C
    assert hip.lib.egr_conv2d_wgrad_f32(C.byref(d), x.data_ptr(), dy.data_ptr(), dw0.data_ptr(), None, ws.data_ptr(), ws.numel(), 0, None) == -1
    assert hip.lib.egr_conv2d_wgrad_ex_f32(C.byref(d), x.data_ptr(), dy.data_ptr(), dw0.data_ptr(), None, ws.data_ptr(), ws.numel(), 0, None, None, None) == -2


def test_many_images_repacked_in_two_launches():
    """egr_pack_wh2_many_f32 (the training step's per-update refresh) == egr_pack_wh2_f32 per operand, bit for bit."""
    from egorear_amd import hip
    ops = []
    for i, (g, n, k) in enumerate([(1, 64, 576), (2, 128, 1152), (1, 32, 64), (3, 96, 288)]):
        w = (rnd(g, n, k, seed=400 + i) if g > 1 else rnd(n, k, seed=400 + i)) * (10.0 ** (i - 2))
        ops.append(hip.add_wh2(hip.pack_w6(w.to(DEV))))
    want = [(o.h2.clone(), o.h2_ds.clone()) for o in ops]
    for o in ops:
        o.h2.zero_()
        o.h2_ds.zero_()
    hip.WH2Table(ops).run()
    for o, (img, ds) in zip(ops, want):
        assert torch.equal(o.h2.view(torch.int16), img.view(torch.int16)) and torch.equal(o.h2_ds, ds)


def test_grouped_split_k_and_persistent_launches():
    """Split-K (the slabs hold descaled partial sums, the reduction pass leaves the record) and the persistent short-K kernel."""
    from egorear_amd import hip
    G, n, h, cin, cout, k = 3, 2, 16, 64, 96, 3
    x = rnd(G * n, h, h, cin, seed=21)
    wts = [rnd(cout, cin, k, k, seed=30 + g, scale=1 / 24) for g in range(G)]
    wp = torch.stack([pack_w(t) for t in wts])
    ws = torch.empty(1 << 22, device=DEV)
    a, b, c, kern, rec = three(hip, hip.Img(x.to(DEV)), wp, cout, k, k, 1, 1, groups=G, split_k=3, workspace=ws, act=hip.ACT_RELU)
    ref = torch.cat([F.relu(F.conv2d(x[g * n:(g + 1) * n].permute(0, 3, 1, 2).double(), wts[g].double(), None, 1, 1)) for g in range(G)])
    check(a.t.permute(0, 3, 1, 2), b.t.permute(0, 3, 1, 2), c.t.permute(0, 3, 1, 2), ref, rec, "grouped split-K")
    # persistent launch: K = 256 1x1 (outside the streaming kernel's K) over enough tiles
    n, h, cin, cout = 40, 64, 256, 128
    x = rnd(n, h, h, cin, seed=41)
    wt = rnd(cout, cin, 1, 1, seed=42, scale=1 / 16)
    hip.conv_set_persist(64, 8)
    try:
        a, b, c, kern, rec = three(hip, hip.Img(x.to(DEV)), pack_w(wt), cout, 1, 1, 1, 0, act=hip.ACT_RELU)
    finally:
        hip.conv_set_persist(512, 4)
    assert kern == 1
    ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double()))
    check(a.t.permute(0, 3, 1, 2), b.t.permute(0, 3, 1, 2), c.t.permute(0, 3, 1, 2), ref, rec, "persistent")


@pytest.mark.parametrize("xscale,wscale", [(1e-6, 1.0), (3e4, 1e-3), (1e-12, 1e6), (2.0 ** 40, 2.0 ** -50)])
def test_prescale_follows_the_record(xscale, wscale):
    """Operands far outside fp16's range: the power-of-two pre-scales bring them in, the result keeps fp32-launch accuracy."""
    from egorear_amd import hip
    n, hw, cin, cout = 8, 32, 64, 128
    x = rnd(n, hw, hw, cin, seed=1) * xscale
    wt = rnd(cout, cin, 3, 3, seed=2, scale=1.0 / 24) * wscale
    a, b, c, kern, rec = three(hip, hip.Img(x.to(DEV)), pack_w(wt), cout, 3, 3, 1, 1)
    assert kern == 2
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), None, 1, 1)
    check(a.t.permute(0, 3, 1, 2), b.t.permute(0, 3, 1, 2), c.t.permute(0, 3, 1, 2), ref, rec, f"scales {xscale} {wscale}")


def test_wide_dynamic_range_zero_rows_and_a_loose_record():
    """Operands spanning many binades below the maximum lose nothing that matters (absolute error 2^-40 of the largest magnitude);
    a record that overstates max |x| by 2^6 (an inherited bound) still serves."""
    from egorear_amd import hip
    n, h, cin, cout = 4, 32, 64, 64
    g = torch.Generator().manual_seed(5)
    x = rnd(n, h, h, cin, seed=1) * torch.exp2(torch.randint(-20, 4, (n, h, h, cin), generator=g).float())
    x[0, :4] = 0.0
    wt = rnd(cout, cin, 3, 3, seed=2) * torch.exp2(torch.randint(-12, 4, (cout, cin, 3, 3), generator=g).float())
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), None, 1, 1)
    a, b, c, kern, rec = three(hip, hip.Img(x.to(DEV)), pack_w(wt), cout, 3, 3, 1, 1)
    check(a.t.permute(0, 3, 1, 2), b.t.permute(0, 3, 1, 2), c.t.permute(0, 3, 1, 2), ref, rec, "dynamic range")
    saved = hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS
    hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = 0, 0.0
    try:
        w6 = hip.add_wh2(hip.pack_w6(pack_w(wt).to(DEV)))
        loose = hip.conv2d(hip.Img(x.to(DEV), amax=record_of(x * 64.0, slot=17)), w6, cout, 3, 3, 1, 1)
    finally:
        hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = saved
    judge(a.t.permute(0, 3, 1, 2), loose.t.permute(0, 3, 1, 2), ref, "loose record")


def test_nan_and_inf_propagate():
    from egorear_amd import hip
    n, h, cin, cout = 8, 32, 64, 64
    x = rnd(n, h, h, cin, seed=1)
    wt = rnd(cout, cin, 3, 3, seed=2, scale=1 / 24)
    saved = hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS
    hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = 0, 0.0
    try:
        w6 = hip.add_wh2(hip.pack_w6(pack_w(wt).to(DEV)))
        xn = x.clone()
        xn[3, 10, 10, 5] = float("nan")
        rec = record_of(x)                                  # (a NaN never enters a record: fmax drops it)
        y = hip.conv2d(hip.Img(xn.to(DEV), amax=rec), w6, cout, 3, 3, 1, 1).t.cpu()
        assert hip.lib.egr_conv_last_kernel() == 2
        bad = torch.isnan(y).any(-1)
        assert bad[3, 9:12, 9:12].all() and int(bad.sum()) == 9     # exactly the 3 x 3 pixels that see the NaN
        xi = x.clone()
        xi[2, 5, 5, 0] = float("inf")
        reci = torch.zeros(64, dtype=torch.int32)
        reci[0] = 0x7f800000                                # the producer stored an Inf: the record says so
        y = hip.conv2d(hip.Img(xi.to(DEV), amax=reci.to(DEV)), w6, cout, 3, 3, 1, 1).t.cpu()
        nonfinite = ~torch.isfinite(y).all(-1)
        assert nonfinite[2, 4:7, 4:7].all()                 # no finite garbage where the fp32 launch has Inf / NaN
    finally:
        hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = saved


def test_operand_placement_and_inherited_record():
    """Input as a channel slice behind an image map, output into a channel slice; the record travels on the tensor object and
    through max-pool / up-sampling (their outputs are bounded by their input)."""
    from egorear_amd import hip
    V, B, h, cin, cout = 4, 8, 16, 64, 128
    n = V * B
    x = rnd(n, h, h, cin, seed=411)
    stored = torch.zeros(B, V, h, h, 2 * cin)
    stored[..., cin:] = x.view(V, B, h, h, cin).permute(1, 0, 2, 3, 4)
    stored[..., :cin] = 9.0
    stored = stored.to(DEV)
    wt = rnd(cout, cin, 3, 3, seed=412, scale=1.0 / math.sqrt(9 * cin))
    ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), None, 2, 1))
    img = h * h * 2 * cin
    saved = hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS
    hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = 0, 0.0
    try:
        w6 = hip.add_wh2(hip.pack_w6(pack_w(wt).to(DEV)))
        cat = torch.full((n, h // 2, h // 2, 2 * cout), 5.0, device=DEV)
        rec = torch.zeros(64, dtype=torch.int32, device=DEV)
        out = hip.Img(cat[..., cout:])
        hip.conv2d(hip.Img(stored.view(n, h, h, 2 * cin)[..., cin:], amax=record_of(x)), w6, cout, 3, 3, 2, 1, act=hip.ACT_RELU,
                   xmap=hip.NMap(B, V * img, img), out=out, amax_out=rec)
        assert hip.lib.egr_conv_last_kernel() == 3
        got = cat[..., cout:].permute(0, 3, 1, 2).double().cpu()
        assert float((got - ref).abs().max()) / float(ref.abs().max()) <= 2e-5
        assert float((cat[..., :cout] - 5.0).abs().max()) == 0
        assert out.amax is rec and record_value(rec) == float(got.abs().max())
        # the record rides on the tensor object and is inherited by the bounded ops
        y = hip.conv2d(hip.Img(x.to(DEV), amax=record_of(x)), w6, cout, 3, 3, 2, 1, act=hip.ACT_RELU, amax_out=torch.zeros(64, dtype=torch.int32, device=DEV))
        assert hip.Img(y.t).amax is y.amax and y.amax is not None
        assert hip.upsample2x(y).amax is y.amax and hip.maxpool(y, 3, 2, 1).amax is y.amax
        # a launch that keeps no record clears a stale one
        z = hip.conv2d(hip.Img(x.to(DEV)), w6, cout, 3, 3, 2, 1, out=hip.Img(y.t))
        assert z.amax is None and hip.Img(y.t).amax is None
    finally:
        hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = saved


def test_fp16_scheme_needs_its_side_operands():
    """The C entry refuses an EGR_W_F16X2 launch without record / descale, or a record with channel-major output."""
    import ctypes as C
    from egorear_amd import hip
    x = rnd(2, 16, 16, 64, seed=1).to(DEV)
    wt = rnd(64, 64, 3, 3, seed=2, scale=1 / 24)
    w6 = hip.add_wh2(hip.pack_w6(pack_w(wt).to(DEV)))
    y = torch.empty(2, 16, 16, 64, device=DEV)
    d = hip.ConvDesc()
    d.n, d.h, d.w, d.cin, d.cout, d.kh, d.kw, d.stride, d.pad, d.ho, d.wo = 2, 16, 16, 64, 64, 3, 3, 1, 1, 16, 16
    d.ldx = d.ldy = 64
    d.xmap = d.ymap = hip.NMap(2, 16 * 16 * 64, 0)
    d.rmap = hip.NMap(1, 0, 0)
    d.groups, d.split_k, d.w_format = 1, 1, 4
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call(aux):
        return hip.lib.egr_conv2d_nhwc_ex_f32(C.byref(d), x.data_ptr(), w6.h2.data_ptr(), None, None, None, None, None, y.data_ptr(), None, 0,
                                              C.byref(aux) if aux is not None else None, s)
    rec = record_of(x)
    assert call(None) == -2 and call(hip.ConvAux(None, rec.data_ptr(), None)) == -2 and call(hip.ConvAux(w6.h2_ds.data_ptr(), None, None)) == -2
    assert call(hip.ConvAux(w6.h2_ds.data_ptr(), rec.data_ptr(), None)) == 0
    d.out_nchw, d.w_format = 1, 0
    out = torch.zeros(64, dtype=torch.int32, device=DEV)
    assert hip.lib.egr_conv2d_nhwc_ex_f32(C.byref(d), x.data_ptr(), pack_w(wt).to(DEV).data_ptr(), None, None, None, None, None, y.data_ptr(), None, 0,
                                          C.byref(hip.ConvAux(None, None, out.data_ptr())), s) == -1
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", [(1, 2, 64, 64, 64, "res_before"), (2, 2, 32, 128, 128, "scale_relu"), (1, 8, 16, 256, 64, "plain")])
def test_tap_sharing_kernel_64_row_tiles_for_small_launches(case):
    """Launches with fewer than 256 tiles of 128 x 64 (batch 1) take 64-row tiles on conv_igemm_tap_kernel - twice the workgroups on the
    chip; against fp64 and bit for bit against the 128-row tiling (egr_conv_set_tap(3)): the tile height does not touch the K order."""
    from egorear_amd import hip
    G, n, hw, cin, cout, extra = case
    x = rnd(G * n, hw, hw, cin, seed=901)
    wts = [rnd(cout, cin, 3, 3, seed=902 + g, scale=1.0 / math.sqrt(9 * cin)) for g in range(G)]
    wp = torch.stack([pack_w(t) for t in wts]) if G > 1 else pack_w(wts[0])
    kw, res, sc, sh = _epilogue_kw(hip, extra, G, n, hw, hw, cout, wp.shape[-2], 905)
    hip.lib.egr_conv_set_tapx(0, -1, -1)
    try:
        a, b, c, kern, rec = three(hip, hip.Img(x.to(DEV)), wp, cout, 3, 3, 1, 1, **kw)
        assert kern == 2
        ref = _reference(x, wts, G, n, 1, 1, extra, res, sc, sh, cout)
        check(a.t.permute(0, 3, 1, 2), b.t.permute(0, 3, 1, 2), c.t.permute(0, 3, 1, 2), ref, rec, f"tap 64-row {case}")
        hip.lib.egr_conv_set_tap(3)
        _, _, c0, kern0, rec0 = three(hip, hip.Img(x.to(DEV)), wp, cout, 3, 3, 1, 1, **kw)
        assert kern0 == 2 and torch.equal(c.t, c0.t) and record_value(rec) == record_value(rec0)
    finally:
        hip.lib.egr_conv_set_tap(1)
        hip.lib.egr_conv_set_tapx(1, -1, -1)
