"""Training-step kernels (include/egorear_train.h) against PyTorch-CPU autograd in fp64 — the per-op half of the §8(f)
rank-2 parity bar; the assembled step is checked in test_gpu_train_step.py."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def close(got, ref, rel=2e-5, what=""):
    ref = ref.detach().double()
    tol = rel * max(float(ref.abs().max()), 1e-6)
    err = float((got.double().cpu() - ref).abs().max())
    assert err <= tol, f"{what} max err {err:.3e} > tol {tol:.3e}"


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("c,n,h,groups", [(64, 4, 32, 2), (128, 6, 16, 1), (512, 4, 8, 2), (64, 2, 128, 2)])
def test_batchnorm_train_forward_backward(c, n, h, groups):
    from egorear_amd import hip_train as T
    ws = T.bn_workspace(DEV)
    x = rnd(groups * n, c, h, h, seed=1) * 2 + 0.3
    res = rnd(groups * n, c, h, h, seed=2)
    gamma, beta = rnd(groups, c, seed=3) + 1.5, rnd(groups, c, seed=4)
    rm, rv = rnd(groups, c, seed=5), rnd(groups, c, seed=6) + 2
    dy = rnd(groups * n, c, h, h, seed=7)
    # reference: each group is its own BatchNorm2d in training mode, followed by +res, ReLU
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rr = res.double().requires_grad_(True)
    rms, rvs, ys = [], [], []
    for g in range(groups):
        m, v = rm[g].double().clone(), rv[g].double().clone()
        ys.append(F.relu(F.batch_norm(xr[g * n:(g + 1) * n], m, v, gr[g], br[g], True, 0.1, 1e-5) + rr[g * n:(g + 1) * n]))
        rms.append(m)
        rvs.append(v)
    yref = torch.cat(ys)
    dx_ref, dg_ref, db_ref, dres_ref = torch.autograd.grad(yref, (xr, gr, br, rr), dy.double())
    rm_d, rv_d = rm.to(DEV).contiguous(), rv.to(DEV).contiguous()
    y, ctx = T.bn_train(nhwc(x).to(DEV), gamma.to(DEV), beta.to(DEV), rm_d, rv_d, groups, ws, res=nhwc(res).to(DEV), relu=True)
    close(y.permute(0, 3, 1, 2), yref, what="y")
    close(rm_d, torch.stack(rms), rel=1e-6, what="running_mean")
    close(rv_d, torch.stack(rvs), rel=1e-6, what="running_var")
    dx, dgam, dbet, dz = T.bn_backward(ctx, nhwc(dy).to(DEV), y, ws, want_dz=True)
    close(dx.permute(0, 3, 1, 2), dx_ref, rel=1e-4, what="dx")
    close(dgam, dg_ref, rel=1e-4, what="dgamma")
    close(dbet, db_ref, rel=1e-4, what="dbeta")
    close(dz.permute(0, 3, 1, 2), dres_ref, what="dres")
    # no ReLU, no residual (the downsample branch)
    y2, ctx2 = T.bn_train(nhwc(x).to(DEV), gamma.to(DEV), beta.to(DEV), None, None, groups, ws, relu=False)
    y2ref = torch.cat([F.batch_norm(xr[g * n:(g + 1) * n], None, None, gr[g], br[g], True, 0.1, 1e-5) for g in range(groups)])
    close(y2.permute(0, 3, 1, 2), y2ref, what="y2")
    dx2_ref, = torch.autograd.grad(y2ref, xr, dy.double())
    dx2, _, _, _ = T.bn_backward(ctx2, nhwc(dy).to(DEV), None, ws)
    close(dx2.permute(0, 3, 1, 2), dx2_ref, rel=1e-4, what="dx2")


def _rec_value(rec):
    return float(rec.cpu().view(torch.float32).max())


@pytest.mark.parametrize("c,n,h,groups,shift", [(64, 4, 32, 2, 0.3), (128, 6, 16, 1, 25.0), (512, 4, 8, 2, -3.0)])
def test_batchnorm_records_are_upper_bounds_from_the_batch_extremes(c, n, h, groups, shift):
    """egr_bn_stats_ex_f32 / egr_bn_backward_ex_f32: the abs-max records of the normalised output and of its gradient are BOUNDS built
    from per-channel batch extremes (no pass over the tensors): never below the true maximum, and tight enough to cost at most two of
    the fp16 scheme's 22 bits (<= 4x) - also when the channel means sit far from zero (shift = 25 standard deviations)."""
    from egorear_amd import hip, hip_train as T
    ws = T.bn_workspace(DEV)
    x = (rnd(groups * n, h, h, c, seed=11) * 2 + shift).to(DEV)
    res = rnd(groups * n, h, h, c, seed=12).to(DEV)
    gamma, beta = (rnd(groups, c, seed=13) + 1.5).to(DEV), rnd(groups, c, seed=14).to(DEV)
    dy = rnd(groups * n, h, h, c, seed=17).to(DEV)
    arena = hip.AmaxArena(torch.device(DEV), records=8)
    arena.begin()
    hip.absmax_record(res, arena.new())                       # the residual carries a record (a conv launch's, in the step)
    for with_res in (True, False):
        rec = arena.new()
        y, ctx = T.bn_train(x, gamma, beta, None, None, groups, ws, res=res if with_res else None, relu=True, amax_out=rec)
        true = float(y.abs().max())
        assert y._egr_amax is rec and true <= _rec_value(rec) <= 4.0 * true, (with_res, true, _rec_value(rec))
        assert ctx.xhat_max is not None
        xh = ((x.view(groups, -1, c) - ctx.mean.view(groups, 1, c)) * ctx.invstd.view(groups, 1, c)).abs().amax(1)
        assert float((ctx.xhat_max / xh).min()) >= 1.0 - 1e-5 and float((ctx.xhat_max / xh).max()) <= 1.0 + 1e-4
        rdx = arena.new()
        dx, _, _, dz = T.bn_backward(ctx, dy, y, ws, want_dz=with_res, amax_dx=rdx)
        true = float(dx.abs().max())
        assert dx._egr_amax is rdx and true <= _rec_value(rdx) <= 6.0 * true, (with_res, true, _rec_value(rdx))
    # a residual without a record: no bound can be given - the output stays untagged (the consumer reads it once instead)
    y, _ = T.bn_train(x, gamma, beta, None, None, groups, ws, res=res.clone(), relu=True, amax_out=arena.new())
    assert getattr(y, "_egr_amax", None) is None


@pytest.mark.parametrize("n,hw,cin,cout,k,s,G,split", [(8, 32, 64, 64, 3, 1, 2, True), (8, 32, 64, 128, 3, 2, 2, True), (8, 32, 64, 128, 1, 2, 1, True),
                                                   (4, 16, 128, 128, 3, 1, 1, False), (2, 8, 256, 512, 3, 1, 2, True), (16, 64, 64, 64, 3, 1, 1, True)])
def test_batchnorm_statistics_from_the_conv_epilogue(n, hw, cin, cout, k, s, G, split):
    """egr_conv_aux.bn_partials + egr_bn_finalize_f32: the conv launch leaves the per-tile channel statistics of its output, the
    BatchNorm only finalises them - same batch statistics, running buffers, normalised output and bounds as the pass over the tensor
    (tap-sharing, stride-2 tap-sharing, generic split and fp32 launches)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_conv_x6 import pack_w
    from egorear_amd import hip, hip_train as T
    x = rnd(G * n, hw, hw, cin, seed=31).to(DEV)
    wts = [rnd(cout, cin, k, k, seed=32 + g, scale=1.0 / math.sqrt(cin * k * k)) for g in range(G)]
    wp = (torch.stack([pack_w(t) for t in wts]) if G > 1 else pack_w(wts[0])).to(DEV)
    gamma, beta = (rnd(G, cout, seed=35) + 1.5).to(DEV), rnd(G, cout, seed=36).to(DEV)
    ws = T.bn_workspace(DEV)
    saved = hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS
    if split:
        hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = 0, 0.0
    try:
        w = hip.add_wh2(hip.pack_w6(wp)) if split else wp
        xin = hip.Img(x)
        if split:
            xin.amax = torch.zeros(64, dtype=torch.int32, device=DEV)
            hip.absmax_record(x, xin.amax)
        slabs = []
        y1 = hip.conv2d(xin, w, cout, k, k, s, k // 2, groups=G, bn_ws=ws, bn_slabs=slabs).t
        assert len(slabs) == 1 and slabs[0] > 0
        rm1, rv1 = torch.zeros(G, cout, device=DEV), torch.ones(G, cout, device=DEV)
        r1 = torch.zeros(64, dtype=torch.int32, device=DEV)
        o1, c1 = T.bn_train(y1, gamma, beta, rm1, rv1, G, ws, relu=True, amax_out=r1, slabs=slabs[0])
        y2 = hip.conv2d(xin, w, cout, k, k, s, k // 2, groups=G).t
        assert torch.equal(y1, y2)                                   # the statistics epilogue stores the same tile
        rm2, rv2 = torch.zeros(G, cout, device=DEV), torch.ones(G, cout, device=DEV)
        r2 = torch.zeros(64, dtype=torch.int32, device=DEV)
        o2, c2 = T.bn_train(y2, gamma, beta, rm2, rv2, G, ws, relu=True, amax_out=r2)
    finally:
        hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = saved
    for a_, b_, what in ((c1.mean, c2.mean, "mean"), (c1.invstd, c2.invstd, "invstd"), (rm1, rm2, "running_mean"), (rv1, rv2, "running_var")):
        close(a_, b_.cpu(), rel=2e-6, what=what)
    close(o1, o2.cpu(), rel=2e-6, what="y")
    assert torch.equal(c1.xhat_max, c2.xhat_max) and torch.equal(r1, r2)          # extremes are exact in both routes
    # refusals: an activation, a residual or a narrow output cannot carry the statistics
    with pytest.raises(RuntimeError):
        hip.conv2d(xin, w, cout, k, k, s, k // 2, groups=G, act=hip.ACT_RELU, bn_ws=ws, bn_slabs=[])
    with pytest.raises(RuntimeError):
        hip.conv2d(xin, w, cout, k, k, s, k // 2, groups=G, bn_ws=ws[:64], bn_slabs=[])      # the slabs do not fit


def test_records_follow_the_elementwise_launches():
    """Outputs bounded by their inputs get a record from the inputs' records: a sum (max|a| + max|b|, one 64-thread launch), a masked
    gradient and a max-pool output (the same record), pooling / up-sampling gradients (a constant factor)."""
    from egorear_amd import hip, hip_train as T
    arena = hip.AmaxArena(torch.device(DEV), records=16)
    arena.begin()
    T.set_arena(arena)
    try:
        a, b = (rnd(4, 16, 16, 64, seed=21) * 3).to(DEV), rnd(4, 16, 16, 64, seed=22).to(DEV)
        hip.absmax_record(a, arena.new())
        hip.absmax_record(b, arena.new())
        s_ = T.add(a, b)
        ma, mb = float(a.abs().max()), float(b.abs().max())
        assert float(s_.abs().max()) <= _rec_value(s_._egr_amax) and abs(_rec_value(s_._egr_amax) - (ma + mb)) <= 1e-5 * (ma + mb)
        assert getattr(T.add(a, b.clone()), "_egr_amax", None) is None            # one input without a record: none
        assert T.relu_bwd(a, b)._egr_amax is a._egr_amax
        y, slot = T.maxpool_train(hip.Img(a), 3, 2, 1)
        assert y.amax is a._egr_amax and y.t._egr_amax is a._egr_amax
        dy = rnd(*y.t.shape, seed=23).to(DEV)
        hip.absmax_record(dy, arena.new())
        dx = T.maxpool_bwd(dy, slot, (16, 16), 3, 2, 1)
        assert float(dx.abs().max()) <= _rec_value(dx._egr_amax) <= 4.0 * float(dy.abs().max()) * 1.00001
        du = T.upsample2x_bwd(a)
        assert float(du.abs().max()) <= _rec_value(du._egr_amax) <= 6.0 * ma * 1.00001
        # the C entry: refusals and the arithmetic
        r = torch.zeros(64, dtype=torch.int32, device=DEV)
        assert hip.lib.egr_record_bound_f32(None, None, 1.0, 1.0, r.data_ptr(), None) == -2
        assert hip.lib.egr_record_bound_f32(a._egr_amax.data_ptr(), None, -1.0, 1.0, r.data_ptr(), None) == -1
    finally:
        T.set_arena(None)


def test_elementwise():
    from egorear_amd import hip_train as T
    a, b = rnd(3, 7, 16, seed=1), rnd(3, 7, 16, seed=2)
    close(T.add(a.to(DEV), b.to(DEV)), a + b, rel=1e-7)
    close(T.relu_bwd(a.to(DEV), b.to(DEV)), a * (b > 0), rel=1e-7)
    z = (rnd(5, 64, seed=3) * 4).double().requires_grad_(True)
    h = F.gelu(z)
    dh = rnd(5, 64, seed=4)
    dz_ref, = torch.autograd.grad(h, z, dh.double())
    close(T.gelu(z.detach().float().to(DEV)), h.detach(), rel=2e-6)
    close(T.gelu_bwd(dh.to(DEV), z.detach().float().to(DEV)), dz_ref, rel=2e-6)
    x = rnd(9, 8, seed=5).to(DEV)
    m = torch.tensor([1, 0, 1, 1, 0, 0, 1, 0, 1], dtype=torch.uint8)
    close(T.rowmask_(x.clone(), m.to(DEV)), x.cpu() * m[:, None], rel=1e-7)
    assert float(T.zeros((4, 4), DEV).abs().sum()) == 0.0


@pytest.mark.parametrize("k,s,p,h", [(3, 2, 1, 32), (2, 2, 0, 32), (3, 2, 1, 13)])
def test_maxpool_train_and_backward(k, s, p, h):
    from egorear_amd import hip, hip_train as T
    x = rnd(3, 8, h, h + 2, seed=1)
    x[:, :, ::3, ::2] = 0.25  # ties inside windows: the first maximum in scan order must get the gradient
    xr = x.double().requires_grad_(True)
    y = F.max_pool2d(xr, k, s, p)
    dy = rnd(*y.shape, seed=2)
    dx_ref, = torch.autograd.grad(y, xr, dy.double())
    yi, slot = T.maxpool_train(hip.Img(nhwc(x).to(DEV)), k, s, p)
    close(yi.t.permute(0, 3, 1, 2), y.detach(), rel=1e-7)
    dx = T.maxpool_bwd(nhwc(dy).to(DEV), slot, (h, h + 2), k, s, p)
    close(dx.permute(0, 3, 1, 2), dx_ref, rel=1e-6)


@pytest.mark.parametrize("h,w,relu", [(8, 8, False), (16, 12, True), (1, 5, False), (32, 32, True)])
def test_upsample2x_backward(h, w, relu):
    from egorear_amd import hip, hip_train as T
    x = rnd(2, 8, h, w, seed=1)
    xr = x.double().requires_grad_(True)
    y = F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=True)
    if relu:
        y = F.relu(y)
    dy = rnd(*y.shape, seed=2)
    dx_ref, = torch.autograd.grad(y, xr, dy.double())
    yd = hip.upsample2x(hip.Img(nhwc(x).to(DEV)), relu=relu).t
    dx = T.upsample2x_bwd(nhwc(dy).to(DEV), yd if relu else None)
    close(dx.permute(0, 3, 1, 2), dx_ref, rel=1e-5)


def test_planes_to_nhwc_and_stem_im2col():
    from egorear_amd import hip, hip_train as T
    from egorear_amd.hip import NMap
    B, V, J, hw = 3, 4, 15, 64
    planes = rnd(B, V, J, hw, seed=1).to(DEV)
    # view-major image order n = v*B + b over a (B,V,...) tensor
    y = T.planes_to_nhwc(planes, NMap(B, V * J * hw, J * hw), V * B, J, hw, 32)
    ref = torch.zeros(V * B, hw, 32)
    ref[:, :, :J] = planes.cpu().permute(1, 0, 3, 2).reshape(V * B, hw, J)
    close(y, ref, rel=1e-7)
    # stem weight gradient through im2col + the 1x1 wgrad
    img = rnd(2, 4, 3, 32, 32, seed=2)
    w = rnd(64, 3, 7, 7, seed=3, scale=0.1).double().requires_grad_(True)
    views = img[:, 1:3].permute(1, 0, 2, 3, 4).reshape(4, 3, 32, 32)   # view-major
    yv = F.conv2d(views.double(), w, None, 2, 3)
    dy = rnd(*yv.shape, seed=4)
    dw_ref, = torch.autograd.grad(yv, w, dy.double())
    cols = T.stem_im2col(img.to(DEV), 1, 2)
    ws = torch.empty(1 << 22, device=DEV)
    dw, _ = hip.conv2d_wgrad(hip.Img(cols.view(-1, 1, 1, 160)), hip.Img(nhwc(dy).reshape(-1, 1, 1, 64).to(DEV)), 1, 1, 1, 0, ws)
    close(dw[:, :147].reshape(64, 3, 7, 7), dw_ref, rel=3e-5)
    assert float(dw[:, 147:].abs().max()) == 0.0


@pytest.mark.parametrize("c,groups", [(256, 4), (128, 1)])
def test_layernorm_backward(c, groups):
    from egorear_amd import hip, hip_train as T
    rows = groups * 30
    x, res = rnd(rows, c, seed=1), rnd(rows, c, seed=2)
    gamma, beta = rnd(groups, c, seed=3) + 1.2, rnd(groups, c, seed=4)
    dy = rnd(rows, c, seed=5)
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    pre = xr + res.double()
    y = torch.cat([F.layer_norm(pre[g * 30:(g + 1) * 30], (c,), gr[g], br[g], 1e-5) for g in range(groups)])
    dx_ref, dg_ref, db_ref = torch.autograd.grad(y, (xr, gr, br), dy.double())
    pre_d = T.add(x.to(DEV), res.to(DEV))
    ds, dgam, dbet = T.layernorm_bwd(dy.to(DEV), pre_d, gamma.reshape(-1).to(DEV), groups)
    close(ds, dx_ref, rel=2e-5)
    close(dgam.view(groups, c), dg_ref, rel=2e-5)
    close(dbet.view(groups, c), db_ref, rel=2e-5)
    yk = hip.layernorm(x.to(DEV), gamma.reshape(-1).to(DEV), beta.reshape(-1).to(DEV), res=res.to(DEV), groups=groups)
    close(yk, y.detach(), rel=2e-5)


@pytest.mark.parametrize("J,heads,d", [(15, 4, 64), (16, 4, 32)])
def test_joint_mha_backward(J, heads, d):
    from egorear_amd import hip_train as T
    B, C = 5, heads * d
    qkv = rnd(B * J, 3 * C, seed=1)
    dout = rnd(B * J, C, seed=2)
    t = qkv.double().requires_grad_(True)
    q, k, v = (t[:, i * C:(i + 1) * C].reshape(B, J, heads, d).permute(0, 2, 1, 3) for i in range(3))
    att = ((q @ k.transpose(-2, -1)) * d ** -0.5).softmax(-1) @ v
    out = att.permute(0, 2, 1, 3).reshape(B * J, C)
    ref, = torch.autograd.grad(out, t, dout.double())
    got = T.joint_mha_bwd(qkv.to(DEV), dout.to(DEV), B, J, heads, d, d ** -0.5)
    close(got, ref, rel=3e-5)


@pytest.mark.parametrize("with_pos,groups,C,cf", [(True, 2, 256, 128), (False, 1, 128, 128)])
def test_msda_gather_backward(with_pos, groups, C, cf):
    """Against autograd through the reference formulation (project every token, then mmcv-style sampling: oracle msda_core)."""
    from egorear_amd import hip, hip_train as T
    from oracle.egorear_oracle import msda_core
    B, V, J, heads, H, W = 2, 4, 5, 4, 16, 16
    dh = C // heads
    feat = rnd(V, B, H * W, cf, seed=1)
    pos = rnd(groups, V, H * W, C, seed=2) if with_pos else None
    ol = rnd(groups * B * J, heads * 48, seed=3)
    ol[:, :heads * 32] *= 6.0                              # offsets of several pixels: some samples leave the map
    anchors = (rnd(B, V, J, 2, seed=4) * 0.5 + 0.5)
    anchors[0, 0, 0] = torch.tensor([0.0, 0.0])            # corner anchor: out-of-bounds corners
    anchors[1, 3, 2] = torch.tensor([63 / 64, 63 / 64])
    valid = torch.ones(B, V, J, dtype=torch.uint8)
    valid[1, 2, 3] = 0
    Wfold = rnd(groups, C, cf, seed=5, scale=1 / math.sqrt(cf))
    cfold = rnd(groups, C, seed=6)
    da = rnd(groups, B * J * V, C, seed=7)                 # upstream gradient of a (rows (b,j,v), C)
    featr = feat.double().requires_grad_(True)
    posr = pos.double().requires_grad_(True) if with_pos else None
    olr = ol.double().requires_grad_(True)
    outs = []
    for g in range(groups):
        o_g = olr[g * B * J:(g + 1) * B * J]
        off = o_g[:, :heads * 32].reshape(B, J, heads, 16, 2)
        aw = o_g[:, heads * 32:].reshape(B, J, heads, 16).softmax(-1)
        per_view = []
        for v in range(V):
            value = featr[v] @ Wfold[g].double().t() + cfold[g].double()
            if with_pos:
                value = value + posr[g, v]
            loc = anchors[:, v].double()[:, :, None, None, :] + off / torch.tensor([W, H], dtype=torch.float64)
            a = msda_core(value.reshape(B, H * W, heads, dh), H, W, loc, aw)               # (B, J, C)
            per_view.append(a * valid[:, v, :, None].double())
        outs.append(torch.stack(per_view, dim=2).reshape(B * J * V, C))                    # rows (b, j, v)
    out = torch.stack(outs)
    wanted = (olr, featr) + ((posr,) if with_pos else ())
    grads = torch.autograd.grad(out, wanted, da.double())
    dol_ref, dfeat_ref = grads[0], grads[1]
    # kernel inputs: dg_h = Wfold_h^T da_h
    da_m = da * valid.permute(0, 2, 1).reshape(1, B * J * V, 1).float()                    # masked rows carry no gradient
    dg = torch.einsum("grhd,ghdc->grhc", da_m.reshape(groups, -1, heads, dh), Wfold.reshape(groups, heads, dh, cf))
    dfeat = T.zeros(feat.shape, DEV)
    dpos = T.zeros(pos.shape, DEV) if with_pos else None
    dol = T.msda_gather_bwd(feat.to(DEV), pos.to(DEV) if with_pos else None, ol.to(DEV), anchors.to(DEV), valid.to(DEV), B, V, J,
                            heads, dh, H, W, dg.contiguous().to(DEV), da_m.reshape(-1, C).contiguous().to(DEV), cfold.to(DEV),
                            dfeat, dpos, groups)
    dol_sum = T.fold_rows(dol, V)
    close(dol_sum, dol_ref, rel=5e-5, what="d offsets/logits")
    close(dfeat, dfeat_ref, rel=5e-5, what="d feat")
    if with_pos:
        close(dpos, grads[2], rel=5e-5, what="d pos")


def test_small_reductions():
    from egorear_amd import hip_train as T
    x = rnd(2, 50, 96, seed=1)
    sc = rnd(2, 50, seed=2)
    got = T.colsum(x.to(DEV), 96, 50, 64, scale=sc.to(DEV), groups=2, gx=50 * 96, gs=50)
    close(got, (x[:, :, :64] * sc[:, :, None]).sum(1), rel=1e-5)
    got2 = T.colsum(x.to(DEV), 96, 100, 96)
    close(got2, x.reshape(100, 96).sum(0, keepdim=True), rel=1e-5)
    close(T.fold_rows(x.reshape(100, 96).to(DEV), 4), x.reshape(25, 4, 96).sum(1), rel=1e-6)
    dx = rnd(8, 15, 32, seed=3)
    de, db = T.jqa_sum_bwd(dx.to(DEV), 8, 15, 32, 4)
    close(de, dx.reshape(4, 2, 15, 32).sum(1), rel=1e-6)
    close(db, dx.sum(1), rel=1e-6)


@pytest.mark.parametrize("d,rows", [(3, 32), (64, 960)])
def test_rownorm_loss(d, rows):
    from egorear_amd import hip_train as T
    pred, gt = rnd(rows, d, seed=1), rnd(rows, d, seed=2)
    gt[3] = pred[3]                                        # zero norm: torch's norm backward gives a zero subgradient
    pr = pred.double().requires_grad_(True)
    loss_ref = torch.mean(torch.linalg.norm(gt.double() - pr, dim=-1, ord=2)) * 10.0
    g_ref, = torch.autograd.grad(loss_ref, pr)
    loss = torch.zeros(1, dtype=torch.float64, device=DEV)
    dp = T.rownorm_loss(pred.to(DEV), gt.to(DEV), d, 10.0, loss)
    assert abs(float(loss) - float(loss_ref.detach())) <= 1e-5 * abs(float(loss_ref.detach()))
    close(dp, g_ref, rel=1e-5)


def test_sumsq_and_adamw_match_torch():
    from egorear_amd import hip_train as T
    n = 4096 + 64
    p0, g1, g2 = rnd(n, seed=1), rnd(n, seed=2) * 3, rnd(n, seed=3) * 0.01
    for wd in (0.0, 5e-4):
        ref = torch.nn.Parameter(p0.clone())
        opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=wd)
        p, m, v = p0.clone().to(DEV), T.zeros((n,), DEV), T.zeros((n,), DEV)
        ss = torch.zeros(1, dtype=torch.float64, device=DEV)
        for step, g in enumerate((g1, g2), 1):
            ref.grad = g.clone()
            total = torch.nn.utils.clip_grad_norm_([ref], 5.0)
            opt.step()
            gd = g.to(DEV)
            T.sumsq(gd, ss)
            assert abs(math.sqrt(float(ss)) - float(total)) <= 1e-5 * float(total)
            T.adamw(p, gd, m, v, 1e-3, 0.9, 0.999, 1e-8, wd, step, ss, 5.0)
            close(p, ref.detach(), rel=2e-6, what=f"wd {wd} step {step}")


def test_repack_descriptor_kinds():
    """egr_repack_f32: forward operand, data-gradient operand, padded vectors and gradient un-packing, incl. channel slices,
    row-concatenated sources and zero padding, against plain torch indexing."""
    from egorear_amd import repack
    from egorear_amd.engine import pack_conv_weight, unpack_conv_weight
    cout, cin_tot, ci0, cin, kh = 40, 48, 16, 20, 3           # a 20-channel slice of a 48-channel 3x3 weight; everything ragged
    taps, cin_pad, cout_pad = kh * kh, 32, 64
    w = rnd(cout, cin_tot, kh, kh, seed=1).to(DEV)
    bias = rnd(cout, seed=2).to(DEV)
    tbl = repack.RepackTable(torch.device(DEV))
    K = cin_pad * taps
    fwd = torch.full((cout_pad, K), 7.0, device=DEV)
    tbl.add(repack.FWD, w, fwd, 0, rows=cout, cin=cin, cin_tot=cin_tot, ci0=ci0, cin_pad=cin_pad, taps=taps, rows_pad=cout_pad, total=cout_pad * K)
    Kt = cout_pad * taps
    dg = torch.full((cin_pad, Kt), 7.0, device=DEV)
    tbl.add(repack.DGRAD, w, dg, 0, rows=cout, cin=cin, cin_tot=cin_tot, ci0=ci0, cin_pad=cin_pad, taps=taps, rows_pad=cout_pad, k_off=0,
            k_tot=cout_pad, total=cin_pad * Kt)
    bp = torch.full((cout_pad,), 7.0, device=DEV)
    tbl.add(repack.COPYPAD, bias, bp, 0, rows=cout, total=cout_pad)
    # gradient side: a packed (cout, cin_pad/32, taps, 32) tensor back into the channel slice of an OIHW gradient
    gpk = rnd(cout, K, seed=3).to(DEV)
    gdst = torch.zeros(cout, cin_tot, kh, kh, device=DEV)
    tbl.add(repack.UNPACK, gpk, gdst, 0, rows=cout, cin=cin, cin_tot=cin_tot, ci0=ci0, cin_pad=cin_pad, taps=taps, total=cout * cin * taps)
    # row-concatenated linear sources into one data-gradient operand (the q|k|v projection)
    qa, qb = rnd(32, 64, seed=4).to(DEV), rnd(48, 64, seed=5).to(DEV)
    cat_t = torch.full((64, 96), 7.0, device=DEV)
    tbl.add(repack.DGRAD, qa, cat_t, 0, rows=32, cin=64, cin_tot=64, cin_pad=64, taps=1, rows_pad=32, k_off=0, k_tot=96, total=64 * 32)
    tbl.add(repack.DGRAD, qb, cat_t, 0, rows=48, cin=64, cin_tot=64, cin_pad=64, taps=1, rows_pad=64, k_off=32, k_tot=96, total=64 * 64)
    tbl.run()
    torch.cuda.synchronize()
    wsl = torch.zeros(cout_pad, cin_pad, kh, kh, device=DEV)
    wsl[:cout, :cin] = w[:, ci0:ci0 + cin]
    assert torch.equal(fwd, pack_conv_weight(wsl))
    assert torch.equal(dg, pack_conv_weight(wsl.transpose(0, 1).contiguous()))
    assert torch.equal(bp[:cout], bias) and float(bp[cout:].abs().max()) == 0.0
    ref = torch.zeros_like(gdst)
    ref[:, ci0:ci0 + cin] = unpack_conv_weight(gpk, cin_pad, kh, kh)[:, :cin]
    assert torch.equal(gdst, ref)
    assert torch.equal(cat_t[:, :80], torch.cat([qa, qb], 0).t()) and float(cat_t[:, 80:].abs().max()) == 0.0


def test_stem_forward_raw_and_weight_gradient():
    """Training-mode stem: the inference kernel without BatchNorm / ReLU, and egr_stem_wgrad_f32 against autograd (two groups)."""
    from egorear_amd import hip, hip_train as T
    B, V, H = 3, 4, 64
    img = rnd(B, V, 3, H, H, seed=1)
    ws = [rnd(64, 3, 7, 7, seed=2 + g, scale=0.1) for g in range(2)]
    wp = torch.zeros(2, 64, 148)
    for g in range(2):
        wp[g, :, :147] = ws[g].reshape(64, 147)
    y = hip.stem(img.to(DEV), 0, 2, wp.to(DEV), None, None, groups=2).t          # group g = views 2g, 2g+1
    dy = rnd(*y.shape, seed=5)
    wsb = torch.empty(2 * 512 * 64 * 160, device=DEV)
    dw = T.stem_wgrad(img.to(DEV), 0, 2, dy.to(DEV), wsb, groups=2)
    n = 2 * B
    for g in range(2):
        views = img[:, 2 * g:2 * g + 2].permute(1, 0, 2, 3, 4).reshape(n, 3, H, H)
        w = ws[g].double().requires_grad_(True)
        ref = F.conv2d(views.double(), w, None, 2, 3)
        close(y[g * n:(g + 1) * n].permute(0, 3, 1, 2), ref.detach(), rel=2e-5, what="raw stem")
        dref, = torch.autograd.grad(ref, w, dy[g * n:(g + 1) * n].permute(0, 3, 1, 2).double())
        close(dw[g], dref, rel=3e-5, what="stem wgrad")


@pytest.mark.parametrize("n,hw,cin,cout,G,bias,acc", [
    (512, 1, 128, 128, 1, True, False),      # a transformer layer's Linear
    (2048, 1, 128, 32, 1, True, True),       # one column of tiles, accumulated into an existing gradient
    (481, 1, 512, 48, 2, True, False),       # odd row count, cout not a multiple of 32, two groups
    (3, 5, 64, 64, 1, False, False),         # a 1x1 conv over 5 x 5 images (rows are (image, pixel))
    (480, 1, 1024, 256, 1, True, False),     # 256 tiles: the most the kernel takes
])
def test_small_weight_gradient_kernel(n, hw, cin, cout, G, bias, acc):
    """wgrad_small_kernel (kernel id 4): weight and bias gradient of a small 1x1 problem in one launch, against float64."""
    from egorear_amd import hip
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(G * n, hw, hw, cin, generator=g) * 2 - 1)
    dy = (torch.rand(G * n, hw, hw, cout, generator=g) * 2 - 1)
    ws = torch.empty(1 << 22, device=DEV)
    dw0 = (torch.rand(G, cout, cin, generator=g) - 0.5) if acc else None
    db0 = (torch.rand(G, cout, generator=g) - 0.5) if acc else None
    dw = dw0.clone().to(DEV) if acc else None
    db = db0.clone().to(DEV) if (acc and bias) else None
    if G == 1 and acc:
        dw, db = dw[0], (db[0] if db is not None else None)
    dw, db = hip.conv2d_wgrad(hip.Img(x.to(DEV)), hip.Img(dy.to(DEV)), 1, 1, 1, 0, ws, want_bias=bias, dw=dw, db=db, accumulate=acc, groups=G, x6=False)
    assert hip.lib.egr_wgrad_last_kernel() == 4
    xr, dr = x.double().view(G, -1, cin), dy.double().view(G, -1, cout)
    ref_w = torch.einsum("gmo,gmk->gok", dr, xr) + (dw0.double() if acc else 0)
    ref_b = dr.sum(1) + (db0.double() if acc else 0)
    sc = float(ref_w.abs().max())
    assert float((dw.double().cpu().view(G, cout, cin) - ref_w).abs().max()) <= 2e-6 * sc
    if bias:
        assert float((db.double().cpu().view(G, cout) - ref_b).abs().max()) <= 2e-6 * float(ref_b.abs().max())


def test_small_weight_gradient_kernel_interleaved_groups():
    """The heads of one batch as groups (channel slices of the same rows: element strides gx / gy), as the lifting layers launch it."""
    from egorear_amd import hip
    g = torch.Generator().manual_seed(6)
    rows, heads, cf, dh = 960, 4, 128, 32
    x = (torch.rand(rows, heads * cf, generator=g) * 2 - 1).to(DEV)
    dy = (torch.rand(rows, heads * dh, generator=g) * 2 - 1).to(DEV)
    ws = torch.empty(1 << 22, device=DEV)
    dw = torch.empty(heads, dh, cf, device=DEV)
    xi = hip.Img(x[:, :cf].view(rows, 1, 1, cf))
    di = hip.Img(dy[:, :dh].view(rows, 1, 1, dh))
    hip.conv2d_wgrad(xi, di, 1, 1, 1, 0, ws, dw=dw, groups=heads, gx=cf, gy=dh, x6=False)
    assert hip.lib.egr_wgrad_last_kernel() == 4
    ref = torch.einsum("rhd,rhk->hdk", dy.double().cpu().view(rows, heads, dh), x.double().cpu().view(rows, heads, cf))
    assert float((dw.double().cpu() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


@pytest.mark.parametrize("rows,cols", [(2048, 4096), (100, 36), (64, 64), (130, 257)])
def test_tiled_transpose(rows, cols):
    """egr_transpose_f32 (the data-gradient operand of a very large Linear after an update) is an exact transpose, ragged sizes included."""
    from egorear_amd import hip_train as T
    src = torch.arange(rows * cols, dtype=torch.float32, device=DEV).reshape(rows, cols) * 0.5
    dst = torch.full((cols, rows), -1.0, device=DEV)
    T.transpose_into(dst, src)
    assert torch.equal(dst, src.t())


@pytest.mark.parametrize("c,n,h,w,groups,pool", [(64, 4, 32, 32, 2, (3, 2, 1)), (64, 3, 20, 28, 1, (3, 2, 1)), (128, 2, 16, 16, 2, (2, 2, 0))])
def test_stem_batchnorm_relu_maxpool_in_one_pass(c, n, h, w, groups, pool):
    """egr_bn_relu_maxpool_f32 / egr_bn_pool_backward_f32 (round 6; resnet.py:16-17 in training mode): BatchNorm(train) + ReLU +
    MaxPool2d without the normalised tensor y and without the max-pool's scattered gradient dy.  Against torch autograd in float64, and
    BIT-identical to the separate launches it replaces (scale-shift -> max-pool with slots; max-pool adjoint -> BatchNorm backward with
    the ReLU mask taken from y) - pooled output, slots, dx, dgamma, dbeta and the abs-max records."""
    from egorear_amd import hip_train as T
    from egorear_amd.hip import Img
    k, s, p = pool
    ws = T.bn_workspace(DEV)
    x = rnd(groups * n, c, h, w, seed=1) * 2 + 0.3
    x[0, :, 3:6, 3:6] = -5.0                                     # a window whose values are all negative: relu -> a tie at zero
    x[1, :, 4, 4] = x[1, :, 4, 5]                                # ... and a positive tie inside a window (first in scan order wins)
    gamma, beta = rnd(groups, c, seed=3) + 1.5, rnd(groups, c, seed=4)
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    dyp = rnd(groups * n, c, ho, wo, seed=7)
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yref = torch.cat([F.max_pool2d(F.relu(F.batch_norm(xr[g * n:(g + 1) * n], None, None, gr[g], br[g], True, 0.1, 1e-5)), k, s, p) for g in range(groups)])
    dx_ref, dg_ref, db_ref = torch.autograd.grad(yref, (xr, gr, br), dyp.double())
    xd, gd, bd, dyd = nhwc(x).to(DEV), gamma.to(DEV), beta.to(DEV), nhwc(dyp).to(DEV)

    def recs():
        return torch.zeros(64, dtype=torch.int32, device=DEV), torch.zeros(64, dtype=torch.int32, device=DEV)
    # --- the separate launches
    r_y, r_dx = recs()
    y_full, ctx_a = T.bn_train(xd, gd, bd, None, None, groups, ws, relu=True, amax_out=r_y, want_extremes=True)
    pooled_a, slot_a = T.maxpool_train(Img(y_full), k, s, p)
    dy_full = T.maxpool_bwd(dyd, slot_a, (h, w), k, s, p)
    dx_a, dg_a, db_a, _ = T.bn_backward(ctx_a, dy_full, y_full, ws, amax_dx=r_dx)
    # --- one pass
    q_y, q_dx = recs()
    pooled_b, ctx_b = T.bn_train(xd, gd, bd, None, None, groups, ws, relu=True, amax_out=q_y, want_extremes=True, pool=pool)
    dx_b, dg_b, db_b, _ = T.bn_backward(ctx_b, dyd, None, ws, amax_dx=q_dx)
    torch.cuda.synchronize()
    close(pooled_b.permute(0, 3, 1, 2), yref, what="pooled")
    close(dx_b.permute(0, 3, 1, 2), dx_ref, rel=1e-4, what="dx")
    close(dg_b, dg_ref, rel=1e-4, what="dgamma")
    close(db_b, db_ref, rel=1e-4, what="dbeta")
    assert torch.equal(pooled_a.t, pooled_b) and torch.equal(slot_a, ctx_b.slot)
    assert torch.equal(dx_a, dx_b) and torch.equal(dg_a, dg_b) and torch.equal(db_a, db_b)
    assert torch.equal(r_y, q_y) and torch.equal(r_dx, q_dx) and pooled_b._egr_amax is q_y and dx_b._egr_amax is q_dx
    # the pooled form has no residual branch and no un-rectified variant
    with pytest.raises(RuntimeError):
        T.bn_train(xd, gd, bd, None, None, groups, ws, relu=False, pool=pool)
    with pytest.raises(RuntimeError):
        T.bn_backward(ctx_b, dy_full, None, ws)              # a full-resolution gradient for a pooled context
