"""Per-kernel parity on a real MI355X: every C-ABI entry point against a plain PyTorch
(CPU, fp64 where it matters) statement of the same op, including the analytic known-answer
cases of SURVEY.md §8c (argmax ties, MSDA at pixel centres / outside, F7 chain)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    from egorear_amd import hip as h
    assert torch.cuda.is_available()
    assert "gfx950" in h.device_arch()
    return h


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def pack_w(w):  # OIHW -> [cout_pad][(ci/32, kh, kw, ci%32)]
    from egorear_amd.engine import pack_conv_weight
    co = w.shape[0]
    w2 = pack_conv_weight(w)
    npad = (co + 31) // 32 * 32
    out = torch.zeros(npad, w2.shape[1])
    out[:co] = w2
    return out


def close(got, ref, rel=2e-5):
    ref = ref.double()
    tol = rel * max(float(ref.abs().max()), 1e-6)
    err = float((got.double().cpu() - ref).abs().max())
    assert err <= tol, f"max err {err:.3e} > tol {tol:.3e}"


CONV_CASES = [
    # n, h, w, cin, cout, k, stride, act, res_mode, bn
    (2, 16, 16, 64, 64, 3, 1, 1, 1, True),      # BasicBlock conv2 + identity + relu
    (2, 16, 16, 64, 128, 3, 2, 1, 0, True),     # strided 3x3
    (2, 16, 16, 64, 128, 1, 2, 0, 0, True),     # downsample 1x1 s2
    (3, 8, 8, 512, 128, 1, 1, 1, 0, False),     # lateral 1x1
    (1, 64, 64, 128, 256, 3, 2, 1, 0, False),   # big tile path (M = 1024 <= 4096 -> 64x64)
    (5, 64, 64, 32, 128, 3, 1, 1, 2, False),    # M = 20480 -> 128x128 tiles, res after act
    (5, 64, 64, 64, 64, 3, 1, 1, 0, True),      # M = 20480, N = 64 -> 256x64 tiles
    (5, 64, 64, 128, 192, 1, 1, 0, 0, False),   # N = 192 (three 64-wide tiles)
    (2, 32, 32, 128, 15, 1, 1, 0, 0, False),    # 15 output channels (padded to 32)
    (37, 1, 1, 256, 48, 1, 1, 2, 0, False),     # linear + GELU, ragged rows
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_matches_torch(hip, case):
    n, h, w, cin, cout, k, stride, act, res_mode, bn = case
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
    if act == 1:
        ref = F.relu(ref)
    elif act == 2:
        ref = F.gelu(ref)
    if res_mode == 2:
        ref = ref + res.permute(0, 3, 1, 2).double()
    out = hip.conv2d(hip.Img(x.to(DEV)), pack_w(wt).to(DEV), cout, k, k, stride, pad,
                     scale=scale.to(DEV) if scale is not None else None, shift=shift.to(DEV), act=act,
                     res=hip.Img(res.to(DEV)) if res is not None else None, res_mode=res_mode)
    close(out.t.permute(0, 3, 1, 2), ref)


def test_conv_channel_slices_and_nchw_output(hip):
    n, h, w, cin, cout = 4, 16, 16, 64, 32
    wide = rnd(n, h, w, 2 * cin, seed=7).to(DEV)
    wt = rnd(cout, cin, 1, 1, seed=8, scale=0.1)
    ref = F.conv2d(wide[..., cin:].permute(0, 3, 1, 2).cpu().double(), wt.double())
    cat = torch.zeros(n, h, w, 3 * cout, device=DEV)
    hip.conv2d(hip.Img(wide[..., cin:]), pack_w(wt).to(DEV), cout, 1, 1, 1, 0, out=hip.Img(cat[..., cout:2 * cout]))
    close(cat[..., cout:2 * cout].permute(0, 3, 1, 2), ref)
    assert float(cat[..., :cout].abs().max()) == 0 and float(cat[..., 2 * cout:].abs().max()) == 0
    # channel-major output with a (v,b) -> (b,v) image map: n = v*B + b, B = 2, V = 2
    out = torch.zeros(2, 2, cout, h, w, device=DEV)
    plane = cout * h * w
    hip.conv2d(hip.Img(wide[..., cin:]), pack_w(wt).to(DEV), cout, 1, 1, 1, 0, out_nchw=out, ymap=hip.NMap(2, 2 * plane, plane))
    got = out.permute(1, 0, 2, 3, 4).reshape(n, cout, h, w)
    close(got, ref)


def test_conv_split_k_linear_rowscale_rowmask(hip):
    m, k, n_out = 6, 4096, 96
    x = rnd(m, k, seed=11)
    wt = rnd(n_out, k, seed=12, scale=1 / 64)
    b = rnd(n_out, seed=13)
    rs = rnd(m, seed=14)
    mask = torch.tensor([1, 0, 1, 1, 0, 1], dtype=torch.uint8)
    ref = x.double() @ wt.double().t() + b.double()[None] * rs.double()[:, None]
    ref = ref * mask.double()[:, None]
    ws = torch.empty(1 << 22, device=DEV)
    wp = torch.zeros(96, k)
    wp[:n_out] = wt
    for split in (1, 0, 7):
        out = hip.conv2d(hip.Img(x.to(DEV).view(m, 1, 1, k)), wp.to(DEV), n_out, 1, 1, 1, 0, shift=b.to(DEV),
                         rowscale=rs.to(DEV), rowmask=mask.to(DEV), workspace=ws, split_k=split)
        close(out.t.view(m, n_out), ref)


@pytest.mark.parametrize("m,k,n_out,groups,extra", [
    (16, 128, 128, 1, "gelu"),            # the heads' MLPs at batch 1
    (37, 96, 45, 1, "rowscale_mask"),     # ragged rows and channels, K = 3 x 32
    (512, 512, 128, 2, "res_before_relu"),
    (2048, 128, 32, 1, "res_after"),      # N = 32: one column of tiles
    (8192, 64, 128, 1, "plain"),          # K = 64: any tile count
])
def test_small_linear_kernel(hip, m, k, n_out, groups, extra):
    """Small fp32 1x1 launches run on linear_small_kernel (kernel id 5: one 32 x 32 tile per workgroup, K split over its waves) and
    match the float64 product with the full epilogue; the same launch with an explicit split stays on the tiled kernel and agrees."""
    G = groups
    x = rnd(G * m, k, seed=31)
    npad = (n_out + 31) // 32 * 32
    wt = rnd(G, n_out, k, seed=32, scale=k ** -0.5)
    b = torch.zeros(G, npad)
    b[:, :n_out] = rnd(G, n_out, seed=33)
    wp = torch.zeros(G, npad, k)
    wp[:, :n_out] = wt
    ref = torch.einsum("gmk,gnk->gmn", x.view(G, m, k).double(), wt.double()).reshape(G * m, n_out)
    bias = b[:, :n_out].double().repeat_interleave(m, 0)
    kw = dict(shift=b.to(DEV) if G > 1 else b[0].to(DEV))
    if extra == "gelu":
        ref = F.gelu(ref + bias)
        kw.update(act=2)
    elif extra == "rowscale_mask":
        rs = rnd(G * m, seed=34)
        mask = (torch.arange(G * m) % 3 != 1).to(torch.uint8)
        ref = (ref + bias * rs.double()[:, None]) * mask.double()[:, None]
        kw.update(rowscale=rs.to(DEV), rowmask=mask.to(DEV))
    elif extra.startswith("res"):
        r = rnd(G * m, 1, 1, n_out, seed=35)
        rr = r.view(G * m, n_out).double()
        ref = F.relu(ref + bias + rr) if extra == "res_before_relu" else F.relu(ref + bias) + rr
        kw.update(act=1, res=hip.Img(r.to(DEV)), res_mode=1 if extra == "res_before_relu" else 2)
    else:
        ref = ref + bias
    ws = torch.empty(1 << 22, device=DEV)
    xi = hip.Img(x.to(DEV).view(G * m, 1, 1, k))
    wd = (wp if G > 1 else wp[0]).to(DEV)
    outs = []
    for split in (0, 2):
        y = hip.conv2d(xi, wd, n_out, 1, 1, 1, 0, workspace=ws, split_k=split, groups=G, **kw)
        assert hip.lib.egr_conv_last_kernel() == (5 if split == 0 else 0)
        close(y.t.view(G * m, n_out), ref, rel=2e-6)
        outs.append(y.t)
    assert float((outs[0] - outs[1]).abs().max()) <= 4e-6 * float(ref.abs().max())


def test_split_k_in_the_last_slice_on_two_streams_side_by_side(hip):
    """The arrival counters are one region per workspace: two streams running the fused split-K form at the same time (two engine
    lanes do, each with its own workspace) each get the result of a lone launch, bit for bit, over many overlapped launches."""
    n, h, cin, cout, k = 2, 16, 256, 128, 3
    xs = [rnd(n, h, h, cin, seed=320 + i).to(DEV) for i in range(2)]
    wp = pack_w(rnd(cout, cin, k, k, seed=322, scale=1.0 / math.sqrt(cin * 9))).to(DEV)
    wss = [torch.empty(1 << 22, device=DEV) for _ in range(2)]
    try:
        hip.lib.egr_conv_set_splitk_fused(1)
        lone = [hip.conv2d(hip.Img(xs[i]), wp, cout, k, k, 2, 1, workspace=wss[i], split_k=8).t.clone() for i in range(2)]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream() for _ in range(2)]
        outs = [[], []]
        for rep in range(40):
            for i, st in enumerate(streams):
                with torch.cuda.stream(st):
                    outs[i].append(hip.conv2d(hip.Img(xs[i]), wp, cout, k, k, 2, 1, workspace=wss[i], split_k=8).t)
        torch.cuda.synchronize()
    finally:
        hip.lib.egr_conv_set_splitk_fused(0)
    for i in range(2):
        for o in outs[i]:
            assert torch.equal(o, lone[i])


def test_split_k_in_the_last_slice_in_two_captured_graphs_replayed_side_by_side(hip):
    """What runner.PipelinedForward does with the knob on: both lanes' graphs are captured on torch's ONE capture stream and then
    replayed on two streams at the same time.  The counter region follows the workspace (one per lane), not the capture stream, so
    every replay gives the lone launch's bits."""
    n, h, cin, cout, k = 2, 16, 256, 128, 3
    xs = [rnd(n, h, h, cin, seed=330 + i).to(DEV) for i in range(2)]
    wp = pack_w(rnd(cout, cin, k, k, seed=332, scale=1.0 / math.sqrt(cin * 9))).to(DEV)
    wss = [torch.empty(1 << 22, device=DEV) for _ in range(2)]
    reps = 12
    try:
        hip.lib.egr_conv_set_splitk_fused(1)
        lone = [hip.conv2d(hip.Img(xs[i]), wp, cout, k, k, 2, 1, workspace=wss[i], split_k=8).t.clone() for i in range(2)]
        torch.cuda.synchronize()
        graphs, outs = [], []
        for i in range(2):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):           # no stream argument: the same capture stream for both, as GraphedForward captures
                o = [hip.conv2d(hip.Img(xs[i]), wp, cout, k, k, 2, 1, workspace=wss[i], split_k=8).t for _ in range(reps)]
            graphs.append(g)
            outs.append(o)
        streams = [torch.cuda.Stream() for _ in range(2)]
        for rnd_ in range(6):
            for o in outs:
                for t in o:
                    t.fill_(-7.0)
            torch.cuda.synchronize()
            for i, st in enumerate(streams):
                with torch.cuda.stream(st):
                    graphs[i].replay()
            torch.cuda.synchronize()
            for i in range(2):
                for t in outs[i]:
                    assert torch.equal(t, lone[i])
    finally:
        hip.lib.egr_conv_set_splitk_fused(0)


@pytest.mark.parametrize("case", [
    # n, h, cin, cout, k, stride, groups, split, extras
    (6, 1, 4096, 96, 1, 1, 1, 7, "rowscale_mask"),       # skinny linear, ragged rows and channels
    (4, 32, 64, 64, 3, 1, 2, 3, "res_before_relu"),      # grouped 3x3 (the small-batch layers that are split 2-4 ways)
    (2, 16, 256, 128, 3, 2, 1, 0, "nchw"),               # automatic split, channel-major output
    (64, 1, 8192, 256, 1, 1, 1, 16, "gelu"),             # 16 slices on four 64-column tiles
])
def test_split_k_reduction_in_the_last_slice_equals_the_second_pass(hip, case):
    """Opt-in form of split-K (egr_conv_set_splitk_fused): a tile is reduced in whichever K slice arrives last (arrival counter,
    agent-scope slab accesses) - summed in slice order like
    splitk_reduce_kernel, so the result does not depend on the arrival order: equal to the two-pass form within one rounding of
    the epilogue's multiply-add, identical from run to run, correct against fp64, and the counters are left at zero (second call)."""
    n, h, cin, cout, k, stride, G, split, extra = case
    pad = k // 2
    x = rnd(G * n, h, h, cin, seed=301)
    wts = [rnd(cout, cin, k, k, seed=302 + g, scale=1.0 / math.sqrt(cin * k * k)) for g in range(G)]
    wp = (torch.stack([pack_w(w) for w in wts]) if G > 1 else pack_w(wts[0])).to(DEV)
    npad = wp.shape[-2]
    ho = (h + 2 * pad - k) // stride + 1
    kw = dict(groups=G, workspace=torch.empty(1 << 23, device=DEV), split_k=split)
    sh = rnd(G, npad, seed=305)
    kw["shift"] = (sh if G > 1 else sh[0]).to(DEV)
    res = rs = mask = None
    if extra == "rowscale_mask":
        rs, mask = rnd(n, seed=306), torch.tensor([1, 0, 1, 1, 0, 1], dtype=torch.uint8)
        kw.update(rowscale=rs.to(DEV), rowmask=mask.to(DEV))
    if extra == "res_before_relu":
        res = rnd(G * n, ho, ho, cout, seed=307)
        kw.update(res=hip.Img(res.to(DEV)), res_mode=hip.RES_BEFORE_ACT, act=hip.ACT_RELU)
    if extra == "gelu":
        kw["act"] = hip.ACT_GELU

    def run():
        if extra == "nchw":
            planes = torch.full((n, cout, ho, ho), 7.0, device=DEV)
            hip.conv2d(hip.Img(x.to(DEV)), wp, cout, k, k, stride, pad, out_nchw=planes, ymap=hip.NMap(n, cout * ho * ho, 0), **kw)
            return planes
        return hip.conv2d(hip.Img(x.to(DEV)), wp, cout, k, k, stride, pad, **kw).t.permute(0, 3, 1, 2).clone()

    try:
        hip.lib.egr_conv_set_splitk_fused(1)
        a1, a2 = run(), run()
        hip.lib.egr_conv_set_splitk_fused(0)
        b = run()
    finally:
        hip.lib.egr_conv_set_splitk_fused(0)
    assert torch.equal(a1, a2)
    refs = []
    for g in range(G):
        r = F.conv2d(x[g * n:(g + 1) * n].permute(0, 3, 1, 2).double(), wts[g].double(), None, stride, pad)
        bias = (sh[g] if G > 1 else sh[0])[:cout].double().view(1, -1, 1, 1)
        r = r + (bias * rs.double().view(-1, 1, 1, 1) if rs is not None else bias)
        if extra == "res_before_relu":
            r = F.relu(r + res[g * n:(g + 1) * n].permute(0, 3, 1, 2).double())
        if extra == "gelu":
            r = F.gelu(r)
        if mask is not None:
            r = r * mask.double().view(-1, 1, 1, 1)
        refs.append(r)
    ref = torch.cat(refs)
    scale = float(ref.abs().max())
    assert float((a1.double().cpu() - ref).abs().max()) <= 2e-5 * scale
    assert float((a1 - b).abs().max()) <= 3e-7 * scale


def test_grouped_conv_equals_separate_launches(hip):
    """groups > 1: same-shape problems with their own weights / scale / shift / residual in one launch."""
    G, n, h, w, cin, cout = 3, 2, 16, 16, 64, 48
    x = rnd(G * n, h, w, cin, seed=201).to(DEV)
    res = rnd(G * n, h, w, cout, seed=202).to(DEV)
    wts = [rnd(cout, cin, 3, 3, seed=210 + g, scale=0.05) for g in range(G)]
    scs = [rnd(64, seed=220 + g) * 0.3 + 1.0 for g in range(G)]
    shs = [rnd(64, seed=230 + g) for g in range(G)]
    wp = torch.stack([pack_w(wt) for wt in wts]).to(DEV)
    out = hip.conv2d(hip.Img(x), wp, cout, 3, 3, 1, 1, scale=torch.stack(scs).to(DEV), shift=torch.stack(shs).to(DEV), act=1,
                     res=hip.Img(res), res_mode=1, groups=G)
    assert out.t.shape == (G * n, h, w, cout)
    for g in range(G):
        one = hip.conv2d(hip.Img(x[g * n:(g + 1) * n]), wp[g].contiguous(), cout, 3, 3, 1, 1, scale=scs[g].to(DEV),
                         shift=shs[g].to(DEV), act=1, res=hip.Img(res[g * n:(g + 1) * n]), res_mode=1)
        assert torch.equal(out.t[g * n:(g + 1) * n], one.t)           # bitwise: same kernel, same tiles
        ref = F.conv2d(x[g * n:(g + 1) * n].permute(0, 3, 1, 2).cpu().double(), wts[g].double(), None, 1, 1)
        ref = F.relu(ref * scs[g][:cout].double().view(1, -1, 1, 1) + shs[g][:cout].double().view(1, -1, 1, 1)
                     + res[g * n:(g + 1) * n].permute(0, 3, 1, 2).cpu().double())
        close(one.t.permute(0, 3, 1, 2), ref)


def test_grouped_small_kernels(hip):
    G, rows, c = 4, 30, 256
    x, r = rnd(G * rows, c, seed=241), rnd(G * rows, c, seed=242)
    gam, bet = rnd(G, c, seed=243) + 1.5, rnd(G, c, seed=244)
    y = hip.layernorm(x.to(DEV), gam.to(DEV), bet.to(DEV), res=r.to(DEV), groups=G).cpu()
    for g in range(G):
        close(y[g * rows:(g + 1) * rows], F.layer_norm((x + r)[g * rows:(g + 1) * rows].double(), (c,), gam[g].double(), bet[g].double(), 1e-5), rel=5e-6)
    xs, ws, bs = rnd(G * 20, 15, seed=245), rnd(G, 64, 15, seed=246), rnd(G, 64, seed=247)
    y = hip.linear_smallk(xs.to(DEV), 15, 1, ws.to(DEV), bs.to(DEV), G * 20, 64, 15, 1, groups=G).cpu()
    for g in range(G):
        close(y[g * 20:(g + 1) * 20], F.relu(xs[g * 20:(g + 1) * 20].double() @ ws[g].double().t() + bs[g].double()), rel=2e-6)
    hm_e, emb, bfb = rnd(G * 2 * 15, 256, seed=248), rnd(G, 15, 256, seed=249), rnd(G * 2, 256, seed=250)
    y = hip.jqa_sum(hm_e.to(DEV), emb.to(DEV), bfb.to(DEV), G * 2, 15, 256, groups=G).cpu().view(G, 2, 15, 256)
    assert torch.equal(y, (emb[:, None] + bfb.view(G, 2, 1, 256)) + hm_e.view(G, 2, 15, 256))


def test_grouped_gather_equals_separate(hip):
    B, V, J, heads, hgt, wid, cf, dh, G = 2, 4, 15, 4, 64, 64, 128, 64, 3
    feat = rnd(V, B, hgt * wid, cf, seed=261).to(DEV)
    pos = rnd(G, V, hgt * wid, heads * dh, seed=262).to(DEV)
    ol = torch.cat([rnd(G * B * J, heads * 32, seed=263, scale=10.0), rnd(G * B * J, heads * 16, seed=264)], 1).contiguous().to(DEV)
    anchors = ((rnd(B, V, J, 2, seed=265) + 1) / 2).to(DEV)
    valid = (rnd(B, V, J, seed=266) > -0.5).to(torch.uint8).to(DEV)
    g, e, s, m = hip.msda_gather(feat, pos, ol, anchors, valid, B, V, J, heads, dh, hgt, wid, groups=G)
    for k in range(G):
        g1, e1, s1, m1 = hip.msda_gather(feat, pos[k].contiguous(), ol[k * B * J:(k + 1) * B * J].contiguous(), anchors, valid,
                                         B, V, J, heads, dh, hgt, wid)
        assert torch.equal(g[k], g1[0]) and torch.equal(e[k], e1[0]) and torch.equal(s[k], s1[0]) and torch.equal(m, m1)


def test_conv_rejects_bad_arguments(hip):
    x = torch.zeros(1, 4, 4, 30, device=DEV)
    with pytest.raises(RuntimeError):
        hip.conv2d(hip.Img(x), torch.zeros(32, 30, device=DEV), 32, 1, 1, 1, 0)  # cin % 32 != 0
    with pytest.raises(RuntimeError):
        hip.conv2d(hip.Img(torch.zeros(1, 4, 4, 32)), torch.zeros(32, 32), 32, 1, 1, 1, 0)  # CPU tensors


def test_stem_matches_torch(hip):
    B, V = 2, 4
    img = rnd(B, V, 3, 64, 128, seed=21)
    wt = rnd(64, 3, 7, 7, seed=22, scale=0.1)
    scale, shift = rnd(64, seed=23) * 0.3 + 1.0, rnd(64, seed=24)
    wp = torch.zeros(64, 148)
    wp[:, :147] = wt.reshape(64, 147)
    y = hip.stem(img.to(DEV), 2, 2, wp.to(DEV), scale.to(DEV), shift.to(DEV))  # views 2,3 -> n = v*B + b
    ref = F.conv2d(img[:, 2:].permute(1, 0, 2, 3, 4).reshape(2 * B, 3, 64, 128).double(), wt.double(), None, 2, 3)
    ref = F.relu(ref * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    close(y.t.permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("B,H,W,G", [(2, 64, 128, 1), (1, 256, 256, 2), (3, 128, 64, 1)])
def test_stem_pool_is_stem_then_maxpool_bit_for_bit(hip, B, H, W, G):
    """egr_stem_conv7x7_pool_f32 (resnet.py:16-17 in one pass; tile seams through atomic max) == the two kernels, exactly;
    the output buffer starts dirty (the kernel zeroes what it combines atomically)."""
    V = 2 * G
    img = rnd(B, V, 3, H, W, seed=25).to(DEV)
    wp = torch.zeros(G, 64, 148)
    wp[:, :, :147] = rnd(G, 64, 147, seed=26, scale=0.1)
    scale, shift = (rnd(G, 64, seed=27) * 0.3 + 1.0).to(DEV), rnd(G, 64, seed=28).to(DEV)
    wp = wp.to(DEV)
    two = hip.maxpool(hip.stem(img, 0, 2, wp, scale, shift, groups=G), 3, 2, 1).t
    torch.empty_like(two).fill_(1e30)        # likely the block stem_pool's output gets
    one = hip.stem_pool(img, 0, 2, wp, scale, shift, groups=G).t
    assert one.shape == two.shape == (G * 2 * B, H // 4, W // 4, 64)
    assert torch.equal(one, two)
    again = hip.stem_pool(img, 0, 2, wp, scale, shift, groups=G).t
    assert torch.equal(again, one)
    with pytest.raises(RuntimeError):
        hip.stem_pool(img, 0, 2, wp, None, None, groups=G)


@pytest.mark.parametrize("B,H,W,G", [(2, 64, 128, 1), (1, 256, 256, 2), (3, 128, 64, 1)])
def test_stem_x6_matches_torch_and_pools_bit_for_bit(hip, B, H, W, G):
    """egr_stem_conv7x7_x6_f32 (split-bf16 operands): eval mode and raw mode vs fp64 at fp32-conv accuracy; pool=1 ==
    pool=0 + maxpool exactly."""
    V = 2 * G
    img = rnd(B, V, 3, H, W, seed=35)
    wt = rnd(G, 64, 3, 7, 7, seed=36, scale=0.1)
    wp = torch.zeros(G, 64, 148)
    wp[:, :, :147] = wt.reshape(G, 64, 147)
    scale, shift = rnd(G, 64, seed=37) * 0.3 + 1.0, rnd(G, 64, seed=38)
    w6 = hip.pack_stem_w6(wp.to(DEV))
    assert w6.shape == (G, 11 * 2 * 3 * 1024) and w6.dtype == torch.uint8
    y = hip.stem_x6(img.to(DEV), 0, 2, w6, scale.to(DEV), shift.to(DEV), groups=G)
    yraw = hip.stem_x6(img.to(DEV), 0, 2, w6, None, None, groups=G)
    for g in range(G):
        xin = img[:, 2 * g:2 * g + 2].permute(1, 0, 2, 3, 4).reshape(2 * B, 3, H, W).double()
        ref = F.conv2d(xin, wt[g].double(), None, 2, 3)
        close(yraw.t[g * 2 * B:(g + 1) * 2 * B].permute(0, 3, 1, 2), ref)
        close(y.t[g * 2 * B:(g + 1) * 2 * B].permute(0, 3, 1, 2), F.relu(ref * scale[g].double().view(1, -1, 1, 1) + shift[g].double().view(1, -1, 1, 1)))
    # same accuracy class as the fp32-MFMA stem
    y32 = hip.stem(img.to(DEV), 0, 2, wp.to(DEV), scale.to(DEV), shift.to(DEV), groups=G)
    assert float((y.t - y32.t).abs().max()) <= 2e-6 * float(y32.t.abs().max())
    two = hip.maxpool(y, 3, 2, 1).t
    torch.empty_like(two).fill_(1e30)
    one = hip.stem_x6(img.to(DEV), 0, 2, w6, scale.to(DEV), shift.to(DEV), groups=G, pool=True).t
    assert one.shape == (G * 2 * B, H // 4, W // 4, 64) and torch.equal(one, two)
    assert torch.equal(hip.stem_x6(img.to(DEV), 0, 2, w6, scale.to(DEV), shift.to(DEV), groups=G, pool=True).t, one)
    with pytest.raises(RuntimeError):
        hip.stem_x6(img.to(DEV), 0, 2, w6, None, None, groups=G, pool=True)


@pytest.mark.parametrize("B,H,W,G,amp", [(2, 64, 64, 1, 1.0), (1, 96, 128, 2, 1.0), (2, 64, 64, 1, 3e4), (2, 64, 64, 1, 1e-6)])
def test_stem_h2_matches_torch_and_pools_bit_for_bit(hip, B, H, W, G, amp):
    """egr_stem_conv7x7_h2_f32 (the fp16 scheme, per-tile pre-scale of the input patch): eval and raw mode vs fp64 at fp32-conv
    accuracy whatever the input's range (images far above fp16's largest number, and far below its smallest normal), with black
    tiles and one bright pixel in the batch; pool=1 == pool=0 + maxpool exactly; the abs-max record covers the pooled output."""
    V = 2 * G
    img = rnd(B, V, 3, H, W, seed=45) * amp
    img[0, 0, :, :32, :] = 0.0                   # black tiles: the pre-scale of an all-zero patch
    img[-1, -1, 1, 40, 17] = 900.0 * amp         # one outlier among small values: the low plane of its neighbours is still exact
    wt = rnd(G, 64, 3, 7, 7, seed=46, scale=0.1)
    wt[:, 5] *= 1e-4                              # channels of very different size: the per-channel scale
    wt[:, 9] *= 300.0
    wp = torch.zeros(G, 64, 148)
    wp[:, :, :147] = wt.reshape(G, 64, 147)
    scale, shift = rnd(G, 64, seed=47) * 0.3 + 1.0, rnd(G, 64, seed=48) * amp
    wh2, wds = hip.pack_stem_wh2(wp.to(DEV))
    assert wh2.shape == (G, 11 * 2 * 2 * 1024) and wh2.dtype == torch.uint8 and wds.shape == (G, 64)
    m = wt.reshape(G, 64, 147).abs().amax(-1) / wds.cpu()          # the scaled channel maxima sit in [2^14, 2^15)
    assert float(m.min()) >= 2.0 ** 14 * (1 - 1e-6) and float(m.max()) < 2.0 ** 15
    y = hip.stem_x6(img.to(DEV), 0, 2, wh2, scale.to(DEV), shift.to(DEV), groups=G, w_descale=wds)
    yraw = hip.stem_x6(img.to(DEV), 0, 2, wh2, None, None, groups=G, w_descale=wds)
    y6raw = hip.stem_x6(img.to(DEV), 0, 2, hip.pack_stem_w6(wp.to(DEV)), None, None, groups=G)     # the split-bf16 launch, same operands
    for g in range(G):
        xin = img[:, 2 * g:2 * g + 2].permute(1, 0, 2, 3, 4).reshape(2 * B, 3, H, W).double()
        ref = F.conv2d(xin, wt[g].double(), None, 2, 3)
        # judged per output channel (the channels differ by 1e6) against the sum of |terms|, the bound an fp32 convolution has
        mag = F.conv2d(xin.abs(), wt[g].double().abs(), None, 2, 3)
        got = yraw.t[g * 2 * B:(g + 1) * 2 * B].permute(0, 3, 1, 2).double().cpu()
        e_h2 = float(((got - ref).abs() / (mag + 1e-300)).max())
        e_x6 = float(((y6raw.t[g * 2 * B:(g + 1) * 2 * B].permute(0, 3, 1, 2).double().cpu() - ref).abs() / (mag + 1e-300)).max())
        # an fp32 chain of 147 terms is bounded by 147 * 2^-24 = 8.8e-6 of the sum of magnitudes; the split launches sit far inside it
        assert e_h2 <= 1e-6 and e_h2 <= 1.5 * e_x6 + 3e-7, (e_h2, e_x6)
        sg, hg = scale[g].double().view(1, -1, 1, 1), shift[g].double().view(1, -1, 1, 1)
        got = y.t[g * 2 * B:(g + 1) * 2 * B].permute(0, 3, 1, 2).double().cpu()
        assert float(((got - F.relu(ref * sg + hg)).abs() / (mag * sg.abs() + hg.abs() + 1e-300)).max()) <= 1.2e-6
    two = hip.maxpool(y, 3, 2, 1).t
    rec = torch.zeros(64, dtype=torch.int32, device=DEV)
    one = hip.stem_x6(img.to(DEV), 0, 2, wh2, scale.to(DEV), shift.to(DEV), groups=G, pool=True, w_descale=wds, amax_out=rec)
    assert one.t.shape == (G * 2 * B, H // 4, W // 4, 64) and torch.equal(one.t, two)
    assert float(rec.view(torch.float32).max()) == float(two.abs().max()) and one.amax is rec
    with pytest.raises(RuntimeError):
        hip.stem_x6(img.to(DEV), 0, 2, wh2[:, :-16], scale.to(DEV), shift.to(DEV), groups=G, w_descale=wds)
    with pytest.raises(RuntimeError):
        hip.stem_x6(img.to(DEV), 0, 2, wh2, scale.to(DEV), shift.to(DEV), groups=G, w_descale=wds[:, :32])


def test_pool_and_upsample_match_torch(hip):
    x = rnd(3, 16, 16, 64, seed=31)
    xc = x.permute(0, 3, 1, 2)
    y = hip.maxpool(hip.Img(x.to(DEV)), 3, 2, 1)
    assert torch.equal(y.t.cpu().permute(0, 3, 1, 2), F.max_pool2d(xc, 3, 2, 1))
    y = hip.maxpool(hip.Img(x.to(DEV)), 2, 2, 0)
    assert torch.equal(y.t.cpu().permute(0, 3, 1, 2), F.max_pool2d(xc, 2))
    up = hip.upsample2x(hip.Img(x.to(DEV)))
    ref = F.interpolate(xc, scale_factor=2, mode="bilinear", align_corners=True)
    close(up.t.permute(0, 3, 1, 2), ref, rel=2e-6)
    close(hip.upsample2x(hip.Img(x.to(DEV)), relu=True).t.permute(0, 3, 1, 2), F.relu(ref), rel=2e-6)
    # a ramp is reproduced exactly at the original grid points' images (align_corners=True)
    ramp = torch.arange(8.0).view(1, 1, 8, 1).expand(1, 8, 8, 4).contiguous()
    up = hip.upsample2x(hip.Img(ramp.to(DEV))).t.cpu()
    assert float(up[0, 0, 0, 0]) == 0.0 and float(up[0, 0, 15, 0]) == 7.0
    avg = hip.avgpool(hip.Img(x.to(DEV)))
    close(avg, xc.double().mean(dim=(2, 3)), rel=2e-6)


def test_argmax_known_answers(hip):
    hm = rnd(2, 3, 64, 64, seed=41) * 0.4
    hm[0, 0, 10, 20] = 2.0
    hm[0, 1, 5, 7] = 3.0
    hm[0, 1, 40, 1] = 3.0          # tie: first (smaller flat index) wins
    hm[0, 2] = 0.25                 # flat map: index 0, below threshold
    hm[1, 0, 63, 63] = 0.5          # exactly at threshold -> valid
    anchors, maxvals, valid, idx = hip.argmax_rows(hm.to(DEV), 0.5)
    ref_max, ref_idx = torch.max(hm.view(6, -1), dim=1)
    assert torch.equal(idx.cpu().long(), ref_idx)
    assert torch.equal(maxvals.cpu(), ref_max)
    assert idx.cpu().tolist()[:3] == [10 * 64 + 20, 5 * 64 + 7, 0]
    assert valid.cpu().tolist()[:4] == [1, 1, 0, 1]
    a = anchors.cpu()
    assert a[0].tolist() == [20 / 64, 10 / 64] and a[3].tolist() == [63 / 64, 63 / 64]


def test_pack_layer_w_layout_and_packed_weights_give_identical_results(hip):
    """egr_pack_layer_w_f32: [16-row block][128-deep chunk][16-deep block u][lane = 16 q + i][4] = w[16 block + i][128 chunk + 16 u + 4 q ..];
    egr_joint_layer_f32 with plain and with packed weight matrices: the same bits (only the address pattern of the loads differs)."""
    w = rnd(3, 48, 256, seed=91).to(DEV)
    pk = hip.pack_layer_w(w)
    ref = w.view(3, 3, 16, 2, 8, 4, 4).permute(0, 1, 3, 4, 5, 2, 6).contiguous().view(3, 48, 256)     # (m, nb, i, kc, u, q, t) -> (m, nb, kc, u, q, i, t)
    assert torch.equal(pk, ref)
    with pytest.raises(RuntimeError):
        hip.pack_layer_w(rnd(40, 128, seed=1).to(DEV))            # rows not a multiple of 16
    # one layer launch, C = 128, two query sets, every tail on
    G, B, J, V, C = 2, 3, 16, 4, 128
    r = lambda *s_, seed, scale=0.1: (rnd(*s_, seed=seed) * scale).to(DEV)      # noqa: E731
    W = {"w_fold": r(G, C, 128, seed=1), "c_fold": r(G, C, seed=2), "w_out": r(G, C, C, seed=3), "b_out": r(G, C, seed=4),
         "w_fuse": r(G, C, V * C, seed=5), "b_fuse": r(G, C, seed=6), "ln1_g": r(G, C, seed=7) + 1, "ln1_b": r(G, C, seed=8),
         "w_qkv": r(G, 3 * C, C, seed=9), "b_qkv": r(G, 3 * C, seed=10), "w_mo": r(G, C, C, seed=11), "b_mo": r(G, C, seed=12),
         "ln2_g": r(G, C, seed=13) + 1, "ln2_b": r(G, C, seed=14), "w_f0": r(G, 512, C, seed=15), "b_f0": r(G, 512, seed=16),
         "w_f1": r(G, C, 512, seed=17), "b_f1": r(G, C, seed=18), "ln3_g": r(G, C, seed=19) + 1, "ln3_b": r(G, C, seed=20)}
    ol = {"w": r(G, 192, C, seed=21), "b": r(G, 192, seed=22)}
    post = {"g": r(G, C, seed=23) + 1, "b": r(G, C, seed=24)}
    reg = {"w0": r(G, C, C, seed=25), "b0": r(G, C, seed=26), "w2": r(G, 3, C, seed=27), "b2": r(G, 3, seed=28), "anchors": r(G * B * J, 3, seed=29, scale=10.0)}
    x = r(G * B * J, C, seed=30, scale=1.0)
    g = r(G * B * J * V, 4, 128, seed=31, scale=1.0)
    sigma = (rnd(G, 4, B * J * V, seed=32).abs() * 0.5).to(DEV)
    rowmask = (torch.arange(B * J * V) % 7 != 0).to(torch.uint8).to(DEV)
    plain = hip.joint_layer(x, g, None, sigma, rowmask, W, B, J, V, C, G, ol=ol, post=post, reg=reg, want_xn=True)
    mats = ("w_fold", "w_out", "w_fuse", "w_qkv", "w_mo", "w_f0", "w_f1")
    Wp = {k: (hip.pack_layer_w(v) if k in mats else v) for k, v in W.items()}
    Wp["packed"] = True
    packed = hip.joint_layer(x, g, None, sigma, rowmask, Wp, B, J, V, C, G, ol={"w": hip.pack_layer_w(ol["w"]), "b": ol["b"]}, post=post,
                             reg=dict(reg, w0=hip.pack_layer_w(reg["w0"])), want_xn=True)
    for a, b in zip(plain, packed):
        assert a is not None and torch.equal(a, b)


def _layer_case(hip, C, G, B, J, xs=1.0, gs=1.0):
    V = 4
    r = lambda *s_, seed, scale=0.1: (rnd(*s_, seed=seed) * scale).to(DEV)      # noqa: E731
    W = {"w_fold": r(G, C, 128, seed=1), "c_fold": r(G, C, seed=2), "w_out": r(G, C, C, seed=3), "b_out": r(G, C, seed=4),
         "w_fuse": r(G, C, V * C, seed=5), "b_fuse": r(G, C, seed=6), "ln1_g": r(G, C, seed=7) + 1, "ln1_b": r(G, C, seed=8),
         "w_qkv": r(G, 3 * C, C, seed=9), "b_qkv": r(G, 3 * C, seed=10), "w_mo": r(G, C, C, seed=11), "b_mo": r(G, C, seed=12),
         "ln2_g": r(G, C, seed=13) + 1, "ln2_b": r(G, C, seed=14), "w_f0": r(G, 512, C, seed=15), "b_f0": r(G, 512, seed=16),
         "w_f1": r(G, C, 512, seed=17), "b_f1": r(G, C, seed=18), "ln3_g": r(G, C, seed=19) + 1, "ln3_b": r(G, C, seed=20)}
    W["w_out"][0, 5] *= 1e-5                 # a row of tiny weights and a row of large ones: per-row scales
    W["w_f0"][0, 9] *= 300.0
    ol = {"w": r(G, 192, C, seed=21), "b": r(G, 192, seed=22)}
    post = {"g": r(G, C, seed=23) + 1, "b": r(G, C, seed=24)}
    reg = {"w0": r(G, C, C, seed=25), "b0": r(G, C, seed=26), "w2": r(G, 3, C, seed=27), "b2": r(G, 3, seed=28), "anchors": r(G * B * J, 3, seed=29, scale=10.0)}
    x = r(G * B * J, C, seed=30, scale=xs)
    g = r(G * B * J * V, 4, 128, seed=31, scale=gs)
    sigma = (rnd(G, 4, B * J * V, seed=32).abs() * 0.5).to(DEV)
    rowmask = (torch.arange(B * J * V) % 7 != 0).to(torch.uint8).to(DEV)
    mats = ("w_fold", "w_out", "w_fuse", "w_qkv", "w_mo", "w_f0", "w_f1")
    outs = []
    for pk, flag in ((hip.pack_layer_w, True), (hip.pack_layer_wh2, 2)):
        Wp = {k: (pk(v) if k in mats else v) for k, v in W.items()}
        Wp["packed"] = flag
        outs.append(hip.joint_layer(x, g, None, sigma, rowmask, Wp, B, J, V, C, G, ol={"w": pk(ol["w"]), "b": ol["b"]}, post=post,
                                    reg=dict(reg, w0=pk(reg["w0"])), want_xn=True))
    return outs


@pytest.mark.parametrize("C,G,B,J", [(128, 2, 3, 16), (256, 4, 2, 15), (128, 1, 5, 15)])
def test_layer_in_the_fp16_scheme_matches_the_fp32_matrix_cores(hip, C, G, B, J):
    """egr_joint_layer_f32 with w_packed = 2 (egr_pack_layer_wh2_f32 images: two fp16 planes per weight, per-row power-of-two scales; the
    activation tiles scaled from their in-kernel abs-max and split on the fly; three products per fp32 product) against the same launch
    on the fp32 matrix cores: every output (tokens, next offsets / logits, post_norm, 3-D prediction) within a few 2^-22 of its scale -
    also with inputs far outside fp16's range."""
    # image layout: [16-row block][128-deep chunk][32-deep block][plane][lane = 16 q + i][8 fp16] * 2^e(row), descales behind
    w = rnd(2, 32, 256, seed=93).to(DEV)
    w[0, 3] *= 1e-4
    img = hip.pack_layer_wh2(w)
    assert img.shape == (2, 32 * 256 + 32)
    ds = img[:, 32 * 256:].double()
    m, _ = torch.frexp(ds)
    assert torch.all(m == 0.5)                                               # exact powers of two
    amax = w.abs().amax(-1).double()
    assert torch.all(amax / ds >= 2.0 ** 14) and torch.all(amax / ds < 2.0 ** 15)
    planes = img[:, :32 * 256].contiguous().view(torch.float16).view(2, 2, 2, 4, 2, 4, 16, 8).double()    # (m, nb, kc, kb, plane, q, i, t)
    back = (planes[:, :, :, :, 0] + planes[:, :, :, :, 1]).permute(0, 1, 5, 2, 3, 4, 6).reshape(2, 32, 256) * ds.view(2, 32, 1)
    assert float((back - w.double()).abs().max() / w.abs().max()) < 2.0 ** -21      # h + l carries 22 bits of each weight (relative to the row)
    assert float(((back - w.double()).abs() / amax.view(2, 32, 1)).max()) < 2.0 ** -21
    for xs, gs in ((1.0, 1.0), (3e4, 1e5), (1e-6, 1e-7)):
        f32, h2 = _layer_case(hip, C, G, B, J, xs, gs)
        if xs == 1.0:                           # the same bits from launch to launch (the abs-max slots are order-independent maxima)
            for rep in range(3):
                again = _layer_case(hip, C, G, B, J, xs, gs)[1]
                assert all(torch.equal(a, b) for a, b in zip(h2, again))
        for name, a, b in zip(("tokens", "offsets/logits", "post_norm", "pred"), f32, h2):
            assert a is not None and b is not None and torch.isfinite(b).all(), name
            scale = float(a.abs().max())
            err = float((a - b).abs().max())
            # (far from unit scale the residual + LayerNorm in front cancels leading digits: both arithmetics then sit ~1e-5 from fp64)
            assert err <= (4e-6 if xs == 1.0 else 2e-5) * scale, (name, xs, gs, err, scale)


def test_layernorm_and_mha_match_torch(hip):
    for c in (128, 256):
        x, r = rnd(45, c, seed=51), rnd(45, c, seed=52)
        g, b = rnd(c, seed=53) + 1.5, rnd(c, seed=54)
        y = hip.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), res=r.to(DEV))
        close(y, F.layer_norm((x + r).double(), (c,), g.double(), b.double(), 1e-5), rel=5e-6)
    for (B, J, heads, d) in ((3, 15, 4, 64), (2, 16, 4, 32)):
        C = heads * d
        qkv = rnd(B, J, 3 * C, seed=55)
        out = hip.joint_mha(qkv.to(DEV), B, J, heads, d, d ** -0.5)
        q, k, v = [t.reshape(B, J, heads, d).permute(0, 2, 1, 3).double() for t in qkv.split(C, dim=-1)]
        ref = ((q @ k.transpose(-1, -2)) * d ** -0.5).softmax(-1) @ v
        close(out.view(B, J, C), ref.permute(0, 2, 1, 3).reshape(B, J, C), rel=5e-6)


def test_smallk_jqa_tokens(hip):
    x, w, b = rnd(33, 4, seed=61), rnd(128, 4, seed=62), rnd(128, seed=63)
    y = hip.linear_smallk(x.to(DEV), 4, 1, w.to(DEV), b.to(DEV), 33, 128, 4, 1)
    close(y, F.relu(x.double() @ w.double().t() + b.double()), rel=2e-6)
    hm_e, emb, bfb = rnd(2 * 15, 256, seed=64), rnd(15, 256, seed=65), rnd(2, 256, seed=66)
    y = hip.jqa_sum(hm_e.to(DEV), emb.to(DEV), bfb.to(DEV), 2, 15, 256)
    assert torch.equal(y.cpu().view(2, 15, 256), (emb[None] + bfb[:, None]) + hm_e.view(2, 15, 256))
    t = rnd(2, 15, 256, seed=67)
    assert torch.equal(hip.tokens_to_nhwc(t.to(DEV), 2, 15, 256).cpu(), t.permute(0, 2, 1).contiguous())


def _msda_dense(feat, pos, Wv, bv, offs, logits, anchors, valid, heads, hgt, wid):
    """Dense statement: project all tokens, then mmcv-sample (oracle.msda_core), per view."""
    from oracle import egorear_oracle as O
    V, B, HW, cf = feat.shape
    J = anchors.shape[2]
    outs = []
    for v in range(V):
        mem = feat[v].double()
        if pos is not None:
            mem = mem + pos[v].double()[None]
        value = (mem @ Wv.double().t() + bv.double()).view(B, HW, heads, -1)
        aw = logits.double().view(B, J, heads, 16).softmax(-1)
        loc = anchors[:, v].double()[:, :, None, None, :] + offs.double().view(B, J, heads, 16, 2) / torch.tensor([wid, hgt]).double()
        outs.append(O.msda_core(value, hgt, wid, loc, aw))
    return torch.stack(outs, dim=2)  # (B, J, V, C)


@pytest.mark.parametrize("with_pos", [True, False])
def test_msda_sample_then_project_equals_dense(hip, with_pos):
    B, V, J, heads, hgt, wid, cf = 2, 4, 15, 4, 64, 64, 128
    C = 256 if with_pos else 128
    dh = C // heads
    feat = rnd(V, B, hgt * wid, cf, seed=71)
    # pre-projection f -> W_pre f + b_pre, positional table added after it, then W_v (see engine.pack_layer)
    Wpre, bpre = rnd(C, cf, seed=72, scale=0.1), rnd(C, seed=73, scale=0.1)
    pos = rnd(V, hgt * wid, C, seed=74, scale=0.5) if with_pos else None
    Wv, bv = rnd(C, C, seed=75, scale=0.08), rnd(C, seed=76, scale=0.1)
    offs = rnd(B * J, heads * 32, seed=77, scale=12.0)       # pixels; some samples leave the map
    logits = rnd(B * J, heads * 16, seed=78, scale=2.0)
    anchors = (rnd(B, V, J, 2, seed=79) + 1) / 2
    anchors[0, 0, 0] = torch.tensor([0.0, 0.0])               # corner anchor: many zero-padded corners
    anchors[0, 1, 1] = torch.tensor([63 / 64, 63 / 64])
    valid = (rnd(B, V, J, seed=80) > -0.6).to(torch.uint8)
    # dense reference: value = W_v (W_pre f + b_pre + pos) + b_v
    mem_feat = feat.double() @ Wpre.double().t() + bpre.double()
    ref = _msda_dense(mem_feat.float(), pos, Wv, bv, offs, logits, anchors, valid, heads, hgt, wid)
    # sample-then-project on the device
    ol = torch.cat([offs, logits], dim=1).contiguous()
    Wfold = (Wv.double() @ Wpre.double()).float()
    cfold = (Wv.double() @ bpre.double() + bv.double()).float()
    pos_proj = (pos.double() @ Wv.double().t()).float().contiguous() if with_pos else None
    g, e, sigma, rowmask = hip.msda_gather(feat.to(DEV), pos_proj.to(DEV) if with_pos else None, ol.to(DEV), anchors.to(DEV),
                                           valid.to(DEV), B, V, J, heads, dh, hgt, wid)
    g, e, sigma = g[0], (e[0] if e is not None else None), sigma[0]   # single query set (groups = 1)
    rows = B * J * V
    assert torch.equal(rowmask.cpu().view(B, J, V), valid.permute(0, 2, 1))
    a = torch.empty(rows, C, device=DEV)
    g2 = g.view(rows, heads * cf)
    for h in range(heads):
        hip.conv2d(hip.Img(g2[:, h * cf:(h + 1) * cf].view(rows, 1, 1, cf)), Wfold[h * dh:(h + 1) * dh].contiguous().to(DEV), dh,
                   1, 1, 1, 0, shift=cfold[h * dh:(h + 1) * dh].contiguous().to(DEV), rowscale=sigma[h],
                   res=hip.Img(e[:, h * dh:(h + 1) * dh].view(rows, 1, 1, dh)) if with_pos else None,
                   res_mode=2 if with_pos else 0, out=hip.Img(a[:, h * dh:(h + 1) * dh].view(rows, 1, 1, dh)))
    got = a.cpu().view(B, J, V, C)
    m = valid.permute(0, 2, 1).bool()[..., None]
    close(got * m, ref * m, rel=3e-5)


def test_msda_known_answers(hip):
    """Sampling exactly at a pixel centre returns that feature row; fully outside returns 0; sigma is the in-bounds mass."""
    B, V, J, heads, hgt, wid, cf = 1, 1, 2, 4, 8, 8, 128
    feat = rnd(V, B, hgt * wid, cf, seed=91)
    ol = torch.zeros(B * J, heads * 48)
    anchors = torch.zeros(B, V, J, 2)
    anchors[0, 0, 0] = torch.tensor([(3 + 0.5) / wid, (5 + 0.5) / hgt])   # centre of pixel (y=5, x=3)
    anchors[0, 0, 1] = torch.tensor([5.0, 5.0])                             # far outside
    valid = torch.ones(B, V, J, dtype=torch.uint8)
    g, e, sigma, rowmask = hip.msda_gather(feat.to(DEV), None, ol.to(DEV), anchors.to(DEV), valid.to(DEV), B, V, J, heads, 32,
                                           hgt, wid)
    g, sigma = g[0].cpu(), sigma[0]
    for h in range(heads):
        close(g[0, h], feat[0, 0, 5 * wid + 3], rel=2e-6)
        assert float(g[1, h].abs().max()) == 0.0
    assert torch.allclose(sigma.cpu()[:, 0], torch.ones(heads), atol=1e-6) and float(sigma.cpu()[:, 1].abs().max()) == 0.0


def test_fisheye_matches_oracle_and_f7_chain(hip, calib_dir):
    from egorear_amd.camera import FishEyeCameraCalibratedModel as Cam
    from oracle import egorear_oracle as O
    names = O.CAMERAS
    B, J = 3, 16
    pts = rnd(B, J, 3, seed=101) * torch.tensor([40.0, 40.0, 50.0]) + torch.tensor([0.0, 10.0, -30.0])
    for mode in ("ego4view_syn", "ego4view_rw"):
        cams = torch.from_numpy(np.stack([Cam(mode, calib_dir, n).packed() for n in names])).to(DEV)
        from egorear_amd import synth
        ctm = synth.synth_coord_trans_mat(B) if mode == "ego4view_rw" else None
        dpts = pts.clone().to(DEV)
        a2, valid, q4 = hip.fisheye_project(dpts, ctm.to(DEV) if ctm is not None else None, cams)
        opts = pts.clone()
        ra, rv = O.reproject_3d_to_2d(O.make_cameras(mode, calib_dir), opts, ctm)
        assert torch.equal(valid.cpu().bool(), rv)
        assert float((a2.cpu() - ra).abs().max()) < 2e-6
        assert torch.allclose(dpts.cpu(), opts, atol=1e-5)      # syn: (+12, 0, 0) chain; rw: untouched
        if mode == "ego4view_syn":
            assert torch.allclose(dpts.cpu() - pts, torch.tensor([12.0, 0.0, 0.0]).expand_as(pts), atol=1e-4)
        else:
            assert torch.equal(dpts.cpu(), pts)
        q = q4.cpu().view(B, J, 4)
        assert torch.allclose(q[..., 0], (torch.arange(1, J + 1) / J).expand(B, J))
        assert torch.equal(q[..., 1:], dpts.cpu())


@pytest.mark.parametrize("groups,cout,h,w", [(2, 15, 32, 32), (1, 16, 4, 16), (4, 3, 8, 32)])
def test_up2_relu_head_matches_torch(groups, cout, h, w):
    """Fused tail of a heat-map head: bilinear x2 (align_corners=True) + ReLU + 1x1 conv into channel-major planes, grouped,
    view-major images scattered into a (B, V, cout, 2h, 2w) tensor through the image map."""
    import torch.nn.functional as F
    from egorear_amd import hip
    from egorear_amd.hip import NMap
    B, cin = 3, 128
    V = groups
    g = torch.Generator().manual_seed(7)
    lo = torch.randn(V * B, h, w, cin, generator=g)
    wt = torch.randn(groups, cout, cin, generator=g) / 11.0
    bias = torch.randn(groups, cout, generator=g)
    planes = torch.zeros(B, V, cout, 2 * h, 2 * w, device=DEV)
    plane = cout * 4 * h * w
    hip.up2_relu_head(hip.Img(lo.to(DEV)), wt.to(DEV), bias.to(DEV), planes, NMap(B, V * plane, 0), plane, groups=groups)   # group g = view g
    for v in range(V):
        x = lo[v * B:(v + 1) * B].permute(0, 3, 1, 2).double()
        up = F.relu(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True))
        ref = F.conv2d(up, wt[v].double()[:, :, None, None], bias[v].double())
        err = float((planes[:, v].cpu().double() - ref).abs().max())
        assert err <= 2e-5 * max(1.0, float(ref.abs().max())), (v, err)


@pytest.mark.parametrize("groups,cout,B", [(2, 15, 48), (4, 15, 16), (1, 16, 64)])
def test_up2_relu_head_persistent_kernel_is_bit_identical_to_the_tile_kernel(groups, cout, B):
    """From 1024 tiles the launch takes the persistent kernel (next tile's source pixels requested under the current tile's
    arithmetic): same bits as the one-workgroup-per-tile kernel, also where a workgroup's tiles cross images and groups."""
    from egorear_amd import hip
    from egorear_amd.hip import NMap
    h = w = 32
    cin, V = 128, groups
    g = torch.Generator().manual_seed(23)
    lo = torch.randn(V * B, h, w, cin, generator=g).to(DEV)
    wt = (torch.randn(groups, cout, cin, generator=g) / 11.0).to(DEV)
    bias = torch.randn(groups, cout, generator=g).to(DEV)
    plane = cout * 4 * h * w
    outs = []
    for persist in (1, 0):
        old = hip.lib.egr_head_set_persist(persist)
        try:
            planes = torch.zeros(B, V, cout, 2 * h, 2 * w, device=DEV)
            hip.up2_relu_head(hip.Img(lo), wt, bias, planes, NMap(B, V * plane, 0), plane, groups=groups)
            outs.append(planes)
        finally:
            hip.lib.egr_head_set_persist(old)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and float(outs[0].abs().max()) > 0


@pytest.mark.parametrize("groups,k", [(1, 1), (2, 3)])
def test_conv_with_residual_upsampled_in_the_epilogue(groups, k):
    """EGR_RES_UP2_BEFORE_ACT: y = relu(conv(x) + bias + up2_bilinear_ac(lo)) with lo at half resolution (the FPN top-down add)."""
    import torch.nn.functional as F
    from egorear_amd import hip
    from egorear_amd.engine import pack_conv_weight
    n, h, w, cin, cout = 2, 16, 32, 64, 128
    g = torch.Generator().manual_seed(11)
    x = torch.randn(groups * n, h, w, cin, generator=g)
    lo = torch.randn(groups * n, h // 2, w // 2, cout, generator=g)
    wt = torch.randn(groups, cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(groups, cout, generator=g)
    wp = torch.stack([pack_conv_weight(wt[i]) for i in range(groups)])
    y = hip.conv2d(hip.Img(x.to(DEV)), (wp if groups > 1 else wp[0]).contiguous().to(DEV), cout, k, k, 1, k // 2,
                   shift=(bias if groups > 1 else bias[0]).contiguous().to(DEV), act=hip.ACT_RELU, res=hip.Img(lo.to(DEV)),
                   res_mode=hip.RES_UP2_BEFORE_ACT, groups=groups)
    for i in range(groups):
        xs = x[i * n:(i + 1) * n].permute(0, 3, 1, 2).double()
        up = F.interpolate(lo[i * n:(i + 1) * n].permute(0, 3, 1, 2).double(), scale_factor=2, mode="bilinear", align_corners=True)
        ref = F.relu(F.conv2d(xs, wt[i].double(), bias[i].double(), 1, k // 2) + up)
        err = float((y.t[i * n:(i + 1) * n].permute(0, 3, 1, 2).cpu().double() - ref).abs().max())
        assert err <= 3e-5 * max(1.0, float(ref.abs().max())), (i, err)
