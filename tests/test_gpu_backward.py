"""Training-row bricks: backward kernels against PyTorch autograd (CPU, fp64).
These are building blocks; the end-to-end training step is not assembled yet."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def close(got, ref, rel=2e-5):
    ref = ref.double()
    tol = rel * max(float(ref.abs().max()), 1e-6)
    err = float((got.double().cpu() - ref).abs().max())
    assert err <= tol, f"max err {err:.3e} > tol {tol:.3e}"


def pack_w_dgrad(w):  # forward OIHW -> data-gradient operand: in/out channels swapped, then the usual K order
    from egorear_amd.engine import pack_conv_weight
    wt = pack_conv_weight(w.transpose(0, 1).contiguous())
    npad = (wt.shape[0] + 31) // 32 * 32
    out = torch.zeros(npad, wt.shape[1])
    out[:wt.shape[0]] = wt
    return out


DGRAD_CASES = [
    # n, h, w (forward input), cin, cout, k, stride
    (2, 16, 16, 64, 64, 3, 1),
    (2, 32, 32, 64, 128, 3, 2),
    (3, 16, 16, 64, 128, 1, 2),
    (2, 8, 8, 128, 32, 1, 1),
    (5, 64, 64, 32, 64, 3, 2),      # many rows: 128x128 / 64x64 tile paths with halo masks
    (2, 15, 17, 32, 64, 3, 2),      # odd sizes: the stride-2 parity-class split does not apply, generic gather path
    (2, 16, 16, 64, 64, 3, 1),      # (duplicate geometry on purpose: class mode must leave stride 1 untouched)
    (1, 8, 8, 512, 256, 3, 2),      # deep K (16 channel chunks x class taps)
]


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_conv_dgrad_matches_autograd(case):
    from egorear_amd import hip
    n, h, w, cin, cout, k, s = case
    pad = k // 2
    x = rnd(n, cin, h, w, seed=1).double().requires_grad_(True)
    wt = rnd(cout, cin, k, k, seed=2, scale=1.0 / math.sqrt(cin * k * k))
    y = F.conv2d(x, wt.double(), None, s, pad)
    dy = rnd(*y.shape, seed=3)
    (dx_ref,) = torch.autograd.grad(y, x, dy.double())
    dy_nhwc = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    out = hip.conv2d(hip.Img(dy_nhwc), pack_w_dgrad(wt).to(DEV), cin, k, k, s, pad, transposed_out_hw=(h, w))
    assert out.t.shape == (n, h, w, cin)
    close(out.t.permute(0, 3, 1, 2), dx_ref)
    # accumulate into an existing gradient (a tensor consumed by two ops): residual input, no activation
    g0 = rnd(n, h, w, cin, seed=4).to(DEV)
    out2 = hip.conv2d(hip.Img(dy_nhwc), pack_w_dgrad(wt).to(DEV), cin, k, k, s, pad, transposed_out_hw=(h, w),
                      res=hip.Img(g0), res_mode=hip.RES_BEFORE_ACT)
    close(out2.t.permute(0, 3, 1, 2), dx_ref + g0.cpu().permute(0, 3, 1, 2).double())


WGRAD_CASES = [
    # n, h, w, cin, cout, k, stride
    (2, 16, 16, 64, 64, 3, 1),       # cout 64 -> 64-wide tiles, 18 K-chunks (ragged last K tile)
    (2, 32, 32, 64, 128, 3, 2),
    (3, 16, 16, 64, 128, 1, 2),
    (2, 8, 8, 128, 32, 1, 1),
    (5, 64, 64, 32, 64, 3, 1),       # many rows: several splits x stages
    (4, 16, 16, 256, 256, 3, 1),     # 2 x 18 tiles
    (77, 1, 1, 256, 48, 1, 1),       # linear layer, ragged rows, cout not a multiple of 32
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad_matches_autograd(case):
    from egorear_amd import hip
    from egorear_amd.engine import unpack_conv_weight
    n, h, w, cin, cout, k, s = case
    pad = k // 2
    x = rnd(n, cin, h, w, seed=11)
    wt = rnd(cout, cin, k, k, seed=12, scale=0.1).double().requires_grad_(True)
    b = rnd(cout, seed=13).double().requires_grad_(True)
    y = F.conv2d(x.double(), wt, b, s, pad)
    dy = rnd(*y.shape, seed=14)
    dw_ref, db_ref = torch.autograd.grad(y, (wt, b), dy.double())
    ws = torch.empty(1 << 24, device=DEV)
    xi = hip.Img(x.permute(0, 2, 3, 1).contiguous().to(DEV))
    dyi = hip.Img(dy.permute(0, 2, 3, 1).contiguous().to(DEV))
    dw, db = hip.conv2d_wgrad(xi, dyi, k, k, s, pad, ws, want_bias=True)
    close(unpack_conv_weight(dw, cin, k, k), dw_ref, rel=3e-5)
    close(db, db_ref, rel=3e-5)
    # deterministic (fixed-order slab reduction), and accumulation into an existing gradient
    dw2, db2 = hip.conv2d_wgrad(xi, dyi, k, k, s, pad, ws, want_bias=True)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)
    hip.conv2d_wgrad(xi, dyi, k, k, s, pad, ws, want_bias=True, dw=dw2, db=db2, accumulate=True)
    close(dw2, 2 * dw.double().cpu(), rel=1e-6)
    close(db2, 2 * db.double().cpu(), rel=1e-6)


def test_conv_wgrad_grouped_launch():
    """Three same-shape layers with own inputs / gradients in one launch (the refiners' and encoders' grouped layers)."""
    from egorear_amd import hip
    from egorear_amd.engine import unpack_conv_weight
    G, n, h, cin, cout, k = 3, 2, 16, 64, 96, 3
    x = rnd(G * n, cin, h, h, seed=21)
    dy = rnd(G * n, cout, h, h, seed=22)
    ws = torch.empty(1 << 24, device=DEV)
    dw, db = hip.conv2d_wgrad(hip.Img(x.permute(0, 2, 3, 1).contiguous().to(DEV)), hip.Img(dy.permute(0, 2, 3, 1).contiguous().to(DEV)),
                              k, k, 1, 1, ws, want_bias=True, groups=G)
    assert dw.shape == (G, cout, k * k * cin) and db.shape == (G, cout)
    for g in range(G):
        wt = torch.zeros(cout, cin, k, k, dtype=torch.float64, requires_grad=True)
        b = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(x[g * n:(g + 1) * n].double(), wt, b, 1, 1)
        dw_ref, db_ref = torch.autograd.grad(y, (wt, b), dy[g * n:(g + 1) * n].double())
        close(unpack_conv_weight(dw[g], cin, k, k), dw_ref, rel=3e-5)
        close(db[g], db_ref, rel=3e-5)


def test_masked_data_gradient_fuses_the_relu_backward():
    """egr_conv2d_masked_f32: dx = (dgrad(dy) + prev) * [x > 0] in one launch (stride 1 and the stride-2 parity classes, grouped)."""
    from egorear_amd import hip
    for (G, n, h, cin, cout, k, s) in ((1, 2, 16, 64, 128, 3, 1), (2, 2, 32, 64, 64, 3, 2), (1, 3, 8, 128, 32, 1, 1)):
        pad = k // 2
        ho = (h + 2 * pad - k) // s + 1
        xs = rnd(G * n, h, h, cin, seed=31)                      # the forward activation (post-ReLU: about half non-positive)
        prev = rnd(G * n, h, h, cin, seed=32)
        dy = rnd(G * n, ho, ho, cout, seed=33)
        wts = [rnd(cout, cin, k, k, seed=34 + g, scale=1.0 / math.sqrt(cin * k * k)) for g in range(G)]
        wt = torch.stack([pack_w_dgrad(w) for w in wts]) if G > 1 else pack_w_dgrad(wts[0])
        out = hip.conv2d(hip.Img(dy.to(DEV)), wt.to(DEV), cin, k, k, s, pad, transposed_out_hw=(h, h), groups=G,
                         res=hip.Img(prev.to(DEV)), res_mode=hip.RES_BEFORE_ACT, mask=hip.Img(xs.to(DEV)))
        for g in range(G):
            xr = torch.zeros(n, cin, h, h, dtype=torch.float64, requires_grad=True)
            y = F.conv2d(xr, wts[g].double(), None, s, pad)
            (dx_ref,) = torch.autograd.grad(y, xr, dy[g * n:(g + 1) * n].permute(0, 3, 1, 2).double())
            ref = (dx_ref.permute(0, 2, 3, 1) + prev[g * n:(g + 1) * n].double()) * (xs[g * n:(g + 1) * n] > 0)
            close(out.t[g * n:(g + 1) * n], ref)
