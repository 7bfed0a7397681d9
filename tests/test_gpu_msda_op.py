"""The reference's native-op boundary (mmcv MultiScaleDeformableAttnFunction, models/utils/deform_attn.py:155-162) on a
real MI355X: egr_msda_fwd_f32 / egr_msda_bwd_f32 through egorear_amd.msda against the oracle's restatement of the mmcv
algorithm (oracle.egorear_oracle.msda_levels, forward and autograd backward on the CPU), the analytic known-answer cases
of SURVEY.md §8c, and the reference-shaped MSDeformAttn module sequence around the op.
Tolerance: fp32, 2e-5 of the tensor's magnitude (the gradient w.r.t. value is accumulated with atomics)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = 2e-5


@pytest.fixture(scope="module")
def msda():
    from egorear_amd import hip, msda as m
    assert torch.cuda.is_available() and "gfx950" in hip.device_arch()
    return m


def make_case(n, heads, d, lq, shapes, points, seed=0, spread=1.4):
    g = torch.Generator().manual_seed(seed)
    shapes_t = torch.tensor(shapes, dtype=torch.int64)
    sizes = [h * w for h, w in shapes]
    starts = torch.tensor([sum(sizes[:i]) for i in range(len(sizes))], dtype=torch.int64)
    lin = sum(sizes)
    levels = len(shapes)
    value = torch.randn(n, lin, heads, d, generator=g)
    # locations reach past every border so the zero-padding and the "outside" branch are both exercised
    loc = torch.rand(n, lq, heads, levels, points, 2, generator=g) * spread - (spread - 1) / 2
    attn = torch.softmax(torch.randn(n, lq, heads, levels * points, generator=g), -1).view(n, lq, heads, levels, points)
    return value, shapes_t, starts, loc, attn


def close(got, ref, what, tol=TOL):
    ref = ref.double()
    got = got.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(float(ref.abs().max()), 1e-6) if ref.numel() else 1.0
    err = float((got - ref).abs().max()) if ref.numel() else 0.0
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


CASES = [
    # (n, heads, d, lq, shapes, points)                 what it pins
    (2, 4, 64, 15, [(64, 64)], 16),                   # heat-map refiner call (egoposeformer_heatmap_mvf_ex.py:767-796)
    (2, 4, 32, 16, [(64, 64)], 16),                   # pose3d decoder call (egoposeformer_mvf_ex.py:455-478)
    (3, 2, 8, 7, [(8, 8), (5, 7), (3, 2)], 4),        # three levels, 2-lane groups
    (1, 3, 6, 5, [(4, 9), (6, 2)], 3),                # d not a multiple of 4: scalar lanes
    (2, 1, 100, 4, [(7, 5)], 5),                      # 25 float4 per head -> 32-lane groups with idle lanes
    (1, 2, 130, 3, [(5, 5), (2, 3)], 2),              # d > 64 scalar lanes: the channel loop runs three times
    (1, 1, 320, 2, [(6, 6)], 2),                      # d > 256: float4 lanes loop twice
    (1, 1, 4, 1, [(1, 1)], 1),                        # degenerate 1x1 map
]


@pytest.mark.parametrize("case", CASES, ids=[f"n{c[0]}h{c[1]}d{c[2]}q{c[3]}L{len(c[4])}p{c[5]}" for c in CASES])
def test_forward_backward_match_oracle(msda, case):
    from oracle import egorear_oracle as O
    n, heads, d, lq, shapes, points = case
    value, shapes_t, starts, loc, attn = make_case(n, heads, d, lq, shapes, points, seed=hash(case[:4]) % 1000)
    v, l, a = value.clone().requires_grad_(), loc.clone().requires_grad_(), attn.clone().requires_grad_()
    ref = O.msda_levels(v, shapes_t, starts, l, a)
    go = torch.randn(ref.shape, generator=torch.Generator().manual_seed(5))
    rv, rl, ra = torch.autograd.grad(ref, [v, l, a], go)

    dv, dl, da = value.to(DEV).requires_grad_(), loc.to(DEV).requires_grad_(), attn.to(DEV).requires_grad_()
    out = msda.MultiScaleDeformableAttnFunction.apply(dv, shapes_t.to(DEV), starts.to(DEV), dl, da, 64)
    close(out, ref.detach(), "out")
    out.backward(go.to(DEV))
    close(dv.grad, rv, "grad_value")
    close(dl.grad, rl, "grad_sampling_locations")
    close(da.grad, ra, "grad_attention_weights")


def test_known_answers(msda):
    """a location on a pixel centre returns that value row; a location outside the map returns 0 and has
    zero gradients; a one-hot attention weight selects one sample."""
    h, w, heads, d = 5, 7, 2, 8
    value = torch.randn(1, h * w, heads, d, generator=torch.Generator().manual_seed(1))
    shapes = torch.tensor([[h, w]], dtype=torch.int64)
    starts = torch.zeros(1, dtype=torch.int64)
    ys, xs = [0, 2, 4, 3], [0, 3, 6, 1]
    lq = len(ys)
    loc = torch.zeros(1, lq, heads, 1, 2, 2)
    for q, (y, x) in enumerate(zip(ys, xs)):
        loc[0, q, :, 0, 0] = torch.tensor([(x + 0.5) / w, (y + 0.5) / h])   # pixel centre
        loc[0, q, :, 0, 1] = torch.tensor([1.0 + 1.5 / w, -1.5 / h])        # outside: pixel = (w + 1, -2)
    attn = torch.zeros(1, lq, heads, 1, 2)
    attn[..., 0] = 1.0                                                       # one-hot on the centre sample
    out = msda.msda_forward(value.to(DEV), shapes.to(DEV), starts.to(DEV), loc.to(DEV), attn.to(DEV)).cpu()
    for q, (y, x) in enumerate(zip(ys, xs)):
        want = value[0, y * w + x].reshape(-1)
        assert torch.allclose(out[0, q], want, atol=1e-6), q
    attn2 = torch.zeros_like(attn)
    attn2[..., 1] = 1.0                                                      # all weight on the outside sample
    out2 = msda.msda_forward(value.to(DEV), shapes.to(DEV), starts.to(DEV), loc.to(DEV), attn2.to(DEV))
    assert float(out2.abs().max()) == 0.0
    gv, gl, ga = msda.msda_backward(value.to(DEV), shapes.to(DEV), starts.to(DEV), loc.to(DEV), attn2.to(DEV), torch.ones_like(out2))
    assert float(gv.abs().max()) == 0.0 and float(gl[..., 1, :].abs().max()) == 0.0 and float(ga[..., 1].abs().max()) == 0.0
    # the weight gradient of the centre sample is the sum of its value row
    assert torch.allclose(ga[0, :, :, 0, 0].cpu(), torch.stack([value[0, y * w + x].sum(-1) for y, x in zip(ys, xs)]), atol=1e-5)


def test_border_band_zero_padding(msda):
    """-1 < pixel < 0 and size-1 < pixel < size: inside the sampling range, with two or three corners read as zero."""
    from oracle import egorear_oracle as O
    h, w = 4, 6
    value = torch.randn(1, h * w, 1, 4, generator=torch.Generator().manual_seed(2))
    shapes = torch.tensor([[h, w]], dtype=torch.int64)
    starts = torch.zeros(1, dtype=torch.int64)
    px = torch.tensor([[-0.75, -0.25], [w - 0.4, 1.3], [2.5, h - 0.2], [-0.999, h - 0.001], [w - 1.0, h - 1.0], [-1.0, 1.0], [float(w), 1.0]])
    loc = ((px + 0.5) / torch.tensor([w, h], dtype=torch.float32)).view(1, len(px), 1, 1, 1, 2)
    attn = torch.ones(1, len(px), 1, 1, 1)
    ref = O.msda_levels(value, shapes, starts, loc, attn)
    out = msda.msda_forward(value.to(DEV), shapes.to(DEV), starts.to(DEV), loc.to(DEV), attn.to(DEV))
    close(out, ref, "border band")


def test_empty_and_errors(msda):
    shapes = torch.tensor([[2, 2]], dtype=torch.int64, device=DEV)
    starts = torch.zeros(1, dtype=torch.int64, device=DEV)
    value = torch.randn(2, 4, 1, 4, device=DEV)
    out = msda.msda_forward(value, shapes, starts, torch.zeros(2, 0, 1, 1, 3, 2, device=DEV), torch.zeros(2, 0, 1, 1, 3, device=DEV))
    assert out.shape == (2, 0, 4)
    gv, gl, ga = msda.msda_backward(value, shapes, starts, torch.zeros(2, 0, 1, 1, 3, 2, device=DEV), torch.zeros(2, 0, 1, 1, 3, device=DEV), out)
    assert gv.shape == value.shape and float(gv.abs().max()) == 0.0 and gl.numel() == 0 and ga.numel() == 0
    with pytest.raises(ValueError):
        msda.msda_forward(value, shapes, starts, torch.zeros(2, 3, 2, 1, 3, 2, device=DEV), torch.zeros(2, 3, 2, 1, 3, device=DEV))
    with pytest.raises(ValueError):
        msda.msda_forward(value, shapes.int(), starts, torch.zeros(2, 3, 1, 1, 3, 2, device=DEV), torch.zeros(2, 3, 1, 1, 3, device=DEV))
    with pytest.raises(RuntimeError):  # no CPU path
        msda.msda_forward(value.cpu(), shapes.cpu(), starts.cpu(), torch.zeros(2, 3, 1, 1, 3, 2), torch.zeros(2, 3, 1, 1, 3))
    with pytest.raises(RuntimeError):  # mmcv: the batch must be a multiple of im2col_step
        msda.MultiScaleDeformableAttnFunction.apply(torch.randn(3, 4, 1, 4, device=DEV), shapes, starts, torch.zeros(3, 3, 1, 1, 3, 2, device=DEV),
                                                    torch.zeros(3, 3, 1, 1, 3, device=DEV), 2)
    # shapes that claim more tokens than value holds: skipped corners, no fault
    big = torch.tensor([[8, 8]], dtype=torch.int64, device=DEV)
    o = msda.msda_forward(value, big, starts, torch.full((2, 3, 1, 1, 3, 2), 0.9, device=DEV), torch.ones(2, 3, 1, 1, 3, device=DEV))
    assert float(o.abs().max()) == 0.0


def test_reference_module_sequence(msda):
    """The statement sequence of MSDeformAttn.forward (deform_attn.py:110-168) with this op in place of mmcv's, on the
    device, against the oracle's ms_deform_attn on the CPU: what a maintainer gets from install_mmcv_shim()."""
    from oracle import egorear_oracle as O
    msda.install_mmcv_shim()
    from mmcv.ops.multi_scale_deform_attn import MultiScaleDeformableAttnFunction as Fn
    assert Fn is msda.MultiScaleDeformableAttnFunction
    g = torch.Generator().manual_seed(3)
    n, lq, c, heads, pts, H, W = 2, 15, 256, 4, 16, 64, 64
    sd = {"a.value_proj.weight": torch.randn(c, c, generator=g) / 16, "a.value_proj.bias": torch.randn(c, generator=g) * 0.1,
          "a.sampling_offsets.weight": torch.randn(heads * pts * 2, c, generator=g) * 0.2, "a.sampling_offsets.bias": torch.randn(heads * pts * 2, generator=g) * 2,
          "a.attention_weights.weight": torch.randn(heads * pts, c, generator=g) * 0.1, "a.attention_weights.bias": torch.randn(heads * pts, generator=g),
          "a.output_proj.weight": torch.randn(c, c, generator=g) / 16, "a.output_proj.bias": torch.randn(c, generator=g) * 0.1}
    query = torch.randn(n, lq, c, generator=g)
    memory = torch.randn(n, H * W, c, generator=g)
    ref_pts = torch.rand(n, lq, 1, 2, generator=g)
    want = O.ms_deform_attn(sd, "a", query, ref_pts, memory, H, W, heads, pts)

    d = {k: v.to(DEV) for k, v in sd.items()}
    q, m, r = query.to(DEV), memory.to(DEV), ref_pts.to(DEV)
    shapes = torch.tensor([[H, W]], dtype=torch.int64, device=DEV)
    starts = torch.zeros(1, dtype=torch.int64, device=DEV)
    value = F.linear(m, d["a.value_proj.weight"], d["a.value_proj.bias"]).view(n, H * W, heads, c // heads)
    off = F.linear(q, d["a.sampling_offsets.weight"], d["a.sampling_offsets.bias"]).view(n, lq, heads, 1, pts, 2)
    aw = F.softmax(F.linear(q, d["a.attention_weights.weight"], d["a.attention_weights.bias"]).view(n, lq, heads, pts), -1).view(n, lq, heads, 1, pts)
    norm = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
    loc = r[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    out = Fn.apply(value.to(torch.float32), shapes, starts, loc, aw, 64)
    got = F.linear(out, d["a.output_proj.weight"], d["a.output_proj.bias"])
    close(got, want, "MSDeformAttn.forward", tol=1e-4)  # rocBLAS vs oneDNN linears around the op


@pytest.mark.parametrize("seed,scale", [(0, 1.0), (2, 0.35)])
def test_op_inside_the_module_sequence_reproduces_the_references_msda_output(msda, seed, scale):
    """Row a13 against the REFERENCE (not the oracle): tests/golden/mvfex_mid_s*.npz holds what every refiner's
    `transformer_layers[0].cross_attn` returned inside the real reference for each of the four views
    (DeformMultiViewAttn.forward, egoposeformer_heatmap_mvf_ex.py:779-796 -> MSDeformAttn.forward, deform_attn.py:90-168, before the
    validity mask of :909).  Here the same statement sequence runs around egr_msda_fwd_f32 - value_proj, offsets / softmax weights
    from the golden's own JQA query, reference points = the golden's 2-D anchors, the op, output_proj - on the memory tokens of the
    same synthetic frames (frame_feat_multi_view_proj + pos_embed of the features the HIP path produced, themselves pinned by the
    feat_init golden).  2e-5, the tolerance the oracle meets on the same vector."""
    import copy
    import os
    import numpy as np
    from egorear_amd import configs, synth
    from egorear_amd.estimator import EgoPoseFormerHeatmapMVFEX
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    mid = np.load(os.path.join(gdir, f"mvfex_mid_s{seed}.npz"))
    end = np.load(os.path.join(gdir, f"mvfex_s{seed}.npz"))
    net = EgoPoseFormerHeatmapMVFEX(**copy.deepcopy(configs.heatmap_mvfex_cfg())).eval()
    synth.load_synth(net, 42)
    net = net.to(DEV)
    B, V, J = 2, 4, 15
    with torch.no_grad():
        _, fts = net(synth.synth_images(B, V, seed=seed, scale=scale).to(DEV))
        feat = fts[0].contiguous()                                              # (B, V, 128, 64, 64) initial features of all four views
        anchors = torch.from_numpy(end["anchors_2d"]).to(DEV)                   # (B, V, 15, 2), bit-equal on the HIP path (test_gpu_pipeline)
        for name in ("front_left", "front_right", "back_left", "back_right"):
            r = getattr(net, "heatmap_refiner_" + name)
            ca = r.transformer_layers[0].cross_attn
            heads, points, C = ca.n_heads, ca.n_points, ca.d_model
            # memory tokens: HeatmapMVF.forward :689-693
            w = r.frame_feat_multi_view_proj.weight.view(C, -1)
            tok = feat.flatten(3).transpose(2, 3)                                # (B, V, 4096, 128)
            memory = tok @ w.t() + r.frame_feat_multi_view_proj.bias + r.frame_feat_multi_view_pos_embed
            query = torch.from_numpy(mid[name + "_query"]).to(DEV)              # (B, 15, 256) from the reference
            outs = []
            for v in range(V):
                # MSDeformAttn.forward, statement by statement (levels = 1)
                value = F.linear(memory[:, v], ca.value_proj.weight, ca.value_proj.bias).view(B, -1, heads, C // heads)
                off = F.linear(query, ca.sampling_offsets.weight, ca.sampling_offsets.bias).view(B, J, heads, 1, points, 2)
                aw = F.softmax(F.linear(query, ca.attention_weights.weight, ca.attention_weights.bias).view(B, J, heads, points), -1)
                aw = aw.view(B, J, heads, 1, points)
                normalizer = torch.stack([ca.spatial_shapes[..., 1], ca.spatial_shapes[..., 0]], -1)
                ref_pts = anchors[:, v].reshape(B, J, 1, 2)
                loc = ref_pts[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
                out = msda.MultiScaleDeformableAttnFunction.apply(value.float(), ca.spatial_shapes, ca.start_index, loc, aw, 64)
                outs.append(F.linear(out, ca.output_proj.weight, ca.output_proj.bias))
            got = torch.stack(outs)[..., ::4].cpu().numpy()
            np.testing.assert_allclose(got, mid[name + "_msda"], rtol=0, atol=2e-5, err_msg=name)


def test_module_sequence_compiles_as_one_graph_around_the_operator(msda):
    """SURVEY.md 8(b) / run.py:7-9: with only the op swapped, the reference-shaped MSDeformAttn statement sequence compiles with
    fullgraph=True (the op is a torch.library operator with a Meta kernel and a registered autograd formula, not a ctypes call Dynamo
    has to break at); forward and the three input gradients equal the eager run bit for bit (the same kernels run either way)."""
    torch.manual_seed(0)
    n, lq, heads, d, points, hw = 2, 15, 4, 64, 16, 64
    C = heads * d
    shapes = torch.tensor([[hw, hw]], dtype=torch.int64, device=DEV)
    starts = torch.tensor([0], dtype=torch.int64, device=DEV)
    lins = [torch.nn.Linear(C, C).to(DEV), torch.nn.Linear(C, heads * points * 2).to(DEV), torch.nn.Linear(C, heads * points).to(DEV),
            torch.nn.Linear(C, C).to(DEV)]

    def seq(query, ref_pts, tokens):
        value = lins[0](tokens).view(n, hw * hw, heads, d)
        off = lins[1](query).view(n, lq, heads, 1, points, 2)
        aw = F.softmax(lins[2](query).view(n, lq, heads, points), -1).view(n, lq, heads, 1, points)
        normalizer = torch.stack([shapes[..., 1], shapes[..., 0]], -1)
        loc = ref_pts[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]
        out = msda.MultiScaleDeformableAttnFunction.apply(value.to(dtype=torch.float32), shapes, starts, loc, aw, 64)
        return lins[3](out)

    query = torch.randn(n, lq, C, device=DEV, requires_grad=True)
    ref = torch.rand(n, lq, 1, 2, device=DEV)
    tokens = torch.randn(n, hw * hw, C, device=DEV, requires_grad=True)
    go = torch.randn(n, lq, C, device=DEV)
    out_e = seq(query, ref, tokens)
    ge = torch.autograd.grad(out_e, [query, tokens], go)
    import torch._dynamo as dynamo
    dynamo.reset()
    cseq = torch.compile(seq, fullgraph=True, backend="aot_eager")
    out_c = cseq(query, ref, tokens)
    gc = torch.autograd.grad(out_c, [query, tokens], go)
    assert torch.equal(out_e, out_c)
    # (the value gradient is accumulated with atomics: order-dependent in the last bits)
    close(gc[0], ge[0].cpu(), "dquery compiled vs eager", tol=1e-5)
    close(gc[1], ge[1].cpu(), "dtokens compiled vs eager", tol=1e-5)
