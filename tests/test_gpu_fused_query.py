"""The round-5 token-chain launches against the per-op launches they replace (`-m gpu`, through the C ABI):
egr_jqa_query_f32 (the refiners' JQA query: heatmap_mvf_ex.py:655-665), egr_pose_query_f32 (mlp_pred[2] -> fisheye reprojection ->
query_gen_mlp -> first offsets / logits: egoposeformer_mvf_ex.py:255-262, 340-348, 400-410) and the head-offset tail of
egr_joint_layer_f32 (heatmap_mvf_ex.py:707-711).  The reference's goldens run through them in test_gpu_pipeline.py (shipped
default); here the two paths are compared with each other, operand for operand, in both weight arithmetics."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _net(cam="syn"):
    from egorear_amd import configs, synth
    from egorear_amd.estimator import EgoPoseFormerMVFEX
    net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg("ego4view_" + cam))).eval()
    synth.load_synth(net, 42)
    return net.to(DEV)


def _run(net, img, ctm, fused):
    from egorear_amd import engine, hip
    saved = engine.FUSED_QUERY
    engine.FUSED_QUERY = fused
    engine.CAPTURE = cap = {}
    hip.PROFILE = []
    try:
        with torch.no_grad():
            preds, hms = net(img, ctm)
        torch.cuda.synchronize()
        names = [p[0] for p in hip.PROFILE]
    finally:
        engine.FUSED_QUERY = saved
        engine.CAPTURE = None
        hip.PROFILE = None
    aux = net.__dict__["_egr_last_aux"]
    return torch.stack(preds).cpu(), [h.cpu() for h in hms], {k: v.cpu() for k, v in cap.items()}, aux, names


@pytest.mark.parametrize("cam,batch", [("syn", 2), ("rw", 3), ("syn", 32)])
def test_fused_token_chains_equal_the_per_op_launches(cam, batch):
    from egorear_amd import synth
    net = _net(cam)
    img = synth.synth_images(batch, 4, seed=5).to(DEV)
    ctm = synth.synth_coord_trans_mat(batch).to(DEV) if cam == "rw" else None
    _run(net, img, ctm, True), _run(net, img, ctm, False)                     # (the first forwards also pack the weights)
    p1, h1, c1, a1, n1 = _run(net, img, ctm, True)
    p0, h0, c0, a0, n0 = _run(net, img, ctm, False)
    # launch census: 6 -> 1 (query), 3 -> tail (head offset), 6 -> 1 (lifting head)
    assert "egr_jqa_query_f32" in n1 and "egr_pose_query_f32" in n1
    for gone in ("egr_jqa_sum_f32", "egr_avgpool_nhwc_f32", "egr_tokens_to_nhwc_f32", "egr_fisheye_project_f32"):
        assert gone in n0 and gone not in n1
    assert len(n0) - len(n1) == 13
    # the query and everything behind it: the same arithmetic class in another summation order
    np.testing.assert_allclose(c1["query"].numpy(), c0["query"].numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(c1["post_norm"].numpy(), c0["post_norm"].numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(c1["head_sum"].numpy(), c0["head_sum"].numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(h1[1].numpy(), h0[1].numpy(), rtol=0, atol=5e-5)
    np.testing.assert_array_equal(h1[0].numpy(), h0[0].numpy())               # (the initial heat maps do not pass through them)
    np.testing.assert_allclose(p1.numpy(), p0.numpy(), rtol=0, atol=2e-4)     # cm
    for k in ("anchors_valid",):
        np.testing.assert_array_equal(a1["pose3d"][k].cpu().numpy(), a0["pose3d"][k].cpu().numpy())
    np.testing.assert_allclose(a1["pose3d"]["anchors_2d"].cpu().numpy(), a0["pose3d"]["anchors_2d"].cpu().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(a1["pose3d"]["anchors_3d_after"].cpu().numpy(), a0["pose3d"]["anchors_3d_after"].cpu().numpy(), rtol=0, atol=2e-4)


def test_fused_token_chains_in_the_fp32_weight_order(monkeypatch):
    """EGR_LAYER_H2=0: the same launches on the fp32 matrix cores (w_packed = 1)."""
    from egorear_amd import engine, synth
    monkeypatch.setattr(engine, "LAYER_H2", False)
    net = _net("syn")
    img = synth.synth_images(2, 4, seed=7).to(DEV)
    p1, h1, c1, _, n1 = _run(net, img, None, True)
    p0, h0, c0, _, _ = _run(net, img, None, False)
    assert "egr_jqa_query_f32" in n1
    np.testing.assert_allclose(c1["query"].numpy(), c0["query"].numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(h1[1].numpy(), h0[1].numpy(), rtol=0, atol=5e-5)
    np.testing.assert_allclose(p1.numpy(), p0.numpy(), rtol=0, atol=2e-4)


def test_head_offset_tail_is_the_three_launches():
    """The head offset alone: tokens -> 16 x 16 image -> Conv2d(15, 64, 1) + ReLU -> up x2, tail vs the three launches, and its
    abs-max record bounds the output."""
    from egorear_amd import engine, hip, synth
    net = _net("syn")
    he = net.heatmap_estimator
    with torch.no_grad():
        net(synth.synth_images(2, 4, seed=3).to(DEV))                          # packs the refiners
    st = engine._state(he, torch.device(DEV))
    P = st.get(he.refiners()[0], lambda: None)
    G, B, J, C, V = 4, 3, 15, 256, 4
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(G * B * J, C, generator=gen).to(DEV)
    g = torch.randn(G * B * J * V, 4 * 128, generator=gen).to(DEV)
    sigma = torch.rand(G * 4 * B * J * V, generator=gen).to(DEV)
    rowmask = (torch.rand(B * J * V, generator=gen) > 0.2).to(torch.uint8).to(DEV)
    e = torch.randn(G * B * J * V, C, generator=gen).to(DEV)
    post = {"g": P.post_norm[0], "b": P.post_norm[1]}
    rec = torch.zeros(64, dtype=torch.int32, device=DEV)
    head = {"w": P.head0_w, "b": P.head0_b, "amax": rec}
    _, _, xn1, _ = hip.joint_layer(x, g, e, sigma, rowmask, P.layer.fused, B, J, V, C, G, post=post, want_xn=True, head=head)
    _, _, xn0, _ = hip.joint_layer(x, g, e, sigma, rowmask, P.layer.fused, B, J, V, C, G, post=post, want_xn=True)
    torch.testing.assert_close(xn1, xn0, rtol=0, atol=0)
    tok = hip.tokens_to_nhwc(xn0, G * B, J, C)
    h0 = hip.linear_smallk(tok, J, 1, P.head0_w, P.head0_b, G * B * C, 64, J, hip.ACT_RELU, groups=G)
    ref = hip.upsample2x(hip.Img(h0.view(G * B, 16, 16, 64))).t
    out = head["out"]
    assert out.shape == ref.shape == (G * B, 32, 32, 64)
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=2e-6)   # (fma contraction differs between the two builds of the interpolation)
    bound = rec.view(torch.float32).max().item()
    assert bound >= out.abs().max().item() and bound == h0.abs().max().item()


@pytest.mark.parametrize("kind", ["refiner", "lifting"])
@pytest.mark.parametrize("B", [1, 3])
def test_split_once_layer_kernel_is_bit_identical_to_the_in_loop_split(kind, B):
    """joint_layer_p_kernel (A tiles split once into LDS planes, weight chunks requested ahead) against joint_layer_kernel<C, true>:
    same scales, same products in the same order - every output bit for bit, all tails."""
    from egorear_amd import engine, hip, synth
    net = _net("syn")
    with torch.no_grad():
        net(synth.synth_images(2, 4, seed=3).to(DEV))
    dev = torch.device(DEV)
    he = net.heatmap_estimator
    gen = torch.Generator().manual_seed(17)
    V = 4
    if kind == "refiner":
        P = engine._state(he, dev).get(he.refiners()[0], lambda: None)
        G, J, C, W = 4, 15, 256, P.layer.fused
        post = {"g": P.post_norm[0], "b": P.post_norm[1]}
        kw = lambda: dict(post=post, want_xn=True, head={"w": P.head0_w, "b": P.head0_b, "amax": torch.zeros(64, dtype=torch.int32, device=DEV)})
        e = torch.randn(G * B * J * V, C, generator=gen).to(DEV)
    else:
        p3 = net.pose3d_estimator
        P = engine._state(p3, dev).get(p3, lambda: None)
        G, J, C, W = 1, 16, 128, P.layers[0].fused
        a3 = torch.randn(B * J, 3, generator=gen).to(DEV)
        kw = lambda: dict(ol=P.layers[1].ol_plain, post={"g": P.post[0][0], "b": P.post[0][1]}, want_xn=True,
                          reg={"w0": P.reg_plain[0][0], "b0": P.reg_plain[0][1], "w2": P.reg_plain[0][2], "b2": P.reg_plain[0][3], "anchors": a3})
        e = None
    assert W["packed"] == 2
    x = torch.randn(G * B * J, C, generator=gen).to(DEV)
    g = (torch.randn(G * B * J * V, 4 * 128, generator=gen) * 3.0).to(DEV)
    sigma = torch.rand(G * 4 * B * J * V, generator=gen).to(DEV)
    rowmask = (torch.rand(B * J * V, generator=gen) > 0.2).to(torch.uint8).to(DEV)
    outs = []
    for planes in (1, 0):
        old = hip.lib.egr_layer_set_planes(planes)
        try:
            k = kw()
            r = hip.joint_layer(x, g, e, sigma, rowmask, W, B, J, V, C, G, **k)
            extra = [k["head"]["out"], k["head"]["amax"]] if "head" in k else []
            outs.append([t for t in r if t is not None] + extra)
        finally:
            hip.lib.egr_layer_set_planes(old)
    torch.cuda.synchronize()
    assert len(outs[0]) == len(outs[1]) >= 3
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert torch.isfinite(outs[0][0]).all()


@pytest.mark.parametrize("batch", [1, 4])
def test_forked_branches_of_a_captured_forward_give_the_same_bits(batch):
    """Small forwards captured as a hipGraph run the shortcut convs and the refiners' own-view projection stack as parallel branches
    (engine.State.fork): outputs equal the inline forward's bit for bit, replay after replay."""
    from egorear_amd import engine, synth
    from egorear_amd.runner import GraphedForward
    net = _net("syn")
    img = synth.synth_images(batch, 4, seed=9).to(DEV)
    with torch.no_grad():
        ref_p, ref_h = net(img)
        ref_p, ref_h = [t.clone() for t in ref_p], [t.clone() for t in ref_h]
    assert engine.FORK_MAX_IMAGES >= 4 * batch
    g = GraphedForward(net)
    for _ in range(3):
        p, h = g(img)
        torch.cuda.synchronize()
        for a, b in zip(list(p) + list(h), ref_p + ref_h):
            assert torch.equal(a, b)
    st = engine._state(net.heatmap_estimator, torch.device(DEV))
    assert getattr(st, "side", None) is not None          # small forward: the side branch exists (taken while capturing only)
    names = []
    real = engine._Fork.__enter__

    def spy(self):
        names.append(self.on)
        return real(self)
    engine._Fork.__enter__ = spy
    try:
        with torch.no_grad():
            net(img)                                       # eager: inline
        g2 = GraphedForward(net, warmup=1)
        g2(img)                                            # warm-up (inline) + capture (forked)
    finally:
        engine._Fork.__enter__ = real
    assert names and not names[0] and any(names)
