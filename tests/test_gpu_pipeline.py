"""End-to-end parity of the HIP path (through the drop-in estimator classes and the C ABI)
on a real MI355X: against the committed golden vectors of the REAL reference and against the
CPU oracle on the same seeded inputs.  Bar (BASELINE.json north_star): bit-exact 2-D argmax joint
indices, 3-D joint coordinates within 1e-3 cm."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

TOL_POSE_CM = 1e-3
TOL_HM = 1e-4  # heatmap values are O(1); fp32 conv stacks in a different summation order


def _golden_check(g, name, t, step_hw=8, step_c=1, tol=TOL_HM):
    t = t.float().cpu()
    np.testing.assert_allclose(t[:, :, ::step_c, ::step_hw, ::step_hw].numpy(), g[name + "_sl"], rtol=0, atol=tol)
    assert abs(t.double().sum().item() - float(g[name + "_sum"])) <= tol * t.numel()


def _build(cls, cfg):
    from egorear_amd import synth
    net = cls(**copy.deepcopy(cfg)).eval()
    synth.load_synth(net, 42)
    return net.to(DEV)


@pytest.fixture(scope="module")
def nets():
    from egorear_amd import configs
    from egorear_amd.estimator import EgoPoseFormerHeatmap, EgoPoseFormerHeatmapMVFEX, EgoPoseFormerMVFEX
    cache = {}

    def get(name):
        if name not in cache:
            if name == "heatmap":
                cache[name] = _build(EgoPoseFormerHeatmap, configs.heatmap_cfg())
            elif name == "mvfex":
                cache[name] = _build(EgoPoseFormerHeatmapMVFEX, configs.heatmap_mvfex_cfg())
            else:
                cache[name] = _build(EgoPoseFormerMVFEX, configs.pose3d_cfg("ego4view_" + name))
        return cache[name]
    return get


@pytest.fixture(params=["by-size", "split-everywhere", "bf16-everywhere"])
def launch_policy(request):
    """The golden comparisons run three times: with the shipped rule (at batch 2 most launches are below the size thresholds and
    stay on the fp32 matrix cores), with every eligible implicit-GEMM launch forced onto the split kernels - the fp16 scheme
    wherever the input carries an abs-max record, i.e. the shipped choice at the benchmarked batch sizes - and the same with the
    fp16 scheme switched off (split-bf16 everywhere), so that the reference's golden vectors pin both kernels end to end."""
    from egorear_amd import hip
    saved = (hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.H2, hip.CHAIN_MIN_ROWS, hip.CHAIN_BIG_MIN_ROWS)
    if request.param != "by-size":
        hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = 0, 0.0
        hip.CHAIN_MIN_ROWS = hip.CHAIN_BIG_MIN_ROWS = 0       # ... and every chained pair as one launch (also the heads' streamed 256 -> 256 -> 128)
    if request.param == "bf16-everywhere":
        hip.H2 = False
    yield request.param
    hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.H2, hip.CHAIN_MIN_ROWS, hip.CHAIN_BIG_MIN_ROWS = saved


def test_forced_policy_really_takes_the_fp16_scheme(nets):
    """Under 'split-everywhere' the forward's convolutions run the fp16 scheme (all but the few whose input has no record)."""
    from egorear_amd import hip, synth
    net = nets("syn")
    saved = (hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.PROFILE, hip.CHAIN_MIN_ROWS, hip.CHAIN_BIG_MIN_ROWS)
    hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.PROFILE = 0, 0.0, []
    hip.CHAIN_MIN_ROWS = hip.CHAIN_BIG_MIN_ROWS = 0
    try:
        with torch.no_grad():
            net(synth.synth_images(2, 4, seed=0).to(DEV))
        tags = [t for name, *_, t in hip.PROFILE if name in ("egr_conv2d_nhwc_f32", "egr_conv1x1_chain_f32")]
    finally:
        hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS, hip.PROFILE, hip.CHAIN_MIN_ROWS, hip.CHAIN_BIG_MIN_ROWS = saved
    h2 = sum(1 for t in tags if t.startswith("h2 "))
    x6 = sum(1 for t in tags if t.startswith("x6 "))
    chains = [t for t in tags if " chain " in t]
    # (a chained pair is one fp16-scheme launch: three resident chains + the two streamed 256 -> 256 -> 128 of the heads)
    assert h2 + len(chains) >= 45 and x6 <= 12 and len(chains) == 5 and sum("mid256" in t for t in chains) == 2, (h2, x6, len(tags), chains)


def test_library_is_the_native_one():
    from egorear_amd import hip
    assert os.path.basename(hip.LIB_PATH) == "libegorear_hip.so" and os.path.exists(hip.LIB_PATH)
    assert "gfx950" in hip.device_arch()


@pytest.mark.parametrize("seed", [0, 1])
def test_heatmap_vs_reference_golden(seed, nets, golden_dir, launch_policy):
    from egorear_amd import synth
    g = np.load(os.path.join(golden_dir, f"heatmap_s{seed}.npz"))
    net = nets("heatmap")
    with torch.no_grad():
        hm, feats, pyr = net(synth.synth_images(2, 2, seed=seed).to(DEV), return_feat=True)
    assert hm.shape == (2, 2, 15, 64, 64) and feats.shape == (2, 2, 128, 64, 64) and pyr[-1].shape == (2, 2, 512, 8, 8)
    _golden_check(g, "hm", hm)
    _golden_check(g, "feat", feats, 16, 16, tol=5e-4)
    _golden_check(g, "s32", pyr[-1], 4, 64, tol=5e-4)


@pytest.mark.parametrize("seed,scale", [(0, 1.0), (1, 1.0), (2, 0.35)])
def test_mvfex_vs_reference_golden(seed, scale, nets, golden_dir, launch_policy):
    from egorear_amd import synth
    g = np.load(os.path.join(golden_dir, f"mvfex_s{seed}.npz"))
    net = nets("mvfex")
    with torch.no_grad():
        hms, fts = net(synth.synth_images(2, 4, seed=seed, scale=scale).to(DEV))
    aux = net.__dict__["_egr_last_aux"]
    np.testing.assert_array_equal(aux["argmax_idx"].cpu().numpy(), g["argmax_idx"])           # bit-exact
    np.testing.assert_array_equal(aux["anchors_valid"].cpu().numpy().astype(bool), g["anchors_valid"])
    np.testing.assert_array_equal(aux["anchors_2d"].cpu().numpy(), g["anchors_2d"])
    np.testing.assert_allclose(aux["maxvals"].cpu().numpy(), g["maxvals"], rtol=0, atol=TOL_HM)
    _golden_check(g, "hm_init", hms[0])
    _golden_check(g, "hm_refined", hms[1])
    _golden_check(g, "feat_init", fts[0], 16, 16, tol=5e-4)
    _golden_check(g, "feat_refined", fts[1], 16, 16, tol=5e-4)
    # get_anchors_2d_from_hm keeps the reference's return contract
    pts, maxvals, mask = net.get_anchors_2d_from_hm(hms[0])
    assert pts.shape == (2, 4, 15, 2) and mask.dtype == torch.bool


@pytest.mark.parametrize("seed,scale", [(0, 1.0), (2, 0.35)])
def test_mvfex_intermediates_vs_reference_golden(seed, scale, nets, golden_dir, launch_policy):
    """SURVEY.md 8c intermediate pins, captured inside the reference's refiners by oracle/make_golden_mid.py: the JQA query behind
    fc_query (a9: egr_jqa_sum_f32 + the small linears), transformer layer + post_norm (a10 / a12-a15: egr_msda_gather_f32 with the
    folded value path + egr_joint_layer_f32), head offset + own-view projection (a16 + a11: egr_tokens_to_nhwc_f32, the head convs,
    frame_feat_proj_layers).  (The per-view MSDA output a13 never exists on the HIP path - sample-then-project, DESIGN.md 4: the
    operand-for-operand op is pinned in test_gpu_msda_op.py, its restatement against this golden in test_oracle_golden.py.)"""
    from egorear_amd import engine, synth
    g = np.load(os.path.join(golden_dir, f"mvfex_mid_s{seed}.npz"))
    net = nets("mvfex")
    engine.CAPTURE = cap = {}
    try:
        with torch.no_grad():
            net(synth.synth_images(2, 4, seed=seed, scale=scale).to(DEV))
    finally:
        engine.CAPTURE = None
    for gi, name in enumerate(("front_left", "front_right", "back_left", "back_right")):
        np.testing.assert_allclose(cap["query"][gi].cpu().numpy(), g[name + "_query"], rtol=0, atol=5e-5)
        np.testing.assert_allclose(cap["post_norm"][gi].cpu().numpy(), g[name + "_post_norm"], rtol=0, atol=2e-4)
        hs = cap["head_sum"].view(4, 2, *cap["head_sum"].shape[1:])[gi].permute(0, 3, 1, 2).float().cpu()      # (B, C, h, w)
        np.testing.assert_allclose(hs[:, ::8, ::4, ::4].numpy(), g[name + "_head_sum_sl"], rtol=0, atol=2e-4)
        assert abs(hs.double().sum().item() - float(g[name + "_head_sum_sum"])) <= 2e-4 * hs.numel()


@pytest.mark.parametrize("cam,seed", [("syn", 0), ("syn", 1), ("rw", 0)])
def test_pose3d_vs_reference_golden(cam, seed, nets, golden_dir, launch_policy):
    from egorear_amd import synth
    from oracle import egorear_oracle as O
    g = np.load(os.path.join(golden_dir, f"pose3d_{cam}_s{seed}.npz"))
    net = nets(cam)
    ctm = synth.synth_coord_trans_mat(2).to(DEV) if cam == "rw" else None
    with torch.no_grad():
        preds, hms = net(synth.synth_images(2, 4, seed=seed).to(DEV), ctm)
    assert len(preds) == 4 and len(hms) == 2 and preds[0].shape == (2, 16, 3)
    pred = torch.stack(preds).cpu().numpy()
    np.testing.assert_allclose(pred, g["pred_pose"], rtol=0, atol=TOL_POSE_CM)
    aux = net.__dict__["_egr_last_aux"]["pose3d"]
    np.testing.assert_array_equal(aux["anchors_valid"].cpu().numpy().astype(bool), g["anchors_valid"])
    np.testing.assert_allclose(aux["anchors_2d"].cpu().numpy(), g["anchors_2d"], rtol=0, atol=2e-5)
    _golden_check(g, "hm_init", hms[0])
    _golden_check(g, "hm_refined", hms[1])
    mp = (O.compute_mpjpe_batch(preds[-1].cpu(), synth.synth_gt_pose(2)) * 10.0).numpy()
    np.testing.assert_allclose(mp, g["mpjpe_mm"], rtol=0, atol=1e-2)   # MPJPE identical within 1e-3 cm = 1e-2 mm


def test_rw_accepts_float64_coord_trans_mat(nets):
    """The reference dataset yields float64 matrices and the reference then raises; the boundary casts."""
    from egorear_amd import synth
    net = nets("rw")
    img = synth.synth_images(1, 4, seed=3).to(DEV)
    ctm = synth.synth_coord_trans_mat(1)
    with torch.no_grad():
        a, _ = net(img, ctm.to(DEV))
        b, _ = net(img, ctm.double().to(DEV))
    assert torch.equal(a[-1], b[-1])
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            net(img, None)


def test_pipeline_vs_oracle_other_batch_and_determinism(nets, calib_dir):
    """Batch 3 (not a multiple of anything), fresh seed: HIP vs CPU oracle; and run-to-run bitwise determinism."""
    from egorear_amd import synth
    from oracle import egorear_oracle as O
    net = nets("syn")
    img = synth.synth_images(3, 4, seed=11)
    with torch.no_grad():
        preds, hms = net(img.to(DEV))
        aux = net.__dict__["_egr_last_aux"]
        idx = aux["heatmap"]["argmax_idx"].cpu()
        preds2, hms2 = net(img.to(DEV))
        sd = {k: v.cpu() for k, v in net.state_dict().items()}
        o_preds, o_hms, o_aux = O.mvfex_forward(sd, O.make_cameras("ego4view_syn", calib_dir), img)
    assert all(torch.equal(a, b) for a, b in zip(preds, preds2)) and all(torch.equal(a, b) for a, b in zip(hms, hms2))
    assert torch.equal(idx.long(), o_aux["heatmap"]["argmax_idx"])
    for p, q in zip(preds, o_preds):
        assert float((p.cpu() - q).abs().max()) < TOL_POSE_CM
    for p, q in zip(hms, o_hms):
        assert float((p.cpu() - q).abs().max()) < TOL_HM


def test_no_cpu_fallback_and_training_is_refused(nets):
    from egorear_amd import configs, synth
    from egorear_amd.estimator import EgoPoseFormerHeatmap
    net = nets("heatmap")
    with torch.no_grad(), pytest.raises(RuntimeError):
        net(synth.synth_images(1, 2, seed=0))              # CPU tensor: refused, never computed on the host
    cpu_net = EgoPoseFormerHeatmap(**configs.heatmap_cfg()).eval()
    with torch.no_grad(), pytest.raises(RuntimeError):
        cpu_net(synth.synth_images(1, 2, seed=0))
    with pytest.raises(NotImplementedError):
        net(synth.synth_images(1, 2, seed=0).to(DEV))      # grad enabled -> training path, a "next" row


def test_state_dict_reload_invalidates_packed_weights(nets):
    from egorear_amd import synth
    net = nets("heatmap")
    img = synth.synth_images(1, 2, seed=5).to(DEV)
    with torch.no_grad():
        a = net(img)
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        sd2 = dict(sd)
        sd2["conv_heatmap.bias"] = sd["conv_heatmap.bias"] + 1.0
        net.load_state_dict(sd2, strict=True)
        b = net(img)
        net.load_state_dict(sd, strict=True)
        c = net(img)
    assert torch.allclose(b, a + 1.0, atol=1e-6) and torch.equal(a, c)


def test_torch_compile_wrapper_calls_through(nets):
    """run.py does `model.network = torch.compile(model.network, mode=...)` for every shipped YAML (compile: True)."""
    from egorear_amd import synth
    net = nets("syn")
    img = synth.synth_images(1, 4, seed=9).to(DEV)
    compiled = torch.compile(net, mode="default")
    with torch.no_grad():
        a, ha = net(img)
        b, hb = compiled(img)
    assert all(torch.equal(x, y) for x, y in zip(a, b)) and all(torch.equal(x, y) for x, y in zip(ha, hb))
    # checkpoints saved from a compiled module carry the "_orig_mod." prefix the reference strips (utils/state_dict.py:3-21)
    keys = list(compiled.state_dict().keys())
    assert all(k.startswith("_orig_mod.") for k in keys) and len(keys) == 698


def test_heatmap_for_anchor_argument(nets, calib_dir):
    """EgoPoseFormerHeatmapMVFEX.forward(img, heatmap_for_anchor): anchors come from the supplied heat map
    (heatmap_mvf_ex.py:293-296), everything else unchanged."""
    from egorear_amd import synth
    from oracle import egorear_oracle as O
    net = nets("mvfex")
    img = synth.synth_images(1, 4, seed=21)
    hfa = synth.uniform("hfa", 3, (1, 4, 15, 64, 64), -0.2, 1.2)
    with torch.no_grad():
        hms, fts = net(img.to(DEV), hfa.to(DEV))
        idx = net.__dict__["_egr_last_aux"]["argmax_idx"].cpu().long()
        sd = {k: v.cpu() for k, v in net.state_dict().items()}
        o_hms, o_fts, o_aux = O.heatmap_mvfex_forward(sd, "", img, 0.5, hfa)
    assert torch.equal(idx, o_aux["argmax_idx"]) and torch.equal(idx, hfa.view(1, 4, 15, -1).argmax(-1))
    assert float((hms[1].cpu() - o_hms[1]).abs().max()) < TOL_HM
    assert float((fts[1].cpu() - o_fts[1]).abs().max()) < 5e-4


def test_pose3d_estimator_standalone_api(nets):
    """EgoPoseFormerPose3D.forward(frame_feats_init, frame_feats_final, heatmap, ...) accepts foreign (B,V,C,H,W) tensors."""
    from egorear_amd import synth
    net = nets("syn")
    img = synth.synth_images(2, 4, seed=23).to(DEV)
    with torch.no_grad():
        preds, hms = net(img)
        _, fts = net.heatmap_estimator(img)
        a = net.pose3d_estimator(fts[0], fts[1], hms[1])                                   # our own views: zero-copy
        b = net.pose3d_estimator(fts[0].contiguous(), fts[1].contiguous(), hms[1])         # plain NCHW copies
    for p, q, r in zip(preds, a, b):
        assert torch.equal(p, q) and torch.equal(p, r)


def test_lifting_head_borrows_the_estimators_arena_for_one_call_only(nets):
    """Inside the pipeline the lifting head goes on in the heat-map estimator's abs-max arena (one clear per forward) - for that call:
    its own State keeps its own arena, so a stand-alone call afterwards does not zero records of tensors the estimator still exposes
    (a zero record = pre-scale 2^60 = fp16 overflow in the next consumer), and no forward runs out of records."""
    from egorear_amd import engine, hip, synth
    net = nets("syn")
    img = synth.synth_images(2, 4, seed=24).to(DEV)
    before = hip.ARENA_EXHAUSTED
    with torch.no_grad():
        preds, hms = net(img)
        st_h = engine._state(net.heatmap_estimator, torch.device(DEV))
        st_p = engine._state(net.pose3d_estimator, torch.device(DEV))
        assert st_p.amax is not st_h.amax
        _, fts = net.heatmap_estimator(img)
        recs = st_h.amax.buf.clone()                                                       # every record the estimator's forward left
        a = net.pose3d_estimator(fts[0].contiguous(), fts[1].contiguous(), hms[1])         # stand-alone: clears ITS arena, not the estimator's
        assert torch.equal(st_h.amax.buf, recs) and int(recs.max()) > 0
        b = net.pose3d_estimator(fts[0], fts[1], hms[1])                                   # the estimator's tensors, records intact
        assert st_p.amax is not st_h.amax
    for p, q, r in zip(preds, a, b):
        assert torch.equal(p, q) and torch.equal(p, r)
    assert hip.ARENA_EXHAUSTED == before


def test_graphed_forward_replays_identically(nets):
    from egorear_amd import synth
    from egorear_amd.runner import GraphedForward
    net = nets("syn")
    g = GraphedForward(net)
    for seed in (31, 32):
        img = synth.synth_images(2, 4, seed=seed).to(DEV)
        with torch.no_grad():
            ref_p, ref_h = net(img)
        p, h = g(img)
        assert all(torch.equal(a, b) for a, b in zip(p, ref_p)) and all(torch.equal(a, b) for a, b in zip(h, ref_h))
    assert len(g._graphs) == 1          # same shape -> one capture, replayed


def test_graphed_forward_recaptures_after_weights_change():
    """A captured hipGraph points into the packed weights of its capture: new weights (load_state_dict drops the packs) must
    lead to a new capture, never to a replay over freed / stale buffers."""
    from egorear_amd import configs, synth
    from egorear_amd.estimator import EgoPoseFormerHeatmap
    from egorear_amd.runner import GraphedForward
    net = _build(EgoPoseFormerHeatmap, configs.heatmap_cfg())
    img = synth.synth_images(2, 2, seed=33).to(DEV)
    g = GraphedForward(net)
    first = g(img).clone()
    sd = {k: (v * 1.25 if v.dtype.is_floating_point and "running_var" not in k else v) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    with torch.no_grad():
        eager = net(img)
    replayed = g(img)
    assert torch.equal(replayed, eager)
    assert not torch.equal(replayed, first)
    again = g(img)                       # and the new capture is replayed from then on
    assert torch.equal(again, eager) and len(g._graphs) == 1


def test_launch_goes_to_the_tensors_device_not_torchs_current_one():
    """hip.py derives the device (and its current stream) from the operands; with one GPU this can only pin the bookkeeping:
    the launch works from inside another stream context and leaves no device state behind."""
    from egorear_amd import hip
    x = torch.randn(4, 8, 8, 32, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        y = hip.maxpool(hip.Img(x), 3, 2, 1).t
    side.synchronize()
    ref = torch.nn.functional.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(y, ref) and hip._DEV[0] is None


@pytest.mark.parametrize("batch", [32, 64])
def test_benchmarked_batch_sizes_default_launch_policy_vs_oracle(batch, nets):
    """The sizes bench.py runs (32 / 64 frames per GPU) with the SHIPPED launch rule - at these sizes the large implicit-GEMM
    launches go to the split-bf16 kernel (persistent variant for the short-K layers) by size, not by a test switch: HIP path vs the
    CPU oracle on the first 16 frames of the batch (frames are independent), arg-max indices bit-equal, 3-D joints within 1e-3 cm,
    and a second run bitwise identical."""
    from egorear_amd import hip, synth
    from oracle import egorear_oracle as O
    assert hip.X6_MIN_ROWS > 0 and hip.X6_MIN_FLOPS > 0          # default policy
    net = nets("syn")
    img = synth.synth_images(batch, 4, seed=1234)
    prof, hip.PROFILE = [], []
    with torch.no_grad():
        preds, hms = net(img.to(DEV))
        prof, hip.PROFILE = hip.PROFILE, None
        preds2, hms2 = net(img.to(DEV))
    tags = [t for name, _, _, _, _, t in prof if name in ("egr_conv2d_nhwc_f32", "egr_conv1x1_chain_f32")]
    assert sum(t.startswith("h2 ") for t in tags) >= 35, "the large launches must have gone to the fp16-scheme kernel by size"
    assert sum(t.startswith("x6 ") for t in tags) <= 3, "only launches whose input has no abs-max record stay on the split-bf16 kernel"
    assert all(torch.equal(a, b) for a, b in zip(preds, preds2)) and all(torch.equal(a, b) for a, b in zip(hms, hms2))
    n = 16
    calib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "egorear_amd", "calib", "ego4view")
    sd = {k: v.cpu() for k, v in net.state_dict().items()}
    with torch.no_grad():
        o_preds, o_hms, o_aux = O.mvfex_forward(sd, O.make_cameras("ego4view_syn", calib), img[:n])
    aux = net.__dict__["_egr_last_aux"]
    assert torch.equal(aux["heatmap"]["argmax_idx"][:n].cpu().long(), o_aux["heatmap"]["argmax_idx"])
    for h, o in zip(hms, o_hms):
        assert torch.equal(h[:n].flatten(-2).argmax(-1).cpu(), o.flatten(-2).argmax(-1))                      # both heat-map sets
        assert float((h[:n].cpu() - o).abs().max()) < TOL_HM
    for p, o in zip(preds, o_preds):
        assert float((p[:n].cpu() - o).abs().max()) < TOL_POSE_CM


def test_a_frame_does_not_depend_on_the_batch_it_arrives_in(nets):
    """The launch rule is a function of the launch size (fp32 matrix cores at batch 2, the fp16 scheme with the batch's own
    power-of-two pre-scales at batch 64): the same two frames alone and inside a batch of 64 give the same arg-max indices and
    3-D joints within the 1e-3 cm bar (pl_wrappers/egoposeformer/pose_3d_mvf_ex.py:117-153 evaluates frame by frame)."""
    from egorear_amd import hip, synth
    net = nets("syn")
    img64 = synth.synth_images(64, 4, seed=77).to(DEV)
    lo = 10
    with torch.no_grad():
        hip.PROFILE = []
        p64, h64 = net(img64)
        tags64, hip.PROFILE = [t for name, *_, t in hip.PROFILE if name in ("egr_conv2d_nhwc_f32", "egr_conv1x1_chain_f32")], None
        idx64 = net.__dict__["_egr_last_aux"]["heatmap"]["argmax_idx"][lo:lo + 2].clone()
        hip.PROFILE = []
        p2, h2 = net(img64[lo:lo + 2].contiguous())
        tags2, hip.PROFILE = [t for name, *_, t in hip.PROFILE if name in ("egr_conv2d_nhwc_f32", "egr_conv1x1_chain_f32")], None
        idx2 = net.__dict__["_egr_last_aux"]["heatmap"]["argmax_idx"]
    assert sum(t.startswith("h2 ") for t in tags64) >= 40 and sum(t.startswith("h2 ") for t in tags2) <= 30   # different kernels did the work
    assert torch.equal(idx64, idx2)
    for a, b in zip(h64, h2):
        assert torch.equal(a[lo:lo + 2].flatten(-2).argmax(-1), b.flatten(-2).argmax(-1))
        assert float((a[lo:lo + 2] - b).abs().max()) < TOL_HM
    for a, b in zip(p64, p2):
        assert float((a[lo:lo + 2] - b).abs().max()) < TOL_POSE_CM


def test_grayscale_input_is_repeated_to_three_channels(nets):
    """resnet.py:44-46: a (B, V, H, W) batch is one plane repeated three times."""
    from egorear_amd import synth
    net = nets("heatmap")
    rgb = synth.synth_images(2, 2, seed=5).to(DEV)
    gray = rgb[:, :, 0].contiguous()
    with torch.no_grad():
        a = net(gray)
        b = net(gray.unsqueeze(2).repeat(1, 1, 3, 1, 1))
    assert a.shape == (2, 2, 15, 64, 64) and torch.equal(a, b)
    with torch.no_grad(), pytest.raises(RuntimeError, match="image batch"):
        net(gray[0])


def test_fused_transformer_layer_matches_the_per_op_launches(nets):
    """egr_joint_layer_f32 (one launch per layer behind the sampling, with the next layer's offsets / post_norm / regression head
    as tails) against the per-op launches it replaces: same arithmetic up to the order of the fp32 sums."""
    from egorear_amd import engine, synth
    for cam in ("syn", "rw"):
        net = nets(cam)
        img = synth.synth_images(3, 4, seed=41).to(DEV)
        ctm = synth.synth_coord_trans_mat(3).to(DEV) if cam == "rw" else None
        saved = engine.FUSED_LAYER
        try:
            with torch.no_grad():
                engine.FUSED_LAYER = True
                p1, h1 = net(img, ctm)
                aux1 = net.__dict__["_egr_last_aux"]["heatmap"]["argmax_idx"].clone()
                engine.FUSED_LAYER = False
                p0, h0 = net(img, ctm)
                aux0 = net.__dict__["_egr_last_aux"]["heatmap"]["argmax_idx"]
        finally:
            engine.FUSED_LAYER = saved
        assert torch.equal(aux0, aux1)
        assert torch.equal(h0[0], h1[0])                                   # the initial heat maps do not involve the layer
        assert float((h0[1] - h1[1]).abs().max()) < 2e-5
        assert torch.equal(h0[1].flatten(-2).argmax(-1), h1[1].flatten(-2).argmax(-1))
        for a, b in zip(p0, p1):
            assert float((a - b).abs().max()) < 2e-4                       # cm


def test_a_layer_outside_the_fused_kernels_shapes_keeps_the_per_op_launches():
    """pack_layers leaves `fused` empty for a transformer layer the one-launch kernel does not cover (here: FFN width 256) instead of
    failing at pack time; the per-op launches then run it - same results as the oracle's arithmetic class (finite, right shape) and
    the fused path untouched for the refiners, whose layers keep the shipped shape."""
    import copy
    from egorear_amd import configs, engine, synth
    from egorear_amd.estimator import EgoPoseFormerMVFEX
    cfg = copy.deepcopy(configs.pose3d_cfg())
    cfg["pose3d_cfg"]["transformer_cfg"]["ffn_cfg"]["feedforward_dims"] = 256
    net = EgoPoseFormerMVFEX(**cfg).eval()
    synth.load_synth(net, 42)
    net = net.to(DEV)
    img = synth.synth_images(2, 4, seed=5).to(DEV)
    with torch.no_grad():
        preds, hms = net(img)
    p3 = net.pose3d_estimator
    P = engine._state(p3, torch.device(DEV)).packs[id(p3)]
    assert all(L.fused is None for L in P.layers) and P.reg_plain is None
    assert torch.isfinite(preds[-1]).all() and preds[-1].shape == (2, 16, 3)


def test_two_forwards_in_flight_give_the_frames_of_one(nets):
    """runner.PipelinedForward: two captured forwards of one module on two streams (engine lanes: own scratch, shared packed weights).
    Every batch comes out bit-identical to the plain forward, whether the lanes run one after the other or overlap."""
    from egorear_amd import synth
    from egorear_amd.runner import PipelinedForward
    net = nets("syn")
    imgs = [synth.synth_images(2, 4, seed=60 + i).to(DEV) for i in range(4)]
    with torch.no_grad():
        refs = [tuple(t.clone() for t in net(img)[0]) + tuple(t.clone() for t in net(img)[1]) for img in imgs]
    p = PipelinedForward(net, lanes=2)
    p.prime(imgs[0])
    assert len({id(f) for f in p.forwards}) == 2 and p.forwards[0].lane != p.forwards[1].lane

    def flat(out):
        return tuple(out[0]) + tuple(out[1])
    # overlapped: two batches submitted back to back, both lanes busy; then read both
    for a, b in ((0, 1), (2, 3), (3, 0)):
        oa = p(imgs[a])
        ob = p(imgs[b])
        p.wait()
        torch.cuda.synchronize()
        for got, ref in ((flat(oa), refs[a]), (flat(ob), refs[b])):
            for g, r in zip(got, ref):
                assert torch.equal(g, r)
    # the lanes keep separate scratch but one set of packed weights
    from egorear_amd import engine
    st0 = engine._state(net, torch.device(DEV))
    with engine.use_lane(1):
        st1 = engine._state(net, torch.device(DEV))
    assert st1 is not st0 and st1.packs is st0.packs and st1.workspace.data_ptr() != st0.workspace.data_ptr()


def test_two_modules_with_their_own_launch_policies_coexist(nets):
    """Round 6: the launch policy is an object a module can hold (engine.set_policy), not process-global state.
    One module on the shipped fp16 scheme and a second on the exact bf16x3 arithmetic run alternately in one process: each keeps
    its own kernel family and packs, neither disturbs the other or the process default."""
    from egorear_amd import configs, engine, hip, synth
    from egorear_amd.estimator import EgoPoseFormerMVFEX
    net_a = nets("syn")
    net_b = _build(EgoPoseFormerMVFEX, configs.pose3d_cfg("ego4view_syn"))
    img = synth.synth_images(16, 4, seed=5).to(DEV)
    default = copy.deepcopy(hip.POLICY)

    def run(net):
        hip.PROFILE = []
        with torch.no_grad():
            p, h = net(img)
        tags, hip.PROFILE = [t for name, *_, t in hip.PROFILE if name in ("egr_conv2d_nhwc_f32", "egr_conv1x1_chain_f32")], None
        return p, h, sum(t.startswith("h2 ") for t in tags), sum(t.startswith("x6 ") for t in tags)
    try:
        engine.set_policy(net_b, hip.POLICY.exact())
        pa, ha, a_h2, a_x6 = run(net_a)
        pb, hb, b_h2, b_x6 = run(net_b)
        pa2, ha2, a2_h2, a2_x6 = run(net_a)
        assert a_h2 >= 30 and a_x6 <= 3 and (a2_h2, a2_x6) == (a_h2, a_x6)
        assert b_h2 == 0 and b_x6 >= 30
        assert hip.POLICY == default and hip.H2 is True
        assert all(torch.equal(x, y) for x, y in zip(pa, pa2)) and all(torch.equal(x, y) for x, y in zip(ha, ha2))
        for x, y in zip(ha, hb):
            assert torch.equal(x.flatten(-2).argmax(-1), y.flatten(-2).argmax(-1)) and float((x - y).abs().max()) < TOL_HM
        for x, y in zip(pa, pb):
            assert float((x - y).abs().max()) < TOL_POSE_CM
    finally:
        engine.set_policy(net_b, None)
