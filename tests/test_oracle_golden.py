"""Pin the CPU oracle (oracle/egorear_oracle.py) against outputs of the REAL reference.

tests/golden/*.npz were produced by oracle/make_golden.py, which imports /root/reference
itself (under shims for absent third-party packages) and runs it on the seeded synthetic
weights/inputs that this test regenerates.  The oracle uses the same PyTorch-CPU ops in
the same order, so agreement is expected at rounding level; the one deliberately
different formulation is the mmcv deformable-attention core (explicit bilinear gather
here, F.grid_sample in the shim), which this comparison cross-checks.
"""
import copy
import os

import numpy as np
import pytest
import torch

from egorear_amd import configs, synth
from egorear_amd.estimator import EgoPoseFormerHeatmap, EgoPoseFormerHeatmapMVFEX, EgoPoseFormerMVFEX
from oracle import egorear_oracle as O

TOL_HM = 2e-5      # heatmap values are O(1)
TOL_POSE_CM = 1e-4  # north_star tolerance is 1e-3 cm; the oracle must sit well inside it


def _sd(cls, cfg):
    return synth.synth_state_dict(synth.spec_of(cls(**copy.deepcopy(cfg))), 42)


def _check_summary(g, name, t, step_hw=8, step_c=1, tol=TOL_HM):
    t = t.float()
    sl = t[:, :, ::step_c, ::step_hw, ::step_hw].numpy()
    np.testing.assert_allclose(sl, g[name + "_sl"], rtol=0, atol=tol)
    n = t.numel()
    assert abs(t.double().sum().item() - float(g[name + "_sum"])) <= tol * n
    assert abs((t.double() ** 2).sum().item() - float(g[name + "_sq"])) <= 1e-5 * float(g[name + "_sq"]) + 1e-6


@pytest.fixture(scope="module")
def sd_heatmap():
    return _sd(EgoPoseFormerHeatmap, configs.heatmap_cfg())


@pytest.fixture(scope="module")
def sd_mvfex():
    return _sd(EgoPoseFormerHeatmapMVFEX, configs.heatmap_mvfex_cfg())


@pytest.fixture(scope="module")
def sd_full():
    return _sd(EgoPoseFormerMVFEX, configs.pose3d_cfg())


@pytest.mark.parametrize("seed", [0, 1])
def test_heatmap_matches_reference(seed, sd_heatmap, golden_dir):
    g = np.load(os.path.join(golden_dir, f"heatmap_s{seed}.npz"))
    with torch.no_grad():
        hm, feats, pyr = _heatmap(sd_heatmap, seed)
    _check_summary(g, "hm", hm)
    _check_summary(g, "feat", feats, 16, 16)
    _check_summary(g, "s32", pyr[-1], 4, 64)


def _heatmap(sd, seed):
    # keys of the standalone model carry no prefix: add a dummy one for the oracle's prefix API
    sdp = {"m." + k: v for k, v in sd.items()}
    return O.heatmap_forward(sdp, "m", synth.synth_images(2, 2, seed=seed), True)


@pytest.mark.parametrize("seed,scale", [(0, 1.0), (1, 1.0), (2, 0.35)])
def test_mvfex_matches_reference(seed, scale, sd_mvfex, golden_dir):
    g = np.load(os.path.join(golden_dir, f"mvfex_s{seed}.npz"))
    assert float(g["scale"]) == pytest.approx(scale)
    with torch.no_grad():
        hms, fts, aux = O.heatmap_mvfex_forward(sd_mvfex, "", synth.synth_images(2, 4, seed=seed, scale=scale))
    # integer / boolean results: bit-exact
    np.testing.assert_array_equal(aux["argmax_idx"].numpy().astype(np.int32), g["argmax_idx"])
    np.testing.assert_array_equal(aux["anchors_valid"].numpy(), g["anchors_valid"])
    np.testing.assert_array_equal(aux["anchors_2d"].numpy(), g["anchors_2d"])
    np.testing.assert_allclose(aux["maxvals"].numpy(), g["maxvals"], rtol=0, atol=TOL_HM)
    _check_summary(g, "hm_init", hms[0])
    _check_summary(g, "hm_refined", hms[1])
    _check_summary(g, "feat_init", fts[0], 16, 16)
    _check_summary(g, "feat_refined", fts[1], 16, 16)


def test_mvfex_goldens_cover_both_sides_of_threshold(golden_dir):
    v = np.concatenate([np.load(os.path.join(golden_dir, f"mvfex_s{s}.npz"))["anchors_valid"].ravel() for s in (0, 1, 2)])
    assert v.any() and (~v).any()


@pytest.mark.parametrize("cam,seed", [("syn", 0), ("syn", 1), ("rw", 0)])
def test_pose3d_matches_reference(cam, seed, sd_full, golden_dir, calib_dir):
    g = np.load(os.path.join(golden_dir, f"pose3d_{cam}_s{seed}.npz"))
    cams = O.make_cameras("ego4view_" + cam, calib_dir)
    ctm = synth.synth_coord_trans_mat(2) if cam == "rw" else None
    with torch.no_grad():
        preds, hms, aux = O.mvfex_forward(sd_full, cams, synth.synth_images(2, 4, seed=seed), ctm)
    pred = torch.stack(preds).numpy()
    np.testing.assert_allclose(pred, g["pred_pose"], rtol=0, atol=TOL_POSE_CM)
    np.testing.assert_array_equal(aux["pose3d"]["anchors_valid"].numpy(), g["anchors_valid"])
    np.testing.assert_allclose(aux["pose3d"]["anchors_2d"].numpy(), g["anchors_2d"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(aux["pose3d"]["anchors_3d_after"].numpy(), g["anchors_3d_after"], rtol=0, atol=TOL_POSE_CM)
    _check_summary(g, "hm_init", hms[0])
    _check_summary(g, "hm_refined", hms[1])
    gt = synth.synth_gt_pose(2)
    mp = (O.compute_mpjpe_batch(preds[-1], gt) * 10.0).numpy()
    np.testing.assert_allclose(mp, g["mpjpe_mm"], rtol=0, atol=1e-2)  # mm; 1e-3 cm == 1e-2 mm
    # F7: syn mode leaves the anchors shifted by (+12, ~0, 0) cm; rw mode does not mutate
    delta = g["anchors_3d_after"] - g["anchors_3d_before"]
    if cam == "syn":
        np.testing.assert_allclose(delta[..., 0], 12.0, atol=1e-4)
        np.testing.assert_allclose(delta[..., 1:], 0.0, atol=1e-4)
    else:
        np.testing.assert_array_equal(delta, 0.0)


@pytest.mark.parametrize("seed,scale", [(0, 1.0), (2, 0.35)])
def test_mvfex_intermediates_match_reference(seed, scale, sd_mvfex, golden_dir):
    """SURVEY.md 8c intermediate pins (oracle/make_golden_mid.py: forward hooks inside the reference's refiners): JQA query (a9),
    the deformable attention's output per view (a13), transformer layer + post_norm (a12 / a14 / a15), head offset + own-view
    projection (a16 + a11)."""
    g = np.load(os.path.join(golden_dir, f"mvfex_mid_s{seed}.npz"))
    with torch.no_grad():
        _, _, aux = O.heatmap_mvfex_forward(sd_mvfex, "", synth.synth_images(2, 4, seed=seed, scale=scale), capture=True)
    for name in ("front_left", "front_right", "back_left", "back_right"):
        c = aux["mid"][name]
        np.testing.assert_allclose(c["query"].numpy(), g[name + "_query"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(torch.stack(c["msda"])[..., ::4].numpy(), g[name + "_msda"], rtol=0, atol=2e-5)
        np.testing.assert_allclose(c["post_norm"].numpy(), g[name + "_post_norm"], rtol=0, atol=5e-5)
        hs = c["head_sum"]
        np.testing.assert_allclose(hs[:, ::8, ::4, ::4].numpy(), g[name + "_head_sum_sl"], rtol=0, atol=5e-5)
        assert abs(hs.double().sum().item() - float(g[name + "_head_sum_sum"])) <= 5e-5 * hs.numel()
