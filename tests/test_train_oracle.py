"""Pin the training-step oracle (oracle/train_oracle.py over oracle/egorear_oracle.py in train mode) against the REAL
reference network run under autograd (tests/golden/train_rw_s0.npz, made by oracle/make_golden_train.py): loss terms,
which parameters receive a gradient at all, every parameter's gradient norm and 16 samples, the BatchNorm buffers after
the step, the clipped-gradient norm and the AdamW update."""
import copy
import os

import numpy as np
import pytest
import torch

from egorear_amd import configs, synth
from egorear_amd.estimator import EgoPoseFormerMVFEX
from oracle import egorear_oracle as O
from oracle import train_oracle as T

CALIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "egorear_amd", "calib", "ego4view")


@pytest.fixture(scope="module")
def step(golden_dir):
    torch.set_num_threads(8)
    net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg("ego4view_rw")))
    sd = synth.synth_state_dict(synth.spec_of(net), 42)
    names = [k for k, _ in net.named_parameters()]
    cams = O.make_cameras("ego4view_rw", CALIB)
    B = 2
    losses, grads, upd, outs = T.forward_backward(sd, cams, synth.synth_images(B, 4, seed=0), synth.synth_coord_trans_mat(B),
                                                  synth.synth_gt_pose(B), T.synth_gt_heatmap(B), names)
    return np.load(os.path.join(golden_dir, "train_rw_s0.npz")), sd, names, losses, grads, upd, outs


def test_losses_and_outputs(step):
    g, sd, names, losses, grads, upd, (preds, hms) = step
    for k, v in losses.items():
        assert abs(v - float(g["loss_" + k])) <= 2e-5 * abs(float(g["loss_" + k])), k
    np.testing.assert_allclose(torch.stack(preds).detach().numpy(), g["pred_pose"], rtol=0, atol=1e-4)
    for i, h in enumerate(hms):
        assert abs(h.detach().double().sum().item() - float(g[f"hm{i}_sum"])) <= 2e-5 * h.numel()


def test_gradients(step):
    g, sd, names, losses, grads, upd, _ = step
    assert list(g["param_names"]) == names
    present = np.array([grads[k] is not None for k in names])
    assert (present == g["grad_present"]).all(), [n for n, a, b in zip(names, present, g["grad_present"]) if a != b]
    scale = float(g["grad_norm"].max())
    for i, k in enumerate(names):
        if not present[i]:
            continue
        gn = float(g["grad_norm"][i])
        assert abs(grads[k].double().norm().item() - gn) <= 2e-4 * gn + 1e-7 * scale, k
        np.testing.assert_allclose(T.sample(grads[k]), g["grad_samples"][i], rtol=2e-3, atol=2e-5 * max(gn, 1e-3), err_msg=k)


def test_batchnorm_buffers(step):
    g, sd, names, losses, grads, upd, _ = step
    for k, ref in zip(g["bn_names"], g["bn_samples"]):
        k = str(k)
        np.testing.assert_allclose(T.sample(upd[k].float(), 8), ref, rtol=1e-5, atol=1e-6, err_msg=k)


def test_optimizer_step(step):
    g, sd, names, losses, grads, upd, _ = step
    params = {k: sd[k].clone() for k in names}
    before = {k: T.sample(params[k]).astype(np.float64) for k in names}
    total_norm, _ = T.optimizer_step(params, grads)
    assert abs(total_norm - float(g["grad_total_norm"])) <= 2e-4 * float(g["grad_total_norm"])
    for i, k in enumerate(names):
        d = (T.sample(params[k]).astype(np.float64) - before[k]).astype(np.float32)
        # AdamW's first step moves every touched element by ~lr*sign(grad): elements whose gradient is ~0 are ill-conditioned
        ok = np.abs(g["grad_samples"][i]) > 1e-6 * max(float(g["grad_norm"][i]), 1e-12)
        np.testing.assert_allclose(d[ok], g["param_delta_samples"][i][ok], rtol=0, atol=2e-5, err_msg=k)
        if not g["grad_present"][i]:
            assert (d == 0).all(), k


def _check_case(g, names, losses, grads, upd, hms):
    for k, v in losses.items():
        assert abs(v - float(g["loss_" + k])) <= 2e-5 * abs(float(g["loss_" + k])), k
    for i, h in enumerate(hms):
        assert abs(h.detach().double().sum().item() - float(g[f"hm{i}_sum"])) <= 2e-5 * h.numel()
    assert list(g["param_names"]) == names
    present = np.array([grads[k] is not None for k in names])
    assert (present == g["grad_present"]).all(), [n for n, a, b in zip(names, present, g["grad_present"]) if a != b]
    scale = float(g["grad_norm"].max())
    for i, k in enumerate(names):
        if present[i]:
            gn = float(g["grad_norm"][i])
            assert abs(grads[k].double().norm().item() - gn) <= 2e-4 * gn + 1e-7 * scale, k
            np.testing.assert_allclose(T.sample(grads[k]), g["grad_samples"][i], rtol=2e-3, atol=2e-5 * max(gn, 1e-3), err_msg=k)
    for k, ref in zip(g["bn_names"], g["bn_samples"]):
        np.testing.assert_allclose(T.sample(upd[str(k)].float(), 8), ref, rtol=1e-5, atol=1e-6, err_msg=str(k))


def test_heatmap_stage_training_matches_reference(golden_dir):
    """Stage 1 of the reference's schedule (pl_wrappers/egoposeformer/heatmap.py): one stereo estimator, MSE loss."""
    from egorear_amd.estimator import EgoPoseFormerHeatmap
    net = EgoPoseFormerHeatmap(**copy.deepcopy(configs.heatmap_cfg()))
    sd = synth.synth_state_dict(synth.spec_of(net), 42)
    names = [k for k, _ in net.named_parameters()]
    losses, grads, upd, hms = T.forward_backward_heatmap(sd, synth.synth_images(2, 2, seed=0), T.synth_gt_heatmap(2), names)
    _check_case(np.load(os.path.join(golden_dir, "train_heatmap_s0.npz")), names, losses, grads, upd, hms)


def test_mvfex_stage_training_matches_reference(golden_dir):
    """Stage 2 (heatmap_mvf_ex.py): refiners + initial heat-map heads on encoders that run under no_grad in train() mode."""
    from egorear_amd.estimator import EgoPoseFormerHeatmapMVFEX
    net = EgoPoseFormerHeatmapMVFEX(**copy.deepcopy(configs.heatmap_mvfex_cfg()))
    sd = synth.synth_state_dict(synth.spec_of(net), 42)
    names = [k for k, _ in net.named_parameters()]
    losses, grads, upd, hms = T.forward_backward_mvfex(sd, synth.synth_images(2, 4, seed=0), T.synth_gt_heatmap(2), names)
    _check_case(np.load(os.path.join(golden_dir, "train_mvfex_s0.npz")), names, losses, grads, upd, hms)
