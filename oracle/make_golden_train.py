"""Golden vectors for the training step: the REAL reference network (/root/reference, under oracle/ref_shims.py) in
train mode, the real MpjpeLoss, torch autograd, clip_grad_norm_ and torch.optim.AdamW.  Build-container only.

    python -m oracle.make_golden_train     # writes tests/golden/train_rw_s0.npz

Stored per parameter: gradient presence, L2 norm, 16 strided samples; after one optimizer step: 16 strided samples of
the parameter; BatchNorm buffers after the forward: 8 samples each; the loss terms; the outputs' checksums.
"""
from __future__ import annotations

import copy
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from egorear_amd import configs, synth  # noqa: E402
from oracle import train_oracle as T  # noqa: E402
from oracle.ref_shims import reference_importable  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
BATCH = 2


def _record(st, net, losses, total, outs, prefix=""):
    st.update({"loss_" + k: np.float64(v.item()) for k, v in losses.items()})
    st["loss_total"] = np.float64(total.item())
    for i, h in enumerate(outs):
        st[f"hm{i}_sum"] = np.float64(h.detach().double().sum().item())
        st[f"hm{i}_sq"] = np.float64((h.detach().double() ** 2).sum().item())
    names, has, norms, samples = [], [], [], []
    for k, p in net.named_parameters():
        names.append(k)
        has.append(p.grad is not None)
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        norms.append(g.double().norm().item())
        samples.append(T.sample(g))
    st["param_names"] = np.array(names)
    st["grad_present"] = np.array(has)
    st["grad_norm"] = np.array(norms, dtype=np.float64)
    st["grad_samples"] = np.stack(samples)
    bn_names, bn_vals = [], []
    for k, b in net.named_buffers():
        if "running_" in k or "num_batches_tracked" in k:
            bn_names.append(k)
            bn_vals.append(T.sample(b.float(), 8))
    st["bn_names"], st["bn_samples"] = np.array(bn_names), np.stack(bn_vals)


def heatmap_cases():
    """The two heat-map training stages (pl_wrappers/egoposeformer/heatmap.py:94-110, heatmap_mvf_ex.py:104-127): MSELoss."""
    with reference_importable():
        from pose_estimation.models.estimator import EgoPoseFormerHeatmap, EgoPoseFormerHeatmapMVFEX
        crit = torch.nn.MSELoss(reduction="mean")
        gt_hm = T.synth_gt_heatmap(BATCH)
        # ---- stage 1: one stereo heat-map estimator
        net = EgoPoseFormerHeatmap(**copy.deepcopy(configs.heatmap_cfg()))
        synth.load_synth(net, 42)
        net.train()
        img = synth.synth_images(BATCH, 2, seed=0)
        hm = net(img)
        loss = sum(crit(hm[:, v], gt_hm[:, v]) * T.W_HEATMAP for v in range(2))
        loss.backward()
        st = {}
        _record(st, net, {"heatmap_loss_0": loss}, loss, [hm])
        np.savez_compressed(os.path.join(OUT, "train_heatmap_s0.npz"), **st)
        print("heatmap stage:", float(loss), "params without grad:", int((~st["grad_present"]).sum()))
        # ---- stage 2: multi-view refinement on frozen (no_grad, train-mode) encoders
        net = EgoPoseFormerHeatmapMVFEX(**copy.deepcopy(configs.heatmap_mvfex_cfg()))
        synth.load_synth(net, 42)
        net.train()
        img = synth.synth_images(BATCH, 4, seed=0)
        hms, feats = net(img)
        losses = {f"heatmap_loss_{i}": sum(crit(h[:, v], gt_hm[:, v]) * T.W_HEATMAP for v in range(4)) for i, h in enumerate(hms[0:])}
        total = sum(losses.values())
        total.backward()
        st = {}
        _record(st, net, losses, total, hms)
        np.savez_compressed(os.path.join(OUT, "train_mvfex_s0.npz"), **st)
        print("mvfex stage:", {k: float(v) for k, v in losses.items()}, "params without grad:", int((~st["grad_present"]).sum()))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    with reference_importable():
        from pose_estimation.models.estimator import EgoPoseFormerMVFEX
        from pose_estimation.models.utils.pose_metric import MpjpeLoss
        net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg("ego4view_rw")))
        synth.load_synth(net, 42)
        net.train()                                                        # pose_3d_mvf_ex.py:115
        crit = MpjpeLoss()
        img = synth.synth_images(BATCH, 4, seed=0)
        gt_pose = synth.synth_gt_pose(BATCH)
        ctm = synth.synth_coord_trans_mat(BATCH)
        gt_hm = T.synth_gt_heatmap(BATCH)
        preds, hms = net(img, ctm, None)
        losses = {}
        for i, p in enumerate(preds):                                       # pose_3d_mvf_ex.py:133-137
            losses["mpjpe_loss_%d" % i] = crit(p, gt_pose) * T.W_MPJPE
        for i, h in enumerate(hms):                                         # :139-143
            losses["heatmap_loss_%d" % i] = sum(crit(h[:, v], gt_hm[:, v]) * T.W_HEATMAP for v in range(4))
        total = sum(losses.values())
        total.backward()
        st = {"loss_" + k: np.float64(v.item()) for k, v in losses.items()}
        st["loss_total"] = np.float64(total.item())
        st["pred_pose"] = torch.stack(preds).detach().numpy()
        for i, h in enumerate(hms):
            st[f"hm{i}_sum"] = np.float64(h.detach().double().sum().item())
            st[f"hm{i}_sq"] = np.float64((h.detach().double() ** 2).sum().item())
        names, has, norms, samples = [], [], [], []
        for k, p in net.named_parameters():
            names.append(k)
            has.append(p.grad is not None)
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            norms.append(g.double().norm().item())
            samples.append(T.sample(g))
        st["param_names"] = np.array(names)
        st["grad_present"] = np.array(has)
        st["grad_norm"] = np.array(norms, dtype=np.float64)
        st["grad_samples"] = np.stack(samples)
        bn_names, bn_vals = [], []
        for k, b in net.named_buffers():
            if "running_" in k or "num_batches_tracked" in k:
                bn_names.append(k)
                bn_vals.append(T.sample(b.float(), 8))
        st["bn_names"], st["bn_samples"] = np.array(bn_names), np.stack(bn_vals)
        # one optimisation step exactly as Lightning would run it: clip -> AdamW (configure_optimizers :219-234)
        no_decay = [p for k, p in net.named_parameters() if T.is_no_decay(k)]
        other = [p for k, p in net.named_parameters() if not T.is_no_decay(k)]
        opt = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0}, {"params": other, "weight_decay": T.WEIGHT_DECAY}], lr=T.LR)
        st["grad_total_norm"] = np.float64(float(torch.nn.utils.clip_grad_norm_(net.parameters(), T.CLIP_NORM)))
        before = {k: T.sample(p) for k, p in net.named_parameters()}
        opt.step()
        st["param_delta_samples"] = np.stack([(T.sample(p).astype(np.float64) - before[k].astype(np.float64)).astype(np.float32)
                                              for k, p in net.named_parameters()])
        np.savez_compressed(os.path.join(OUT, "train_rw_s0.npz"), **st)
        print({k: float(v) for k, v in st.items() if k.startswith("loss_")}, "total grad norm", st["grad_total_norm"],
              "params without grad:", int((~st["grad_present"]).sum()))


if __name__ == "__main__":
    main()
    heatmap_cases()
