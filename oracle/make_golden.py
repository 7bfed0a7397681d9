"""Generate tests/golden/*.npz by running the REAL reference (/root/reference) on
PyTorch-CPU under oracle/ref_shims.py.  Build-container only; TEST INFRASTRUCTURE.

    python -m oracle.make_golden            # writes tests/golden/

Inputs and weights are not stored: they are regenerated bit-identically from
egorear_amd.synth (seeded, platform-independent), so a fixture holds only the
reference's outputs (strided slices + float64 checksums of the large tensors, full
small tensors) and the reference's state_dict key/shape list.
"""
from __future__ import annotations

import copy
import json
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from egorear_amd import configs, synth  # noqa: E402
from oracle.ref_shims import reference_importable  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
WEIGHT_SEED = 42
BATCH = 2


def summarize(name: str, t: torch.Tensor, store: dict, step_hw: int = 8, step_c: int = 1):
    t = t.detach().float()
    store[name + "_sum"] = np.float64(t.double().sum().item())
    store[name + "_sq"] = np.float64((t.double() ** 2).sum().item())
    if t.dim() == 5:
        store[name + "_sl"] = t[:, :, ::step_c, ::step_hw, ::step_hw].contiguous().numpy()
    else:
        store[name] = t.numpy()


def load_synth_into(module):
    return synth.load_synth(module, WEIGHT_SEED)


def spec_json(module):
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in module.state_dict().items()]


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    specs = {}
    with reference_importable(), torch.no_grad():
        from pose_estimation.models.estimator import (EgoPoseFormerHeatmap, EgoPoseFormerHeatmapMVFEX, EgoPoseFormerMVFEX)

        # ---- config 1/2: EgoPoseFormerHeatmap ------------------------------------
        net = EgoPoseFormerHeatmap(**copy.deepcopy(configs.heatmap_cfg())).eval()
        load_synth_into(net)
        specs["EgoPoseFormerHeatmap"] = spec_json(net)
        for seed in (0, 1):
            img = synth.synth_images(BATCH, 2, seed=seed)
            hm, feats, pyramid = net(img, return_feat=True)
            st = {}
            summarize("hm", hm, st)
            summarize("feat", feats, st, 16, 16)
            summarize("s32", pyramid[-1], st, 4, 64)
            np.savez_compressed(os.path.join(OUT, f"heatmap_s{seed}.npz"), **st)
            print("heatmap", seed, st["hm_sum"], float(hm.max()))

        # ---- config 3: EgoPoseFormerHeatmapMVFEX ---------------------------------
        net = EgoPoseFormerHeatmapMVFEX(**copy.deepcopy(configs.heatmap_mvfex_cfg())).eval()
        load_synth_into(net)
        specs["EgoPoseFormerHeatmapMVFEX"] = spec_json(net)
        for seed, scale in ((0, 1.0), (1, 1.0), (2, 0.35)):
            img = synth.synth_images(BATCH, 4, seed=seed, scale=scale)
            cap = {}
            orig = net.get_anchors_2d_from_hm

            def wrapped(h, _orig=orig, _cap=cap):
                r = _orig(h)
                _cap["pts"], _cap["maxvals"], _cap["valid"] = r
                return r
            net.get_anchors_2d_from_hm = wrapped
            hms, fts = net(img)
            net.get_anchors_2d_from_hm = orig
            st = {"scale": np.float32(scale)}
            summarize("hm_init", hms[0], st)
            summarize("hm_refined", hms[1], st)
            summarize("feat_init", fts[0], st, 16, 16)
            summarize("feat_refined", fts[1], st, 16, 16)
            pts = cap["pts"]
            st["argmax_idx"] = (torch.round(pts[..., 0] * 64) + 64 * torch.round(pts[..., 1] * 64)).to(torch.int32).numpy()
            st["anchors_2d"] = pts.numpy()
            st["maxvals"] = cap["maxvals"].numpy()
            st["anchors_valid"] = cap["valid"].numpy()
            np.savez_compressed(os.path.join(OUT, f"mvfex_s{seed}.npz"), **st)
            print("mvfex", seed, "valid frac", float(cap["valid"].float().mean()), st["hm_refined_sum"])

        # ---- config 4/5-fwd: EgoPoseFormerMVFEX (syn and rw) ----------------------
        for cam in ("ego4view_syn", "ego4view_rw"):
            net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg(cam))).eval()
            load_synth_into(net)
            specs["EgoPoseFormerMVFEX"] = spec_json(net)
            for seed in ((0, 1) if cam == "ego4view_syn" else (0,)):
                img = synth.synth_images(BATCH, 4, seed=seed)
                gt = synth.synth_gt_pose(BATCH)
                ctm = synth.synth_coord_trans_mat(BATCH) if cam == "ego4view_rw" else None
                cap = {}
                p3 = net.pose3d_estimator
                orig = p3._reproject_3d_to_2d

                def wrapped(a3, m=None, o=None, _orig=orig, _cap=cap):
                    _cap["a3_before"] = a3.clone()
                    r = _orig(a3, m, o)
                    _cap["a2"], _cap["valid"] = r
                    _cap["a3_after"] = a3.clone()
                    return r
                p3._reproject_3d_to_2d = wrapped
                preds, hms = net(img, ctm)
                p3._reproject_3d_to_2d = orig
                st = {}
                st["pred_pose"] = torch.stack(preds).numpy()
                st["anchors_2d"] = cap["a2"].float().numpy()
                st["anchors_valid"] = cap["valid"].numpy()
                st["anchors_3d_before"] = cap["a3_before"].numpy()
                st["anchors_3d_after"] = cap["a3_after"].numpy()
                summarize("hm_init", hms[0], st)
                summarize("hm_refined", hms[1], st)
                from pose_estimation.utils.loss import compute_mpjpe_batch
                st["mpjpe_mm"] = (compute_mpjpe_batch(preds[-1], gt) * 10.0).numpy()
                st["mpjpe_mm_proposal"] = (compute_mpjpe_batch(preds[0], gt) * 10.0).numpy()
                tag = "syn" if cam == "ego4view_syn" else "rw"
                np.savez_compressed(os.path.join(OUT, f"pose3d_{tag}_s{seed}.npz"), **st)
                print("pose3d", cam, seed, "valid frac", float(cap["valid"].float().mean()), "mpjpe", st["mpjpe_mm"],
                      "delta", (cap["a3_after"] - cap["a3_before"]).abs().amax(dim=(0, 1)).tolist())
    with open(os.path.join(OUT, "state_dict_spec.json"), "w") as f:
        json.dump(specs, f)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
