"""tests/golden/metrics.npz from the REAL reference's metric functions (utils/loss.py, models/utils/pose_metric.py —
plain numpy/torch modules, importable without shims).  Build-container only; TEST INFRASTRUCTURE.
    python -m oracle.make_golden_metrics"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from egorear_amd import synth  # noqa: E402
from oracle.ref_shims import reference_importable  # noqa: E402


def cases():
    gt = synth.synth_gt_pose(24, seed=1235)
    pred = gt + synth.normalish("metric.noise", 7, (24, 16, 3)) * torch.linspace(0.2, 25.0, 24).view(24, 1, 1)
    pred[3] = gt[3]                                              # perfect prediction
    pred[4] = gt[4] * 1.3 + torch.tensor([5.0, -3.0, 2.0])        # pure similarity: PA-MPJPE ~ 0
    return pred.float().contiguous(), gt.float().contiguous()


def main():
    pred, gt = cases()
    with reference_importable():
        from pose_estimation.utils.loss import compute_auc_3d_batch, compute_mpjpe_batch, compute_pck_3d_batch
        from pose_estimation.models.utils.pose_metric import batch_compute_similarity_transform_numpy
        s1 = batch_compute_similarity_transform_numpy(pred, gt.to(dtype=torch.float))
        out = {
            "mpjpe": (compute_mpjpe_batch(pred, gt) * 10.0).numpy(),
            "pa_mpjpe": (compute_mpjpe_batch(s1, gt) * 10.0).numpy(),
            "pck_3d": (compute_pck_3d_batch(pred * 10.0, gt * 10.0) * 100.0).numpy(),
            "auc_3d": (compute_auc_3d_batch(pred * 10.0, gt * 10.0) * 100.0).numpy(),
            "aligned": s1.numpy(),
        }
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "metrics.npz"), **out)
    print({k: v[:5] for k, v in out.items() if k != "aligned"})


if __name__ == "__main__":
    main()
