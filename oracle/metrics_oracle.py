"""CPU oracle for the pose evaluation metrics.  TEST INFRASTRUCTURE ONLY.

Restates, in numpy / torch-CPU, `evaluate_pose` of pl_wrappers/egoposeformer/pose_3d_mvf_ex.py:317-333 and its helpers
compute_mpjpe_batch / compute_pck_3d_batch / compute_auc_3d_batch (utils/loss.py:9-48) and
batch_compute_similarity_transform_numpy -> compute_similarity_transform (models/utils/pose_metric.py:104-167).
Pinned by tests/golden/metrics.npz, produced by the reference's own functions (oracle/make_golden_metrics.py)."""
from __future__ import annotations

import numpy as np
import torch


def compute_similarity_transform(S1: np.ndarray, S2: np.ndarray) -> np.ndarray:
    """models/utils/pose_metric.py:122-167, (J,3) inputs (the 'transposed' branch)."""
    S1, S2 = S1.T, S2.T
    mu1 = S1.mean(axis=1, keepdims=True)
    mu2 = S2.mean(axis=1, keepdims=True)
    X1, X2 = S1 - mu1, S2 - mu2
    var1 = np.sum(X1 ** 2)
    K = X1.dot(X2.T)
    U, s, Vh = np.linalg.svd(K)
    V = Vh.T
    Z = np.eye(U.shape[0])
    Z[-1, -1] *= np.sign(np.linalg.det(U.dot(V.T)))
    R = V.dot(Z.dot(U.T))
    scale = np.trace(R.dot(K)) / var1
    t = mu2 - scale * (R.dot(mu1))
    return (scale * R.dot(S1) + t).T


def evaluate_pose(pred: torch.Tensor, gt: torch.Tensor, cm2mm: float = 10.0):
    """-> dict of per-sample numpy arrays: mpjpe, pa_mpjpe (mm), pck_3d, auc_3d (%)."""
    s1 = np.stack([compute_similarity_transform(p, g) for p, g in zip(pred.numpy(), gt.float().numpy())])
    s1 = torch.from_numpy(s1.astype(np.float32))

    def mpjpe(a, b):
        return torch.linalg.norm(a - b, dim=-1, ord=2).mean(dim=1)

    def pck(a, b, thr):
        d = torch.linalg.norm(b - a, axis=-1)
        return torch.sum(d <= thr, axis=1) / b.size()[1]
    thresholds = np.linspace(0, 150, 31).tolist()
    auc = torch.stack([pck(pred * cm2mm, gt * cm2mm, t) for t in thresholds], axis=-1).mean(axis=-1)
    return {"mpjpe": (mpjpe(pred, gt) * cm2mm).numpy(), "pa_mpjpe": (mpjpe(s1, gt) * cm2mm).numpy(),
            "pck_3d": (pck(pred * cm2mm, gt * cm2mm, 150) * 100.0).numpy(), "auc_3d": (auc * 100.0).numpy(),
            "aligned": s1.numpy()}
