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


def generate_target(joints: np.ndarray, image_size=872, heatmap_size=64, num_joints=15, sigma=1):
    """generate_heatmap.py:10-48, restated (joints (J,2) pixel coordinates) -> (J, hs, hs) float32."""
    target = np.zeros((num_joints, heatmap_size, heatmap_size), dtype=np.float32)
    tmp_size = sigma * 3
    for j in range(num_joints):
        stride = image_size / heatmap_size
        mu_x = int(joints[j][0] / stride + 0.5)
        mu_y = int(joints[j][1] / stride + 0.5)
        ul = [int(mu_x - tmp_size), int(mu_y - tmp_size)]
        br = [int(mu_x + tmp_size + 1), int(mu_y + tmp_size + 1)]
        if ul[0] >= heatmap_size or ul[1] >= heatmap_size or br[0] < 0 or br[1] < 0:
            continue
        size = 2 * tmp_size + 1
        x = np.arange(0, size, 1, np.float32)
        y = x[:, np.newaxis]
        x0 = y0 = size // 2
        g = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))
        gx = max(0, -ul[0]), min(br[0], heatmap_size) - ul[0]
        gy = max(0, -ul[1]), min(br[1], heatmap_size) - ul[1]
        ix = max(0, ul[0]), min(br[0], heatmap_size)
        iy = max(0, ul[1]), min(br[1], heatmap_size)
        target[j][iy[0]:iy[1], ix[0]:ix[1]] = g[gy[0]:gy[1], gx[0]:gx[1]]
    return target
