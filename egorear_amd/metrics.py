"""Pose evaluation metrics on the device.

Mirrors the reference's `evaluate_pose` (pl_wrappers/egoposeformer/pose_3d_mvf_ex.py:317-333) and the helpers it
calls (utils/loss.py:9-48, models/utils/pose_metric.py:104-167) — same names, same units — but as ONE HIP kernel
on the device tensors, instead of a device-to-host copy and a Python loop of numpy SVDs."""
from __future__ import annotations

from collections import OrderedDict

import torch

from . import hip

CM2MM = 10.0


def pose_metrics(pred_pose: torch.Tensor, gt_pose: torch.Tensor, pck_threshold_mm: float = 150.0, n_auc: int = 31,
                 return_aligned: bool = False):
    """pred_pose, gt_pose (B, J, 3) in cm on the device -> (B, 4) fp32 [mpjpe_mm, pa_mpjpe_mm, pck_3d %, auc_3d %]."""
    if not pred_pose.is_cuda:
        raise RuntimeError("egorear_amd.metrics: device tensors expected (no CPU path)")
    pred = pred_pose.detach().to(torch.float32).contiguous()
    gt = gt_pose.detach().to(device=pred.device, dtype=torch.float32).contiguous()   # the dataset yields float64 gt
    if pred.shape != gt.shape or pred.dim() != 3 or pred.shape[-1] != 3:
        raise RuntimeError("egorear_amd.metrics: pred / gt must both be (B, J, 3)")
    B, J = pred.shape[:2]
    out = torch.empty((B, 4), device=pred.device, dtype=torch.float32)
    aligned = torch.empty_like(pred) if return_aligned else None
    hip._launch("egr_pose_metrics_f32", hip.lib.egr_pose_metrics_f32, hip._p(pred), hip._p(gt), B, J, float(pck_threshold_mm), n_auc,
                hip._p(out), hip._p(aligned), hip._stream())
    return (out, aligned) if return_aligned else out


def compute_mpjpe_batch(pred_keypoints: torch.Tensor, gt_keypoints: torch.Tensor) -> torch.Tensor:
    """utils/loss.py:9-12 (result in the inputs' unit: cm)."""
    return pose_metrics(pred_keypoints, gt_keypoints)[:, 0] / CM2MM


def evaluate_pose(pred_pose: torch.Tensor, gt_pose: torch.Tensor, prefix: str) -> "OrderedDict[str, torch.Tensor]":
    """Same keys and units as the reference's evaluate_pose; values stay on the device."""
    m = pose_metrics(pred_pose, gt_pose)
    out = OrderedDict()
    out[prefix + "_mpjpe"] = m[:, 0]
    out[prefix + "_pa_mpjpe"] = m[:, 1]
    out[prefix + "_pck_3d"] = m[:, 2]
    out[prefix + "_auc_3d"] = m[:, 3]
    return out


# --------------------------------------------------------------------------- GT heat maps

def _gauss_table(sigma: float):
    """The reference's window, computed as it computes it (generate_heatmap.py:33-37): float32 numpy."""
    import numpy as np
    tmp_size = sigma * 3
    size = 2 * tmp_size + 1
    x = np.arange(0, size, 1, np.float32)
    y = x[:, np.newaxis]
    x0 = y0 = size // 2
    g = np.exp(-((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))
    return int(tmp_size), np.ascontiguousarray(g.astype(np.float32))


def generate_target(joints: torch.Tensor, image_size: int = 872, heatmap_size: int = 64, sigma: float = 1.0) -> torch.Tensor:
    """generate_heatmap.py:10-48 on the device: joints (..., J, 2) pixel coordinates (any float dtype, device tensor)
    -> (..., J, heatmap_size, heatmap_size) fp32."""
    if not joints.is_cuda:
        raise RuntimeError("egorear_amd.metrics.generate_target: device tensor expected (no CPU path)")
    j64 = joints.detach().to(torch.float64).contiguous()
    lead = tuple(j64.shape[:-1])
    maps = j64.numel() // 2
    tmp, g = _gauss_table(sigma)
    gt = torch.from_numpy(g).to(joints.device)
    out = torch.empty(lead + (heatmap_size, heatmap_size), device=joints.device, dtype=torch.float32)
    hip._launch("egr_gt_heatmap_f32", hip.lib.egr_gt_heatmap_f32, hip._p(j64, torch.float64), maps, float(image_size), heatmap_size, tmp,
                hip._p(gt), hip._p(out), hip._stream())
    return out
