"""CPU oracle for the training step of the hot path.  TEST INFRASTRUCTURE ONLY.

Restates, over the functional oracle in egorear_oracle.py, what the reference's Lightning wrapper does around
`self.network(...)` in one optimisation step (paths relative to /root/reference/pose_estimation/):

  * training_step loss            pl_wrappers/egoposeformer/pose_3d_mvf_ex.py:114-150
      MpjpeLoss                   models/utils/pose_metric.py:10-16 (used for poses AND for heat maps: the L2 norm runs over
                                  the last axis, i.e. over the 64 columns of every heat-map row)
  * parameter groups / AdamW      pose_3d_mvf_ex.py:219-234 (encoder_lr_scale == 1.0 branch): names containing
                                  'norm' | 'bn' | 'ln' | 'bias' get weight_decay 0
  * gradient clipping             configs/ego4view_rw_pose3d.yaml trainer.gradient_clip_val 5.0, algorithm norm
                                  (Lightning -> torch.nn.utils.clip_grad_norm_)
  * warm-up                       pose_3d_mvf_ex.py:212-217: lr is rescaled AFTER optimizer.step with the pre-increment
                                  trainer.global_step: update 1 at the full lr, update t >= 2 at lr * min(1, (t - 1) / 500)

The Lightning wrapper itself cannot be imported here (no pytorch_lightning); oracle/make_golden_train.py pins this file
against the real reference *network* + the real MpjpeLoss run under autograd.
torch.optim.AdamW / clip_grad_norm_ are the reference's own third-party calls and are used as such.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import egorear_oracle as O

W_MPJPE, W_HEATMAP = 0.1, 10.0           # configs/ego4view_rw_pose3d.yaml
LR, WEIGHT_DECAY, CLIP_NORM, WARMUP_ITERS = 1e-3, 5e-4, 5.0, 500


def synth_gt_heatmap(batch: int) -> torch.Tensor:
    """(B,4,15,64,64) Gaussian GT maps (generate_heatmap.py:10-48 as restated in metrics_oracle) at the seeded joint
    positions of egorear_amd.synth.synth_joint_px."""
    import numpy as np
    from egorear_amd import synth
    from .metrics_oracle import generate_target
    px = synth.synth_joint_px(batch).numpy().reshape(batch * 4, 15, 2)
    maps = np.stack([generate_target(px[i]) for i in range(batch * 4)])
    return torch.from_numpy(maps.reshape(batch, 4, 15, 64, 64).astype(np.float32))


def sample(t: torch.Tensor, n: int = 16):
    """n strided samples of a tensor (zero padded) - the per-parameter fingerprint stored in the training goldens."""
    import numpy as np
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    v = f[::step][:n].float().cpu().numpy().copy()
    return np.pad(v, (0, n - len(v)))


def mpjpe_loss(pred, gt):
    """models/utils/pose_metric.py:14-16."""
    return torch.mean(torch.linalg.norm(gt - pred, dim=-1, ord=2))


def training_loss(preds, hms, gt_pose, gt_heatmap, w_mpjpe: float = W_MPJPE, w_heatmap: float = W_HEATMAP) -> "OrderedDict[str, torch.Tensor]":
    """pose_3d_mvf_ex.py:133-145: one MPJPE term per pose prediction, one heat-map term per heat-map set (sum over views)."""
    d = OrderedDict()
    for i, p in enumerate(preds):
        d["mpjpe_loss_%d" % i] = mpjpe_loss(p, gt_pose) * w_mpjpe
    V = gt_heatmap.shape[1]
    for i, h in enumerate(hms):
        d["heatmap_loss_%d" % i] = sum(mpjpe_loss(h[:, v], gt_heatmap[:, v]) * w_heatmap for v in range(V))
    return d


def is_no_decay(name: str) -> bool:
    """pose_3d_mvf_ex.py:223."""
    return ("norm" in name) or ("bn" in name) or ("ln" in name) or ("bias" in name)


def forward_backward(sd: Dict[str, torch.Tensor], cams, img, ctm, gt_pose, gt_heatmap, param_names):
    """One training forward/backward in train mode.  `param_names`: the keys of `sd` that are nn.Parameters.
    Returns (losses, grads {name: tensor | None}, bn buffer updates, outputs)."""
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in param_names}
    work = dict(sd)
    work.update(leaves)
    with O.bn_train() as upd, torch.enable_grad():
        preds, hms, _ = O.mvfex_forward(work, cams, img, ctm)
        losses = training_loss(preds, hms, gt_pose, gt_heatmap)
        total = sum(losses.values())
        total.backward()
    grads = {k: v.grad for k, v in leaves.items()}
    return {k: float(v.detach()) for k, v in losses.items()}, grads, dict(upd), (preds, hms)


def mse_heatmap_losses(hms, gt_heatmap, w_heatmap: float = W_HEATMAP) -> "OrderedDict[str, torch.Tensor]":
    """pl_wrappers/egoposeformer/heatmap.py:94-110 (one heat-map set) and heatmap_mvf_ex.py:104-127 (all sets, the initial
    one included): per set, the sum over views of nn.MSELoss(reduction="mean") * w_heatmap (get_loss :215-218 / :258-261)."""
    d = OrderedDict()
    for i, h in enumerate(hms):
        V = h.shape[1]          # the stage-1 model sees one stereo pair: views [0, V) of the ground truth
        d["heatmap_loss_%d" % i] = sum(torch.nn.functional.mse_loss(h[:, v], gt_heatmap[:, v]) * w_heatmap for v in range(V))
    return d


def forward_backward_heatmap(sd: Dict[str, torch.Tensor], img, gt_heatmap, param_names):
    """Training forward/backward of EgoPoseFormerHeatmap as PoseHeatmapLightningModel.training_step runs it."""
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in param_names}
    work = {"m." + k: v for k, v in sd.items()}
    work.update({"m." + k: v for k, v in leaves.items()})
    with O.bn_train() as upd, torch.enable_grad():
        hm = O.heatmap_forward(work, "m", img)
        losses = mse_heatmap_losses([hm], gt_heatmap)
        sum(losses.values()).backward()
    return ({k: float(v.detach()) for k, v in losses.items()}, {k: v.grad for k, v in leaves.items()},
            {k[2:]: v for k, v in upd.items()}, [hm])


MVFEX_HEATMAP_FLAGS = dict(full_training=False, use_pred_heatmap_init=False, no_detach_feat_init=False, detach_heatmap_feat=False)


def forward_backward_mvfex(sd: Dict[str, torch.Tensor], img, gt_heatmap, param_names):
    """Training forward/backward of EgoPoseFormerHeatmapMVFEX with the constructor defaults of the shipped
    *_heatmap_mvfex-n1_jqa.yaml configs (encoders under no_grad but in train() mode, nothing detached downstream)."""
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in param_names}
    work = dict(sd)
    work.update(leaves)
    with O.bn_train() as upd, torch.enable_grad():
        hms, feats, _ = O.heatmap_mvfex_forward(work, "", img, **MVFEX_HEATMAP_FLAGS)
        losses = mse_heatmap_losses(hms, gt_heatmap)
        sum(losses.values()).backward()
    return {k: float(v.detach()) for k, v in losses.items()}, {k: v.grad for k, v in leaves.items()}, dict(upd), hms


def optimizer_step(params: Dict[str, torch.Tensor], grads: Dict[str, Optional[torch.Tensor]], state: Optional[dict] = None,
                   lr: float = LR, weight_decay: float = WEIGHT_DECAY, clip: float = CLIP_NORM):
    """clip_grad_norm_ + AdamW.step with the reference's two parameter groups.  Mutates `params` in place; parameters
    whose grad is None are left untouched (torch skips them, weight decay included).  Returns (total_norm, optimizer)."""
    ps = OrderedDict((k, torch.nn.Parameter(v)) for k, v in params.items())
    for k, p in ps.items():
        p.grad = None if grads[k] is None else grads[k].clone()
    no_decay = [p for k, p in ps.items() if is_no_decay(k)]
    other = [p for k, p in ps.items() if not is_no_decay(k)]
    opt = state if state is not None else torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0},
                                                              {"params": other, "weight_decay": weight_decay}], lr=lr)
    total_norm = torch.nn.utils.clip_grad_norm_(list(ps.values()), clip)
    opt.step()
    for k, p in ps.items():
        params[k] = p.data
    return float(total_norm), opt


class OracleTrainer:
    """Several optimisation steps of config 5 exactly as Lightning would run the reference: persistent torch.optim.AdamW
    (two parameter groups), clip_grad_norm_, the warm-up hook of pose_3d_mvf_ex.py:212-217 (applied AFTER each step with the
    pre-increment global_step: update t + 1 runs at lr * min(1, t / warmup_iters)), and BatchNorm buffers carried from step to step."""

    def __init__(self, sd: Dict[str, torch.Tensor], param_names, cams, lr: float = LR, weight_decay: float = WEIGHT_DECAY,
                 clip: float = CLIP_NORM, warmup_iters: int = WARMUP_ITERS):
        self.sd = {k: v.clone() for k, v in sd.items()}
        self.names, self.cams, self.lr, self.clip, self.warmup = list(param_names), cams, lr, clip, warmup_iters
        self.params = OrderedDict((k, torch.nn.Parameter(self.sd[k].clone())) for k in self.names)
        no_decay = [p for k, p in self.params.items() if is_no_decay(k)]
        other = [p for k, p in self.params.items() if not is_no_decay(k)]
        self.opt = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0}, {"params": other, "weight_decay": weight_decay}], lr=lr)
        self.global_step = 0

    def step(self, img, ctm, gt_pose, gt_heatmap):
        for k, p in self.params.items():
            self.sd[k] = p.data
        losses, grads, upd, _ = forward_backward(self.sd, self.cams, img, ctm, gt_pose, gt_heatmap, self.names)
        self.sd.update(upd)                                   # running statistics / num_batches_tracked of the training forward
        for k, p in self.params.items():
            p.grad = grads[k]
        total_norm = float(torch.nn.utils.clip_grad_norm_(list(self.params.values()), self.clip))
        self.opt.step()
        # optimizer_step hook (pose_3d_mvf_ex.py:212-217): runs after optimizer.step() but BEFORE Lightning marks the step
        # completed (optim_progress.optimizer.step.increment_completed() follows the hook), so it sees the pre-increment
        # trainer.global_step: after update t the lr is lr * min(1, t / warmup), and update 2 runs at 1 / warmup
        if self.global_step < self.warmup:
            scale = min(1.0, float(self.global_step + 1) / float(self.warmup))
            for pg in self.opt.param_groups:
                pg["lr"] = scale * self.lr
        self.global_step += 1
        return losses, total_norm
