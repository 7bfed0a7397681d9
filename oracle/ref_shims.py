"""sys.modules shims that let the *real* reference (/root/reference) be imported in the
build container for golden generation.  TEST INFRASTRUCTURE ONLY; build-container only.

The reference's hot path needs four third-party packages that are not installed here
(SURVEY.md §8c): loguru, timm, torchvision, mmcv.  The stand-ins below are written by
this project (nothing is taken from /root/reference):

  * loguru.logger               -> no-op logger
  * timm.models.layers.to_2tuple-> only used by a dead class
  * torchvision.models.resnet18 -> the published BasicBlock ResNet-18 with torchvision's
                                   child order (conv1,bn1,relu,maxpool,layer1..4,avgpool,fc)
  * mmcv.ops.multi_scale_deform_attn.MultiScaleDeformableAttnFunction
                                -> forward only, formulated with F.grid_sample
                                   (bilinear, zeros padding, align_corners=False), i.e. a
                                   *different* formulation from oracle.egorear_oracle.msda_core
                                   so that the golden comparison cross-checks the two.

The reference also hard-codes device="cuda" for its calibration tensors (F8); while the
shims are installed, torch.tensor drops that argument when no GPU is present.
"""
from __future__ import annotations

import contextlib
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = "/root/reference"


# ---- torchvision.models.resnet18 -------------------------------------------------

class _Block(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.bn2(self.conv2(self.relu(self.bn1(self.conv1(x)))))
        return self.relu(y + idt)


class _ResNet18(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(_Block(64, 64, 1), _Block(64, 64, 1))
        self.layer2 = nn.Sequential(_Block(64, 128, 2), _Block(128, 128, 1))
        self.layer3 = nn.Sequential(_Block(128, 256, 2), _Block(256, 256, 1))
        self.layer4 = nn.Sequential(_Block(256, 512, 2), _Block(512, 512, 1))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)


def _resnet18(weights=None, **_):
    return _ResNet18()  # weights='DEFAULT' would need a download; goldens load synthetic weights


# ---- mmcv MSDA forward via grid_sample -------------------------------------------

class _MSDAFunction:
    @staticmethod
    def apply(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step):
        N, _, nh, D = value.shape
        _, Lq, _, L, P, _ = sampling_locations.shape
        shapes = [(int(h), int(w)) for h, w in spatial_shapes.tolist()]
        values = value.split([h * w for h, w in shapes], dim=1)
        grids = 2 * sampling_locations - 1
        sampled = []
        for lvl, (h, w) in enumerate(shapes):
            v = values[lvl].flatten(2).transpose(1, 2).reshape(N * nh, D, h, w)
            g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)
            sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))
        aw = attention_weights.transpose(1, 2).reshape(N * nh, 1, Lq, L * P)
        out = (torch.stack(sampled, dim=-2).flatten(-2) * aw).sum(-1).view(N, nh * D, Lq)
        return out.transpose(1, 2).contiguous()


class _Logger:
    def __getattr__(self, _name):
        return lambda *a, **k: None


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


@contextlib.contextmanager
def reference_importable():
    """Context in which `import pose_estimation.models.estimator` resolves to the real
    reference source.  Restores sys.modules / sys.path / cwd / torch.tensor afterwards."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference tree not present (golden generation is build-container only)")
    saved = {k: sys.modules.get(k) for k in (
        "loguru", "timm", "timm.models", "timm.models.layers", "torchvision", "torchvision.models",
        "mmcv", "mmcv.ops", "mmcv.ops.multi_scale_deform_attn")}
    tv_models = _module("torchvision.models", resnet18=_resnet18)
    shims = {
        "loguru": _module("loguru", logger=_Logger()),
        "timm": _module("timm"),
        "timm.models": _module("timm.models"),
        "timm.models.layers": _module("timm.models.layers", to_2tuple=lambda x: x if isinstance(x, (tuple, list)) else (x, x)),
        "torchvision": _module("torchvision", models=tv_models),
        "torchvision.models": tv_models,
        "mmcv": _module("mmcv"),
        "mmcv.ops": _module("mmcv.ops"),
        "mmcv.ops.multi_scale_deform_attn": _module("mmcv.ops.multi_scale_deform_attn", MultiScaleDeformableAttnFunction=_MSDAFunction),
    }
    sys.modules.update(shims)
    real_tensor = torch.tensor

    def _tensor(*a, **k):
        if k.get("device") == "cuda" and not torch.cuda.is_available():
            k.pop("device")
        return real_tensor(*a, **k)

    torch.tensor = _tensor
    cwd = os.getcwd()
    os.chdir(REFERENCE_ROOT)  # relative camera_calib_file_dir_path
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        yield
    finally:
        sys.path.remove(REFERENCE_ROOT)
        os.chdir(cwd)
        torch.tensor = real_tensor
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        for k in [k for k in sys.modules if k == "pose_estimation" or k.startswith("pose_estimation.")]:
            sys.modules.pop(k, None)
