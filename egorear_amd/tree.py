"""Parameter containers: the `state_dict` contract of the reference's hot path.

The reference's checkpoints are loaded `strict=True` into sub-modules
(pl_wrappers/egoposeformer/pose_3d_mvf_ex.py:103-112), so every key and shape
below must equal the reference's (SURVEY.md §8b lists the key families; the
reference's constructors are models/backbones/resnet.py:6-137,
models/estimator/egoposeformer_heatmap.py:9-23,
models/estimator/egoposeformer_heatmap_mvf_ex.py:27-126,442-584,767-970,
models/estimator/egoposeformer_mvf_ex.py:62-265,455-531,
models/utils/deform_attn.py:25-65, models/utils/transformer.py:8-81).

These classes only *hold* parameters (plus the structural integer buffers the
reference registers).  None of them computes anything: the compute lives in the
HIP kernels reached through egorear_amd.hip, driven by egorear_amd.estimator.
Stacks are described by a small op list (`stack(...)`) that builds an
nn.Sequential with the reference's child indices, so the same description also
drives the HIP execution (egorear_amd.engine.run_stack walks the same op lists).
"""
from __future__ import annotations

import copy
from typing import Sequence

import torch
import torch.nn as nn


# --------------------------------------------------------------------------- stack DSL

def stack(*ops) -> nn.Sequential:
    """ops: ("conv", cin, cout, k[, stride]) | "relu" | "up2" | ("maxpool", k) | ("linear", cin, cout) | "gelu" | ("drop", p)."""
    mods = []
    for op in ops:
        if op == "relu":
            mods.append(nn.ReLU(inplace=False))
        elif op == "gelu":
            mods.append(nn.GELU())
        elif op == "up2":
            mods.append(nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True))
        elif op[0] == "conv":
            cin, cout, k = op[1], op[2], op[3]
            s = op[4] if len(op) > 4 else 1
            mods.append(nn.Conv2d(cin, cout, k, s, k // 2))
        elif op[0] == "maxpool":
            mods.append(nn.MaxPool2d(op[1]))
        elif op[0] == "linear":
            mods.append(nn.Linear(op[1], op[2]))
        elif op[0] == "drop":
            mods.append(nn.Dropout(op[1]))
        else:
            raise ValueError(f"unknown stack op {op!r}")
    return nn.Sequential(*mods)


# --------------------------------------------------------------------------- ResNet-18 trunk + FPN

class BasicBlock(nn.Module):
    """torchvision==0.19 BasicBlock parameter layout (conv1,bn1,conv2,bn2[,downsample.{0,1}])."""

    def __init__(self, cin: int, cout: int, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.stride = stride
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))
        else:
            self.downsample = None


def _res_stage(cin: int, cout: int, stride: int) -> nn.Sequential:
    return nn.Sequential(BasicBlock(cin, cout, stride), BasicBlock(cout, cout, 1))


class ResNet18Trunk(nn.Module):
    """Children named as the reference splits torchvision's resnet18 (resnet.py:14-21)."""

    def __init__(self, model_name: str = "resnet18", use_imagenet_pretrain: bool = False, out_stride: int = 4):
        super().__init__()
        if model_name != "resnet18":
            raise NotImplementedError("model type [%s] is invalid" % model_name)
        if out_stride != 4:
            raise NotImplementedError("only out_stride=4 is on the hot path (SURVEY.md §2)")
        self.layer_s2 = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True))
        self.layer_s4 = nn.Sequential(nn.MaxPool2d(3, 2, 1), _res_stage(64, 64, 1))
        self.layer_s8 = _res_stage(64, 128, 2)
        self.layer_s16 = _res_stage(128, 256, 2)
        self.layer_s32 = _res_stage(256, 512, 2)
        self.out_stride = out_stride
        self.use_imagenet_pretrain = bool(use_imagenet_pretrain)
        if use_imagenet_pretrain:   # resnet.py:31-39: torchvision.models.resnet18(weights='DEFAULT')
            load_imagenet_resnet18(self)


_TV_PREFIX = (("conv1.", "layer_s2.0."), ("bn1.", "layer_s2.1."), ("layer1.", "layer_s4.1."), ("layer2.", "layer_s8."),
              ("layer3.", "layer_s16."), ("layer4.", "layer_s32."))


def torchvision_resnet18_to_trunk(tv_state: dict) -> dict:
    """torchvision resnet18 `state_dict` keys -> the reference's split of its children (resnet.py:14-21); `fc.*` is dropped
    (the reference discards avgpool / fc)."""
    out = {}
    for k, v in tv_state.items():
        for src, dst in _TV_PREFIX:
            if k.startswith(src):
                out[dst + k[len(src):]] = v
                break
    return out


def load_imagenet_resnet18(trunk: "ResNet18Trunk") -> str:
    """`use_imagenet_pretrain: True` (all 12 shipped YAMLs; reference resnet.py:31-39).  Sources, in order: the file named by
    EGR_RESNET18_WEIGHTS (a torchvision resnet18 state_dict, for boxes without network), torchvision's own `weights='DEFAULT'`
    (its cache or a download).  If neither is available the trunk keeps its random initialisation and this says so LOUDLY - or
    raises when EGR_STRICT_PRETRAIN=1 - because a stage-1 fit would otherwise silently start from other weights than the
    reference's.  Returns the source used ("file", "torchvision" or "none")."""
    import os
    import warnings
    path = os.environ.get("EGR_RESNET18_WEIGHTS")
    tv_state, source, why = None, "none", ""
    if path:
        tv_state, source = torch.load(path, map_location="cpu"), "file"
        if isinstance(tv_state, dict) and "state_dict" in tv_state:
            tv_state = tv_state["state_dict"]
    else:
        try:
            import torchvision
            tv_state, source = torchvision.models.resnet18(weights="DEFAULT").state_dict(), "torchvision"
        except Exception as exc:   # torchvision missing, or no cached weights and no network
            why = f"{type(exc).__name__}: {exc}"
    if tv_state is None:
        msg = ("egorear_amd: use_imagenet_pretrain=True but the ImageNet ResNet-18 weights are not available here (" + why +
               "); the trunk keeps its RANDOM initialisation, unlike the reference (models/backbones/resnet.py:31-39).  Point "
               "EGR_RESNET18_WEIGHTS at a torchvision resnet18 state_dict file, or load a checkpoint afterwards.")
        if os.environ.get("EGR_STRICT_PRETRAIN") == "1":
            raise RuntimeError(msg)
        warnings.warn(msg, RuntimeWarning, stacklevel=3)
        return "none"
    mapped = torchvision_resnet18_to_trunk(tv_state)
    trunk.load_state_dict(mapped, strict=True)
    return source


class FPNNeck(nn.Module):
    def __init__(self, in_channels: Sequence[int], out_channels: int, with_relu: bool = True):
        super().__init__()
        if not with_relu:
            raise NotImplementedError("with_relu=False is not used by any shipped config")
        self.in_channels = list(in_channels)
        self.out_channels = out_channels
        self.lateral_convs = nn.ModuleList(stack(("conv", c, out_channels, 1), "relu") for c in in_channels)
        self.fuse_convs = nn.ModuleList(stack(("conv", 2 * out_channels, out_channels, 1), "relu") for _ in in_channels[1:])
        self.fpn_convs = nn.ModuleList(stack(("conv", out_channels, out_channels, 3), "relu") for _ in in_channels[1:])


class ResnetBackbone(nn.Module):
    def __init__(self, resnet_cfg: dict, neck_cfg: dict):
        super().__init__()
        self.backbone = ResNet18Trunk(**resnet_cfg)
        self.neck = FPNNeck(**neck_cfg)

    def get_output_channel(self) -> int:
        return self.neck.out_channels


# --------------------------------------------------------------------------- attention / FFN containers

class DeformAttnParams(nn.Module):
    """MSDeformAttn parameters (deform_attn.py:56-65) + the two buffers its subclasses add
    (egoposeformer_heatmap_mvf_ex.py:776-777, egoposeformer_mvf_ex.py:464-465)."""

    def __init__(self, embed_dim: int, num_heads: int, feat_shape, n_points: int = 16, **_unused):
        super().__init__()
        if embed_dim % num_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(embed_dim, num_heads))
        self.d_model, self.n_heads, self.n_points, self.n_levels = embed_dim, num_heads, n_points, 1
        self.sampling_offsets = nn.Linear(embed_dim, num_heads * n_points * 2)
        self.attention_weights = nn.Linear(embed_dim, num_heads * n_points)
        self.value_proj = nn.Linear(embed_dim, embed_dim)
        self.output_proj = nn.Linear(embed_dim, embed_dim)
        self.register_buffer("spatial_shapes", torch.tensor([[feat_shape[0], feat_shape[1]]], dtype=torch.long))
        self.register_buffer("start_index", torch.tensor([0], dtype=torch.long))


class JointMHAParams(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, batch_first: bool = True, **_unused):
        super().__init__()
        assert batch_first
        self.num_heads = num_heads
        self.head_dims = embed_dim // num_heads
        self.scale = self.head_dims ** -0.5
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)


class FFNParams(nn.Module):
    def __init__(self, embed_dims: int, feedforward_dims: int, num_fcs: int, ffn_drop: float):
        super().__init__()
        if num_fcs != 2 or ffn_drop != 0.0:
            raise NotImplementedError("only num_fcs=2, ffn_drop=0.0 are configured by the reference YAMLs")
        self.layers = nn.Sequential(
            nn.Sequential(nn.Linear(embed_dims, feedforward_dims), nn.GELU(), nn.Dropout(ffn_drop)),
            nn.Linear(feedforward_dims, embed_dims),
            nn.Dropout(ffn_drop),
        )


class JointTransformerLayer(nn.Module):
    """Shared container for MultiViewTransformerLayer (heatmap_mvf_ex.py:820-859, C=256, J=15)
    and EgoPoseFormerTransformerLayer (egoposeformer_mvf_ex.py:501-531, C=128, J=16)."""

    def __init__(self, num_views, embed_dims, cross_attn_cfg, spatial_attn_cfg, ffn_cfg, feat_shape, use_normal_cross_attn=False):
        super().__init__()
        if use_normal_cross_attn:
            raise NotImplementedError("use_normal_cross_attn is not enabled by any shipped config (SURVEY.md §2)")
        self.embed_dims = embed_dims
        self.num_views = num_views
        self.cross_attn = DeformAttnParams(embed_dim=embed_dims, feat_shape=feat_shape, **copy.deepcopy(cross_attn_cfg))
        self.fuse_mlp = nn.Linear(num_views * embed_dims, embed_dims)
        self.norm_cross = nn.LayerNorm(embed_dims)
        self.spatial_attn = JointMHAParams(embed_dim=embed_dims, **copy.deepcopy(spatial_attn_cfg))
        self.norm_spatial = nn.LayerNorm(embed_dims)
        self.ffn = FFNParams(embed_dims=embed_dims, **copy.deepcopy(ffn_cfg))
        self.norm_ffn = nn.LayerNorm(embed_dims)


class HeadLayerParams(nn.Module):
    """TransformerHeadLayer, 128-d branch only (heatmap_mvf_ex.py:947-954)."""

    def __init__(self, input_dims: int, output_dims: int):
        super().__init__()
        if output_dims != 128:
            raise NotImplementedError("only the output_dims==128 head is configured (SURVEY.md §2)")
        self.head = stack(("conv", input_dims, output_dims // 2, 1), "relu", "up2", ("conv", output_dims // 2, output_dims, 1), "relu")
