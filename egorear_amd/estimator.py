"""Drop-in operator API of the hot path: the three estimator classes the reference's
Lightning wrappers construct from YAML (pose_estimation/models/estimator/__init__.py:3-5).

Same class names, constructor keywords (unknown keys swallowed), `forward` signatures,
attributes read by the wrappers and `state_dict` keys as the reference
 — but `forward` runs hand-written HIP kernels for gfx950 through the
C-ABI library (egorear_amd.hip).  There is no PyTorch/CPU fallback: a missing
library or a non-GPU tensor raises.

Modes.  eval(): inference under `torch.no_grad()` (engine.py).  train(): `EgoPoseFormerHeatmap`, `EgoPoseFormerHeatmapMVFEX`
and `EgoPoseFormerMVFEX` run the training-mode forward on the HIP kernels and return outputs attached to ONE autograd node
whose backward is the hand-written reverse pass (train.py) - the three stages of the reference's schedule train through their
unchanged Lightning wrappers.  What is refused, loudly: a module in eval() mode with autograd enabled (the reference would
differentiate through eval-mode BatchNorm there; no shipped flow does), and the two sub-entry points no shipped wrapper calls
under autograd: `EgoPoseFormerPose3D.forward` on its own and `EgoPoseFormerHeatmap.forward_backbone` (the parent modules
train; these two are inference-only, torch.no_grad()).
"""
from __future__ import annotations

import copy
import os

import torch
import torch.nn as nn

from .tree import (FFNParams, HeadLayerParams, JointTransformerLayer, ResnetBackbone, stack)

_PKG_CALIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "calib", "ego4view")


def _no_dynamo(fn):
    """run.py wraps `model.network` in torch.compile whenever the YAML says `compile: True` (run.py:7-9; every shipped
    config does).  The forward here is a sequence of C-ABI launches, which Dynamo cannot trace: keep it out of the
    tracer so the compiled wrapper simply calls through (inference gains nothing from Inductor on this path)."""
    try:
        return torch.compiler.disable(fn)
    except AttributeError:  # very old torch
        return fn


def _require_inference(module: nn.Module):
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        raise NotImplementedError(
            "egorear_amd: this entry point is inference-only - call it under torch.no_grad() (eval mode), or put the parent "
            "estimator in train() mode: EgoPoseFormerHeatmap / EgoPoseFormerHeatmapMVFEX / EgoPoseFormerMVFEX train through "
            "their own forward (egorear_amd.train)")


class EgoPoseFormerHeatmap(nn.Module):
    """reference: models/estimator/egoposeformer_heatmap.py:9-44."""

    def __init__(self, encoder_cfg, num_heatmap, detach_heatmap_feat_init=False, **kwargs):
        super().__init__()
        self.num_heatmap = num_heatmap
        self.detach_heatmap_feat_init = detach_heatmap_feat_init
        self.encoder = ResnetBackbone(**encoder_cfg)
        self.conv_heatmap = nn.Conv2d(self.encoder.get_output_channel(), num_heatmap, 1)

    @_no_dynamo
    def forward_backbone(self, img, return_feat=False):
        from . import engine
        _require_inference(self)
        return engine.heatmap_backbone_api(self, img)

    @_no_dynamo
    def forward(self, img, return_feat=False):
        from . import engine
        if self.training:   # network.train() in PoseHeatmapLightningModel.training_step (pl_wrappers/egoposeformer/heatmap.py:95)
            from . import train
            return train.heatmap_training_forward(self, img, return_feat)
        _require_inference(self)
        return engine.heatmap_forward_api(self, img, return_feat)


class HeatmapMVF(nn.Module):
    """MVFEx refiner with joint-query adaptation; reference heatmap_mvf_ex.py:442-584."""

    def __init__(self, input_dims, embed_dims, num_former_layers, image_size, feat_down_stride, detach_heatmap_feat,
                 mvf_transformer_cfg, heatmap_threshold, num_views, num_heatmap, joint_query_adaptation=False,
                 joint_query_adaptation_multi_view=False, joint_query_only=False, use_1by1_conv=False):
        super().__init__()
        if not joint_query_adaptation or joint_query_adaptation_multi_view or joint_query_only or use_1by1_conv:
            raise NotImplementedError("only joint_query_adaptation=True without use_1by1_conv is configured (SURVEY.md §2)")
        if num_former_layers != 1:
            raise NotImplementedError("mvf num_former_layers != 1 is not used by any shipped config")
        if input_dims != 128:
            raise NotImplementedError("mvf input_dims must be 128")
        self.num_heatmap, self.num_views = num_heatmap, num_views
        self.heatmap_threshold = heatmap_threshold
        self.detach_heatmap_feat = detach_heatmap_feat
        self.joint_query_adaptation = True
        self.use_1by1_conv = False
        self.embed_dims = embed_dims
        self.feat_shape = (image_size[0] // feat_down_stride, image_size[1] // feat_down_stride)
        hw = self.feat_shape[0] * self.feat_shape[1]

        self.heatmap_proj = stack(("linear", hw, embed_dims), "relu", ("linear", embed_dims, embed_dims))
        self.fc_bfb = nn.Linear(512, embed_dims)
        self.fc_query = stack(("linear", embed_dims, embed_dims), "relu")
        self.joint_query_embed = nn.Embedding(num_heatmap, embed_dims)
        self.frame_feat_multi_view_proj = nn.Conv2d(input_dims, embed_dims, 1, 1, 0)
        self.frame_feat_multi_view_pos_embed = nn.Parameter(torch.zeros(1, num_views, hw, embed_dims))
        self.frame_feat_proj_layers = stack(("conv", input_dims, input_dims * 2, 1), "relu",
                                            ("conv", input_dims * 2, input_dims * 4, 3, 2), "relu",
                                            ("conv", input_dims * 4, input_dims, 1), "relu")
        cfg = copy.deepcopy(mvf_transformer_cfg)
        cfg.update({"num_views": num_views, "embed_dims": embed_dims, "feat_shape": self.feat_shape})
        self.transformer_layers = nn.ModuleList([JointTransformerLayer(**cfg)])
        self.post_norm = nn.ModuleList([nn.LayerNorm(embed_dims)])
        self.head_layers = nn.ModuleList([HeadLayerParams(input_dims=num_heatmap, output_dims=input_dims)])
        self.frame_feat_refined_proj_layers = nn.ModuleList([
            stack(("conv", input_dims, input_dims, 1), "relu", "up2", ("conv", input_dims, input_dims, 1), "relu")])
        self.conv_heatmap_layers = nn.ModuleList([
            stack(("conv", input_dims, input_dims * 2, 3, 2), "relu", ("conv", input_dims * 2, input_dims * 2, 1), "relu",
                  "up2", ("conv", input_dims * 2, input_dims, 1), "relu", ("conv", input_dims, num_heatmap, 1))])


def _init_heatmap_stack(c: int, num_heatmap: int) -> nn.Sequential:
    return stack(("conv", c, c, 1), "relu", ("conv", c, 2 * c, 3, 2), "relu", ("conv", 2 * c, 2 * c, 1), "relu",
                 "up2", ("conv", 2 * c, c, 1), "relu", ("conv", c, num_heatmap, 1))


class EgoPoseFormerHeatmapMVFEX(nn.Module):
    """reference: models/estimator/egoposeformer_heatmap_mvf_ex.py:27-437 (4-view branch)."""

    def __init__(self, num_views, image_size, num_heatmap, feat_down_stride, heatmap_threshold, encoder_cfg, mvf_cfg,
                 camera_model, full_training=False, detach_heatmap_feat=False, detach_heatmap_feat_init=False,
                 use_pred_heatmap_init=False, no_detach_feat_init=False, **kwargs):
        super().__init__()
        if num_views != 4:
            raise NotImplementedError("2/3-view variants are out of scope (SURVEY.md §2, App. B-15)")
        self.num_views, self.num_heatmap = num_views, num_heatmap
        self.heatmap_threshold = heatmap_threshold
        self.camera_model = camera_model
        self.full_training = full_training
        self.detach_heatmap_feat = detach_heatmap_feat
        self.detach_heatmap_feat_init = detach_heatmap_feat_init
        self.use_pred_heatmap_init = use_pred_heatmap_init
        self.no_detach_feat_init = no_detach_feat_init
        self.image_size = tuple(image_size)

        mvf_cfg = copy.deepcopy(mvf_cfg)
        mvf_cfg.update({"num_views": num_views, "num_heatmap": num_heatmap, "heatmap_threshold": heatmap_threshold,
                        "image_size": image_size, "feat_down_stride": feat_down_stride,
                        "detach_heatmap_feat": detach_heatmap_feat})
        self.heatmap_estimator_stereo_front = EgoPoseFormerHeatmap(encoder_cfg, num_heatmap, detach_heatmap_feat_init)
        self.heatmap_estimator_stereo_back = EgoPoseFormerHeatmap(encoder_cfg, num_heatmap, detach_heatmap_feat_init)
        self.heatmap_refiner_front_left = HeatmapMVF(**mvf_cfg)
        self.heatmap_refiner_front_right = HeatmapMVF(**mvf_cfg)
        self.heatmap_refiner_back_left = HeatmapMVF(**mvf_cfg)
        self.heatmap_refiner_back_right = HeatmapMVF(**mvf_cfg)
        self.use_1by1_conv = False
        self.conv_heatmap_layers_stereo_front = _init_heatmap_stack(128, num_heatmap)
        self.conv_heatmap_layers_stereo_back = _init_heatmap_stack(128, num_heatmap)

    def refiners(self):
        return [self.heatmap_refiner_front_left, self.heatmap_refiner_front_right,
                self.heatmap_refiner_back_left, self.heatmap_refiner_back_right]

    def get_anchors_2d_from_hm(self, heatmap):
        from . import engine
        return engine.anchors_from_heatmap_api(self, heatmap)

    @_no_dynamo
    def forward(self, img, heatmap_for_anchor=None):
        from . import engine
        if self.training:   # PoseHeatmapMVFEXLightningModel.training_step (pl_wrappers/egoposeformer/heatmap_mvf_ex.py:105)
            from . import train
            return train.heatmap_mvfex_training_forward(self, img, heatmap_for_anchor)
        _require_inference(self)
        return engine.heatmap_mvfex_forward_api(self, img, heatmap_for_anchor)


class EgoPoseFormerPose3D(nn.Module):
    """2D-to-3D lifting head; reference egoposeformer_mvf_ex.py:62-452 (conv-MLP proposal branch)."""

    def __init__(self, num_views, image_size, use_pred_heatmap_init, num_joints, input_dims, embed_dims, mlp_dims,
                 mlp_dropout, num_mlp_layers, transformer_cfg, num_former_layers, num_pred_mlp_layers, camera_model,
                 feat_down_stride, coor_norm_max, coor_norm_min, conv_heatmap_dim_init, norm_mlp_pred=False,
                 use_mlp_avgpool=True, use_mlp_heatmap=False, camera_calib_file_dir_path=None, **kwargs):
        super().__init__()
        if use_mlp_avgpool or use_mlp_heatmap or norm_mlp_pred:
            raise NotImplementedError("use_mlp_avgpool / use_mlp_heatmap / norm_mlp_pred are not enabled by any shipped config")
        if camera_model not in ("ego4view_syn", "ego4view_rw") or num_views != 4:
            raise NotImplementedError("stereo (2-view) camera models are out of scope (SURVEY.md §2)")
        if mlp_dropout != 0.0 or num_pred_mlp_layers != 2 or input_dims != 128:
            raise NotImplementedError("unsupported pose3d_cfg for the built hot path")
        self.num_views, self.num_joints, self.embed_dims = num_views, num_joints, embed_dims
        self.feat_down_stride = feat_down_stride
        self.feat_shape = (image_size[0] // feat_down_stride, image_size[1] // feat_down_stride)
        self.image_size = image_size
        self.camera_model = camera_model
        self.use_pred_heatmap_init = use_pred_heatmap_init
        self.use_mlp_avgpool, self.use_mlp_heatmap, self.norm_mlp_pred = False, False, False

        from .camera import FishEyeCameraCalibratedModel
        calib = camera_calib_file_dir_path
        if calib is None or not os.path.isdir(calib):
            calib = _PKG_CALIB  # same four data files, packaged (SURVEY.md §8b "Config compatibility")
        self.camera_front_left_model = FishEyeCameraCalibratedModel(camera_model, calib, "camera_front_left")
        self.camera_front_right_model = FishEyeCameraCalibratedModel(camera_model, calib, "camera_front_right")
        self.camera_back_left_model = FishEyeCameraCalibratedModel(camera_model, calib, "camera_back_left")
        self.camera_back_right_model = FishEyeCameraCalibratedModel(camera_model, calib, "camera_back_right")

        self.feat_proj = nn.Conv2d(input_dims, embed_dims, 1, 1, 0)
        cfg = copy.deepcopy(transformer_cfg)
        cfg.update({"num_views": num_views, "embed_dims": embed_dims, "feat_shape": self.feat_shape})
        self.layers = nn.ModuleList([JointTransformerLayer(**copy.deepcopy(cfg)) for _ in range(num_former_layers)])
        self.query_gen_mlp = nn.Sequential(nn.Linear(4, embed_dims), nn.ReLU(inplace=True), nn.Linear(embed_dims, embed_dims),
                                           nn.ReLU(inplace=True), nn.Linear(embed_dims, embed_dims))
        h = input_dims // 2
        self.conv_frame_feat = stack(("conv", input_dims, h, 1), "relu", ("conv", h, input_dims, 3, 2), "relu", ("maxpool", 2),
                                     ("conv", input_dims, h, 1), "relu", ("conv", h, input_dims, 3, 2), "relu")
        mlp, in_dims = [], num_views * 128 * 8 * 8
        for _ in range(num_mlp_layers):
            mlp.append(nn.Sequential(nn.Linear(in_dims, in_dims // 16), nn.GELU(), nn.Dropout(mlp_dropout)))
            in_dims //= 16
        mlp.append(nn.Linear(in_dims, 3 * num_joints))
        self.mlp_pred = nn.Sequential(*mlp)
        self.reg_mlp = nn.ModuleList([nn.Sequential(nn.Linear(embed_dims, embed_dims), nn.GELU(), nn.Linear(embed_dims, 3))
                                      for _ in range(num_former_layers)])
        self.post_norm = nn.ModuleList([nn.LayerNorm(embed_dims) for _ in range(num_former_layers)])

    def cameras(self):
        return [self.camera_front_left_model, self.camera_front_right_model,
                self.camera_back_left_model, self.camera_back_right_model]

    @_no_dynamo
    def forward(self, frame_feats_init, frame_feats_final, heatmap, coord_trans_mat=None, origin_3d=None):
        from . import engine
        _require_inference(self)
        return engine.pose3d_forward_api(self, frame_feats_init, frame_feats_final, coord_trans_mat)


class EgoPoseFormerMVFEX(nn.Module):
    """reference: models/estimator/egoposeformer_mvf_ex.py:22-59."""

    def __init__(self, num_views, image_size, camera_model, heatmap_mvf_cfg, pose3d_cfg, **kwargs):
        super().__init__()
        heatmap_mvf_cfg = copy.deepcopy(heatmap_mvf_cfg)
        pose3d_cfg = copy.deepcopy(pose3d_cfg)
        heatmap_mvf_cfg.update({"num_views": num_views, "image_size": image_size, "camera_model": camera_model})
        self.heatmap_estimator = EgoPoseFormerHeatmapMVFEX(**heatmap_mvf_cfg)
        self.use_pred_heatmap_init = self.heatmap_estimator.use_pred_heatmap_init
        pose3d_cfg.update({"num_views": num_views, "image_size": image_size,
                           "use_pred_heatmap_init": self.use_pred_heatmap_init, "camera_model": camera_model})
        self.pose3d_estimator = EgoPoseFormerPose3D(**pose3d_cfg)

    @_no_dynamo
    def forward(self, img, coord_trans_mat=None, origin_3d=None):
        from . import engine
        if self.training:
            # network.train() (pl_wrappers/egoposeformer/pose_3d_mvf_ex.py:115): BatchNorm batch statistics, and under
            # autograd the outputs carry a grad_fn whose backward is the hand-written reverse pass (egorear_amd.train)
            from . import train
            return train.mvfex_training_forward(self, img, coord_trans_mat)
        _require_inference(self)
        return engine.mvfex_forward_api(self, img, coord_trans_mat)
