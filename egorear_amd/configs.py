"""Model hyper-parameter presets of the hot path's shipped configurations.

The reference builds its models from `model.init_args.model_cfg` of its YAML files
(configs/ego4view_{syn,rw}_{heatmap_stereo_*,heatmap_mvfex-n1_jqa,pose3d}.yaml).  Those
files do not travel to the GPU box, so the same values are available here as plain
dicts; `load_model_cfg` reads them from an unchanged reference YAML when one is given.
"""
from __future__ import annotations

import copy

_ENCODER = {
    "resnet_cfg": {"model_name": "resnet18", "out_stride": 4, "use_imagenet_pretrain": False},
    "neck_cfg": {"in_channels": [64, 128, 256, 512], "out_channels": 128},
}

_ATTN = {
    "cross_attn_cfg": {"num_heads": 4, "batch_first": True},
    "spatial_attn_cfg": {"num_heads": 4, "batch_first": True},
    "ffn_cfg": {"feedforward_dims": 512, "num_fcs": 2, "ffn_drop": 0.0},
}


def heatmap_cfg() -> dict:
    """configs/ego4view_*_heatmap_stereo_{front,back}.yaml: model_cfg."""
    return {"num_heatmap": 15, "encoder_cfg": copy.deepcopy(_ENCODER)}


def _mvf_cfg() -> dict:
    return {"input_dims": 128, "embed_dims": 256, "num_former_layers": 1, "joint_query_adaptation": True,
            "mvf_transformer_cfg": copy.deepcopy(_ATTN)}


def heatmap_mvfex_cfg(camera_model: str = "ego4view_syn") -> dict:
    """configs/ego4view_*_heatmap_mvfex-n1_jqa.yaml: model_cfg (inference-relevant keys)."""
    return {"num_views": 4, "image_size": [256, 256], "num_heatmap": 15, "feat_down_stride": 4,
            "heatmap_threshold": 0.5, "camera_model": camera_model, "encoder_cfg": copy.deepcopy(_ENCODER),
            "mvf_cfg": _mvf_cfg(),
            "num_joints": 16, "anchor_2d_update": True}   # in the YAMLs, swallowed by the constructor's **kwargs


def pose3d_cfg(camera_model: str = "ego4view_syn") -> dict:
    """configs/ego4view_{syn,rw}_pose3d.yaml: model_cfg."""
    return {
        "num_views": 4, "image_size": [256, 256], "camera_model": camera_model,
        "pose3d_cfg": {
            "num_joints": 16, "input_dims": 128, "embed_dims": 128, "mlp_dims": 1024, "mlp_dropout": 0.0,
            "num_mlp_layers": 2, "num_former_layers": 3, "num_pred_mlp_layers": 2, "feat_down_stride": 4,
            "norm_mlp_pred": False, "coor_norm_max": None, "coor_norm_min": None, "conv_heatmap_dim_init": 32,
            "use_mlp_avgpool": False, "use_mlp_heatmap": False,
            "camera_calib_file_dir_path": "./pose_estimation/utils/camera_calib_file/ego4view",
            "transformer_cfg": copy.deepcopy(_ATTN),
        },
        "heatmap_mvf_cfg": {
            "num_heatmap": 15, "feat_down_stride": 4, "heatmap_threshold": 0.5, "full_training": True,
            "detach_heatmap_feat": True, "detach_heatmap_feat_init": True, "use_pred_heatmap_init": True,
            "encoder_cfg": copy.deepcopy(_ENCODER), "mvf_cfg": _mvf_cfg(),
        },
    }


def load_model_cfg(yaml_path: str) -> dict:
    """model_cfg of an unchanged reference YAML, exactly as written there (`use_imagenet_pretrain: True` included: the trunk
    then loads the ImageNet weights or says loudly that it could not - tree.load_imagenet_resnet18)."""
    import yaml
    with open(yaml_path) as f:
        return yaml.safe_load(f)["model"]["init_args"]["model_cfg"]


def set_imagenet_pretrain(cfg: dict, value: bool) -> dict:
    """Set every `use_imagenet_pretrain` of a model_cfg (the presets above say False: benchmarks and tests load seeded weights
    right after construction; the shipped YAMLs say True)."""
    for k, v in cfg.items():
        if k == "use_imagenet_pretrain":
            cfg[k] = value
        elif isinstance(v, dict):
            set_imagenet_pretrain(v, value)
    return cfg
