"""Fisheye (Scaramuzza) camera calibration holder for the 3D->2D anchor reprojection.

reference: pose_estimation/utils/camera_models.py:14-51 (constructor; the projection
itself, :53-104, runs in the HIP kernel `egr_fisheye_project_f32`).

Unlike the reference this class does not create device tensors at construction
(the reference hard-codes device="cuda"): it keeps the calibration as
fp32-rounded host floats, which the engine uploads once per device.
"""
from __future__ import annotations

import json
import os

import numpy as np

MAX_POLY = 12

# utils/camera_models.py:29-40: constant rigid offsets (cm) used in ego4view_syn mode
_SYN_OFFSETS = {
    "camera_front_left": (6.0, 0.0, 0.0),
    "camera_front_right": (-6.0, 0.0, 0.0),
    "camera_back_left": (-6.0, 37.0, 0.0),
    "camera_back_right": (6.0, 37.0, 0.0),
}


class FishEyeCameraCalibratedModel:
    def __init__(self, camera_model: str, camera_calib_file_dir_path: str, camera_name: str):
        if not (camera_model.startswith("ego4view_syn") or camera_model.startswith("ego4view_rw")):
            raise ValueError("Unknown camera model !")
        self.camera_model = camera_model
        self.camera_name = camera_name
        with open(os.path.join(camera_calib_file_dir_path, "{}.json".format(camera_name))) as f:
            d = json.load(f)
        self.image_size = tuple(int(v) for v in d["size"])  # (H, W) pixel frame, 872x872
        self.image_center = tuple(float(np.float32(v)) for v in d["image_center"])
        self.fisheye_inv_polynomial = [float(np.float32(v)) for v in d["polynomialW2C"]]
        if len(self.fisheye_inv_polynomial) > MAX_POLY:
            raise ValueError("polynomialW2C longer than the kernel's MAX_POLY")
        self.offset = _SYN_OFFSETS[camera_name]
        self.flip_xy = camera_name in ("camera_back_left", "camera_back_right")
        self.m2cm, self.cm2m = 100.0, 0.01

    def packed(self) -> np.ndarray:
        """fp32 record consumed by the kernel: [npoly, cx, cy, W, H, poly[0..MAX_POLY)]."""
        rec = np.zeros(5 + MAX_POLY, dtype=np.float32)
        rec[0] = len(self.fisheye_inv_polynomial)
        rec[1], rec[2] = self.image_center
        rec[3], rec[4] = self.image_size[1], self.image_size[0]
        rec[5:5 + len(self.fisheye_inv_polynomial)] = self.fisheye_inv_polynomial
        return rec
