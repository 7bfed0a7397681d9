"""tests/golden/gt_heatmap.npz from the REAL reference's generate_heatmap.generate_target (module-level imports of cv2 /
natsort / loguru, unused by the function, are stubbed).  Build-container only; TEST INFRASTRUCTURE.
    python -m oracle.make_golden_heatmap_gt"""
import os
import sys
import types

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from egorear_amd import synth  # noqa: E402


def joints_case():
    """(4, 16, 2) pixel coordinates in the 872 frame incl. borders, outside and half-integer cell boundaries."""
    u = synth.uniform01("gt_heatmap.joints", 3, 4 * 16 * 2).reshape(4, 16, 2).astype(np.float64)
    j = u * 1000.0 - 64.0
    j[0, 0] = (0.0, 0.0); j[0, 1] = (871.9, 871.9); j[0, 2] = (-100.0, 400.0); j[0, 3] = (13.625 * 10.5, 13.625 * 20.5)
    j[0, 4] = (13.625 * 63.49, 5.0); j[0, 5] = (2000.0, 2000.0); j[0, 6] = (-40.0, -40.0); j[0, 7] = (13.625 * 66.4, 300.0)
    return j


def main():
    for name in ("cv2", "natsort", "loguru"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["natsort"].natsorted = sorted
    sys.modules["loguru"].logger = None
    sys.path.insert(0, "/root/reference")
    import generate_heatmap as ref
    j = joints_case()
    out = np.stack([ref.generate_target(joints=j[i], image_size=872, heatmap_size=64, num_joints=16, sigma=1.0) for i in range(4)])
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "gt_heatmap.npz"), heatmaps=out)
    print(out.shape, out.sum(axis=(2, 3))[0, :8])


if __name__ == "__main__":
    main()
