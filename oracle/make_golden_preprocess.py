"""Golden vectors for the pre-processing row: Pillow's own output on seeded synthetic frames.
    python -m oracle.make_golden_preprocess      # writes tests/golden/preprocess_pil.npz
Build-container tool (needs Pillow); TEST INFRASTRUCTURE.  The input frames are regenerated from egorear_amd.synth."""
import os
import sys

import numpy as np
from PIL import Image

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from egorear_amd import synth  # noqa: E402

st = {}
for seed in (0, 1):
    frames = synth.synth_raw_frames(1, 4, seed=seed).numpy()      # (1, 4, 872, 872, 3) uint8
    for v in range(4):
        out = np.asarray(Image.fromarray(frames[0, v]).resize([256, 256], Image.BICUBIC))
        if seed == 0 and v == 0:
            st["s0_v0_full"] = out                               # one full image
        st[f"s{seed}_v{v}_sl"] = out[::4, ::4].copy()            # strided sample of every image
        st[f"s{seed}_v{v}_sum"] = np.int64(out.astype(np.int64).sum())
# a small non-square up/down case, full
rng = synth.uniform01("preprocess.small", 0, 50 * 60 * 3)
small = (rng.reshape(50, 60, 3) * 255).astype(np.uint8)
st["small_in"] = small
st["small_out_37x41"] = np.asarray(Image.fromarray(small).resize([41, 37], Image.BICUBIC))
st["small_out_128x100"] = np.asarray(Image.fromarray(small).resize([100, 128], Image.BICUBIC))
np.savez_compressed(os.path.join(REPO, "tests", "golden", "preprocess_pil.npz"), **st)
print("wrote preprocess_pil.npz", {k: getattr(v, "shape", v) for k, v in st.items() if "full" in k or "small" in k})
