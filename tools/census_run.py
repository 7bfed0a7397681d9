#!/usr/bin/env python3
"""A long parity census (oracle/census.py) as evidence for profiles/: N frames through the HIP path at batch 64 under the shipped launch
policy against the CPU oracle - all arg-maxes of both heat-map sets, valid masks, four pose sets, tie exposure, float64 referee.
    python tools/census_run.py --frames 2048 --out gpurun_out/census.json        (one line of progress per GPU batch)
Test infrastructure (imports oracle/): not part of the product or of bench.py's timed region."""
import argparse, copy, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egorear_amd import configs, hip, synth
from egorear_amd.estimator import EgoPoseFormerMVFEX
from oracle import census
from oracle import egorear_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=2048)
ap.add_argument("--seed0", type=int, default=100)
ap.add_argument("--out", default="")
ap.add_argument("--weight-seed", type=int, default=42, help="seed of the synthetic weights (egorear_amd/synth.py); the tests and bench.py use 42")
ap.add_argument("--camera", default="ego4view_syn", choices=["ego4view_syn", "ego4view_rw"], help="camera model / config family (rw: with coord_trans_mat)")
a = ap.parse_args()
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg(a.camera))).eval()
synth.load_synth(net, a.weight_seed)
sd = {k: v.clone() for k, v in net.state_dict().items()}
net = net.to("cuda:0")
cams = O.make_cameras(a.camera, os.path.join(os.path.dirname(os.path.abspath(synth.__file__)), "calib", "ego4view"))
assert hip.H2 and hip.X6_MIN_ROWS > 0
acc, t0 = None, time.time()
nb = a.frames // 64
for i in range(nb):
    scale = (1.0, 0.35, 0.6, 1.5)[i % 4]
    img = synth.synth_images(64, 4, seed=a.seed0 + i, scale=scale)
    ctm = synth.synth_coord_trans_mat(64, seed=500 + i) if a.camera == "ego4view_rw" else None
    part = census.run(net, sd, cams, O, [img], "cuda:0", oracle_batch=8, ctms=[ctm] if ctm is not None else None)
    for d in part["mismatch_detail"]:
        d["batch"], d["seed"], d["scale"] = i, a.seed0 + i, scale
    acc = census.merge(acc, part)
    print(f"batch {i + 1}/{nb} (seed {a.seed0 + i}, scale {scale}): {acc['frames']} frames, {acc['argmax_mismatches']} mismatches "
          f"({acc['argmax_mismatches_outside_rounding']} outside rounding), valid flips {acc['valid_mask_mismatches']}, "
          f"max joint err {acc['max_joint_err_cm']:.2e} cm, {time.time() - t0:.0f} s", flush=True)
acc["policy"] = "shipped (fp16 scheme by size), batch 64"
acc["weight_seed"], acc["camera"] = a.weight_seed, a.camera
acc["seeds"] = [a.seed0, a.seed0 + nb - 1]
acc["image_scales"] = [1.0, 0.35, 0.6, 1.5]
print(json.dumps(acc))
if a.out:
    json.dump(acc, open(a.out, "w"), indent=1)
