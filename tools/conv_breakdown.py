#!/usr/bin/env python3
"""Per-shape breakdown of the implicit-GEMM kernel over one full-pipeline forward (HIP events per launch).
    python tools/conv_breakdown.py [--batch 64] [--reps 3]"""
import argparse, collections, copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egorear_amd import configs, hip, synth
from egorear_amd.estimator import EgoPoseFormerMVFEX

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=64); ap.add_argument("--reps", type=int, default=3); ap.add_argument("--seq", action="store_true")
a = ap.parse_args()
net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg())).eval(); synth.load_synth(net, 42); net = net.cuda()
img = synth.synth_images(a.batch, 4, seed=1234).cuda()
agg = collections.OrderedDict()
with torch.no_grad():
    net(img); torch.cuda.synchronize()
    for _ in range(a.reps):
        hip.PROFILE = []; net(img); torch.cuda.synchronize(); prof, hip.PROFILE = hip.PROFILE, None
        for name, s, e, fl, nb, tag in prof:
            k = (name.replace("egr_", "").replace("_f32", ""), tag)
            d = agg.setdefault(k, [0, 0.0, 0.0, 0.0]); d[0] += 1; d[1] += s.elapsed_time(e); d[2] += fl; d[3] += nb
tot = sum(d[1] for d in agg.values())
print(f"{'kernel':22s} {'shape':44s} {'calls':>5s} {'ms/fwd':>8s} {'%':>6s} {'TFLOP/s':>8s} {'GB/s':>8s}")
for (name, tag), d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    ms = d[1] / a.reps
    print(f"{name:22s} {tag:44s} {d[0]//a.reps:5d} {ms:8.3f} {100*d[1]/tot:6.2f} {d[2]/d[1]/1e9:8.1f} {d[3]/d[1]/1e6:8.0f}")
print(f"total {tot/a.reps:.3f} ms/forward, batch {a.batch}")
if a.seq:   # the last forward, launch by launch
    for i, (name, s, e, fl, nb, tag) in enumerate(prof):
        ms = s.elapsed_time(e)
        print(f"{i:3d} {ms * 1e3:8.1f} us {fl / ms / 1e9 if ms else 0:7.1f} TF {nb / ms / 1e6 if ms else 0:7.0f} GB/s  {name.replace('egr_', '').replace('_f32', ''):20s} {tag}")
