#!/usr/bin/env python3
"""Time of the fused transformer-layer launches inside one forward (HIP events), batch 64 and 1."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egorear_amd import configs, hip, synth
from egorear_amd.estimator import EgoPoseFormerMVFEX
net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg())).eval(); synth.load_synth(net, 42); net = net.cuda()
for B in (64, 1):
    img = synth.synth_images(B, 4, seed=1234).cuda()
    with torch.no_grad():
        net(img); torch.cuda.synchronize()
        tot = []
        for _ in range(5):
            hip.PROFILE = []; net(img); torch.cuda.synchronize(); prof, hip.PROFILE = hip.PROFILE, None
            tot.append([s.elapsed_time(e) * 1e3 for name, s, e, *_ in prof if name == "egr_joint_layer_f32"])
    best = [min(t[i] for t in tot) for i in range(len(tot[0]))]
    print(f"B={B}: joint_layer launches (us): " + ", ".join(f"{v:.1f}" for v in best) + f"  sum {sum(best):.1f}")
