#!/usr/bin/env python3
"""Per-entry-point kernel time of one forward (HIP events; best of 5), batch 64 and 1.   python tools/kernel_times.py [name-substring]"""
import collections, copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egorear_amd import configs, hip, synth
from egorear_amd.estimator import EgoPoseFormerMVFEX
pat = sys.argv[1] if len(sys.argv) > 1 else ""
net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg())).eval(); synth.load_synth(net, 42); net = net.cuda()
for B in (64, 1):
    img = synth.synth_images(B, 4, seed=1234).cuda()
    with torch.no_grad():
        net(img); torch.cuda.synchronize()
        runs = []
        for _ in range(5):
            hip.PROFILE = []; net(img); torch.cuda.synchronize(); prof, hip.PROFILE = hip.PROFILE, None
            runs.append([(name, s.elapsed_time(e) * 1e3) for name, s, e, *_ in prof])
    best = [(runs[0][i][0], min(r[i][1] for r in runs)) for i in range(len(runs[0]))]
    agg = collections.OrderedDict()
    for n, t in best:
        agg.setdefault(n, []).append(t)
    print(f"B={B}: total {sum(t for _, t in best):.1f} us over {len(best)} launches")
    for n, ts in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if pat in n:
            print(f"   {n:32s} {sum(ts):9.1f} us  x{len(ts):3d}   " + (", ".join(f"{t:.1f}" for t in ts) if len(ts) <= 6 else ""))
