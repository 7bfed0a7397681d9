#!/usr/bin/env python3
"""Graph-replay latency at small batches (1, 2, 4, 8).   python tools/latency_small.py"""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egorear_amd import configs, synth
from egorear_amd.estimator import EgoPoseFormerMVFEX
from egorear_amd.runner import GraphedForward
net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg())).eval(); synth.load_synth(net, 42); net = net.cuda()
g = GraphedForward(net)
for B in (1, 2, 4, 8, 16):
    img = synth.synth_images(B, 4, seed=1).cuda()
    with torch.no_grad():
        for _ in range(3): g(img)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): g(img)
        torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 50 * 1e3
    print(f"B={B:3d} graph {tg:7.3f} ms", flush=True)
