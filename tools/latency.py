#!/usr/bin/env python3
"""Latency / throughput of the full pipeline vs batch size, eager launches vs hipGraph replay.  python tools/latency.py"""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egorear_amd import configs, synth
from egorear_amd.estimator import EgoPoseFormerMVFEX
from egorear_amd.runner import GraphedForward

net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg())).eval(); synth.load_synth(net, 42); net = net.cuda()
g = GraphedForward(net)
print(f"{'batch':>5s} {'eager ms':>9s} {'graph ms':>9s} {'graph frames/s':>14s}")
for B in (1, 2, 4, 8, 16, 32, 64, 128):
    img = synth.synth_images(B, 4, seed=1).cuda()
    with torch.no_grad():
        for _ in range(3): net(img)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20 if B <= 16 else 8
        for _ in range(n): net(img)
        torch.cuda.synchronize(); te = (time.perf_counter() - t0) / n * 1e3
    g(img); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g(img)
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / n * 1e3
    print(f"{B:5d} {te:9.3f} {tg:9.3f} {B / tg * 1e3:14.1f}", flush=True)
