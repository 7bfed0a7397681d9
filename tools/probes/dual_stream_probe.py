#!/usr/bin/env python3
"""Probe: does overlapping two half-batch forwards (two hipGraphs on two streams, own weights / workspaces) beat one
full-batch graph?  Measured r01: 2941 vs 2952 frames/s - no (the memory-bound kernels of one half do not fill the gaps of the
other's matrix kernels by enough to pay for the smaller launches).   python tools/dual_stream_probe.py"""
import copy, sys, time
sys.path.insert(0, "/root/repo")
import torch
from egorear_amd import configs, synth
from egorear_amd.estimator import EgoPoseFormerMVFEX
dev = "cuda"
def build():
    net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg("ego4view_syn"))).eval(); synth.load_synth(net, 42); return net.to(dev)
def capture(net, img):
    with torch.no_grad():
        net(img); torch.cuda.synchronize()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            net(img)
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = net(img)
    return g
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
n1 = build(); g64 = capture(n1, synth.synth_images(64, 4, seed=1).to(dev))
t = timeit(g64.replay); print(f"single graph B=64: {t*1e3:.3f} ms  {64/t:.1f} frames/s")
na, nb = build(), build()
ga = capture(na, synth.synth_images(32, 4, seed=2).to(dev)); gb = capture(nb, synth.synth_images(32, 4, seed=3).to(dev))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def dual():
    with torch.cuda.stream(sa): ga.replay()
    with torch.cuda.stream(sb): gb.replay()
t = timeit(dual); print(f"two graphs B=32 on two streams: {t*1e3:.3f} ms  {64/t:.1f} frames/s")
t = timeit(ga.replay); print(f"one graph B=32 alone: {t*1e3:.3f} ms  {32/t:.1f} frames/s")
