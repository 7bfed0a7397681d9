#!/usr/bin/env python3
"""Probe: one forward of batch B against two forwards of batch B/2 captured on two streams of one hipGraph
(the latency-bound tail of one half can overlap the convolutions of the other).  Timing only: both halves share one module,
hence one split-K workspace - results of the two-stream replay are NOT valid, this only measures the overlap.
    python tools/two_stream_probe.py [--batch 64] [--parts 2]"""
import argparse, copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egorear_amd import configs, synth
from egorear_amd.estimator import EgoPoseFormerMVFEX

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--parts", type=int, default=2)
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda:0")
net = EgoPoseFormerMVFEX(**copy.deepcopy(configs.pose3d_cfg("ego4view_syn"))).eval()
synth.load_synth(net, 42)
net = net.to(dev)
B = a.batch
img = synth.synth_images(B, 4, seed=1234).to(dev)
parts = list(img.chunk(a.parts))


def timeit(run):
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.steps * 1e3


with torch.no_grad():
    net(img); [net(p) for p in parts]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        net(img); [net(p) for p in parts]
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        o = net(img)
    t1 = timeit(g1.replay)
    gs = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gs):
        os_ = [net(p) for p in parts]
    ts = timeit(gs.replay)
    g2 = torch.cuda.CUDAGraph()
    streams = [torch.cuda.Stream() for _ in parts]
    with torch.cuda.graph(g2):
        cur = torch.cuda.current_stream()
        outs = []
        for s, p in zip(streams, parts):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs.append(net(p))
        for s in streams:
            cur.wait_stream(s)
    t2 = timeit(g2.replay)
print(f"B={B}: one forward {t1:.3f} ms ({B / t1 * 1e3:.0f} f/s); {a.parts} x B/{a.parts} serial {ts:.3f} ms ({B / ts * 1e3:.0f} f/s); "
      f"{a.parts} x B/{a.parts} on {a.parts} streams {t2:.3f} ms ({B / t2 * 1e3:.0f} f/s)")
