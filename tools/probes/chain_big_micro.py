#!/usr/bin/env python3
"""conv_pw2_kernel (256 -> 256 -> 128, both matrices streamed) against the two launches it replaces, at the heads' shapes, as captured
graphs.    python tools/probes/chain_big_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn as nn
from egorear_amd import engine, hip
from egorear_amd.hip import Img, ACT_RELU, ACT_NONE

dev = "cuda:0"
torch.manual_seed(0)
for G, ng in ((2, 128), (4, 64)):
    c1 = [nn.Conv2d(256, 256, 1).to(dev) for _ in range(G)]
    c2 = [nn.Conv2d(256, 128, 1).to(dev) for _ in range(G)]
    with torch.no_grad():
        p1, p2 = engine.pack_convs(c1), engine.pack_convs(c2)
    x = torch.relu(torch.randn(G * ng, 32, 32, 256, device=dev))
    rec = torch.zeros(64, dtype=torch.int32, device=dev)
    hip.absmax_record(x, rec)
    xin = Img(x, amax=rec)
    ws = torch.empty(1 << 22, device=dev)
    arena = hip.AmaxArena(torch.device(dev))

    def two():
        arena.begin()
        mid = hip.conv2d(xin, p1.w, 256, 1, 1, 1, 0, shift=p1.shift, act=ACT_RELU, workspace=ws, groups=G, amax_out=arena.new())
        return hip.conv2d(mid, p2.w, 128, 1, 1, 1, 0, shift=p2.shift, act=ACT_NONE, workspace=ws, groups=G, amax_out=arena.new()).t

    def one():
        arena.begin()
        return hip.conv1x1_chain(xin, p1.w, p2.w, 256, 128, shift1=p1.shift, shift2=p2.shift, act1=ACT_RELU, act2=ACT_NONE, groups=G,
                                 amax_out=arena.new()).t

    def timeit(fn):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            keep = fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(20):
            g.replay()
        e_.record(); torch.cuda.synchronize()
        return s_.elapsed_time(e_) / 20 * 1e3

    a, b = two(), one()
    print(f"G {G} x {ng} images of 32 x 32: two launches {timeit(two):7.1f} us, one streamed chain {timeit(one):7.1f} us, max |diff| {float((a - b).abs().max()):.2e} "
          f"(|y| max {float(a.abs().max()):.2f})", flush=True)
