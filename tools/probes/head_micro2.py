#!/usr/bin/env python3
"""egr_up2_relu_head_f32 alone at the forward's shapes, persistent vs tile kernel.   python tools/probes/head_micro2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egorear_amd import hip
from egorear_amd.hip import NMap
for groups, B in ((2, 128), (4, 64)):
    V, cout, h, w = groups, 15, 32, 32
    lo = torch.randn(V * B, h, w, 128, device="cuda")
    wt = torch.randn(groups, cout, 128, device="cuda") / 11
    bias = torch.randn(groups, cout, device="cuda")
    plane = cout * 4 * h * w
    planes = torch.zeros(B, V, cout, 2 * h, 2 * w, device="cuda")
    for persist in (1, 0, 1, 0):
        hip.lib.egr_head_set_persist(persist)
        for _ in range(3):
            hip.up2_relu_head(hip.Img(lo), wt, bias, planes, NMap(B, V * plane, 0), plane, groups=groups)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            hip.up2_relu_head(hip.Img(lo), wt, bias, planes, NMap(B, V * plane, 0), plane, groups=groups)
        e.record(); torch.cuda.synchronize()
        print(f"groups {groups} images {V * B}: persist={persist} {s.elapsed_time(e) / 20 * 1e3:7.1f} us", flush=True)
