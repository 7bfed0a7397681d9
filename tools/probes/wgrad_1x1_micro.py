#!/usr/bin/env python3
"""The 1x1 weight gradients of the training step (HBM-bound: two tensors read once, a tiny result), per launch and in TB/s.
   python tools/wgrad_1x1_micro.py          (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egorear_amd import hip

# (groups, images per group, h = w, cin, cout, bias)   -- batch 32 x 4 views
SH = [(2, 64, 64, 128, 128, True), (2, 64, 32, 128, 128, True), (2, 64, 64, 64, 128, True), (1, 128, 64, 128, 64, True),
      (2, 64, 32, 256, 256, True), (4, 32, 32, 128, 128, True), (2, 64, 64, 128, 32, True), (4, 32, 64, 128, 256, True)]
ws = torch.empty(1 << 26, device="cuda")


def timeit(f, reps=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"{'shape':44s} {'ms':>8s} {'TB/s':>6s} {'TF':>7s}   no bias: ms")
for (G, n, hw, cin, cout, bias) in SH:
    x = hip.Img(torch.randn(G * n, hw, hw, cin, device="cuda"))
    dy = hip.Img(torch.randn(G * n, hw, hw, cout, device="cuda"))
    for im in (x, dy):
        rec = torch.zeros(64, dtype=torch.int32, device="cuda")
        hip.absmax_record(im.t, rec)
        im.amax = rec
    t = timeit(lambda: hip.conv2d_wgrad(x, dy, 1, 1, 1, 0, ws, want_bias=bias, groups=G))
    t0 = timeit(lambda: hip.conv2d_wgrad(x, dy, 1, 1, 1, 0, ws, want_bias=False, groups=G))
    M = n * hw * hw
    nbytes = 4.0 * G * M * (cin + cout)
    print(f"G{G} M{M} cin{cin} cout{cout} h2={hip.lib.egr_wgrad_last_h2()} kern={hip.lib.egr_wgrad_last_kernel()}".ljust(44)
          + f" {t:8.4f} {nbytes / t / 1e9:6.2f} {2.0 * G * M * cin * cout / t / 1e9:7.1f}   {t0:8.4f}", flush=True)
