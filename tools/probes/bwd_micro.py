#!/usr/bin/env python3
"""Throughput of the conv backward bricks (dgrad = transposed mode of egr_conv2d_nhwc_f32, wgrad = egr_conv2d_wgrad_f32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egorear_amd import hip
from egorear_amd.engine import pack_conv_weight

SH = [(128, 64, 64, 64, 64, 3, 1, "layer1 3x3 64->64"), (128, 32, 32, 128, 128, 3, 1, "layer2 3x3 128->128"),
      (128, 16, 16, 256, 256, 3, 1, "layer3 3x3 256->256"), (128, 8, 8, 512, 512, 3, 1, "layer4 3x3 512->512"),
      (128, 64, 64, 64, 128, 3, 2, "layer2.0 3x3 s2 64->128"), (64, 64, 64, 256, 512, 3, 2, "refiner 3x3 s2 256->512"),
      (128, 64, 64, 256, 128, 1, 1, "1x1 256->128 @64")]
ws = torch.empty(1 << 26, device="cuda")
def timeit(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
print(f"{'shape':28s} {'fwd TF':>8s} {'dgrad TF':>9s} {'wgrad TF':>9s}")
for (n, h, w, cin, cout, k, s, label) in SH:
    pad = k // 2; ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    x = hip.Img(torch.randn(n, h, w, cin, device="cuda")); dy = hip.Img(torch.randn(n, ho, wo, cout, device="cuda"))
    wt = torch.randn(cout, cin, k, k, device="cuda") * 0.05
    wf = pack_conv_weight(wt); wd = pack_conv_weight(wt.transpose(0, 1).contiguous())
    fl = 2.0 * n * ho * wo * cout * k * k * cin
    tf = timeit(lambda: hip.conv2d(x, wf, cout, k, k, s, pad))
    td = timeit(lambda: hip.conv2d(dy, wd, cin, k, k, s, pad, transposed_out_hw=(h, w)))
    tw = timeit(lambda: hip.conv2d_wgrad(x, dy, k, k, s, pad, ws))
    print(f"{label:28s} {fl/tf/1e9:8.1f} {fl/td/1e9:9.1f} {fl/tw/1e9:9.1f}", flush=True)
