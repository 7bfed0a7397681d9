#!/usr/bin/env python3
"""upsample2x timing at the pipeline's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egorear_amd import hip
for (n, h, c) in ((256, 32, 128), (256, 16, 64)):
    x = hip.Img(torch.randn(n, h, h, c, device="cuda"))
    f = lambda: hip.upsample2x(x, relu=True)
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"upsample2x {n}x{h}x{h}x{c}: {us:.1f} us  {n * h * h * c * 4 * 5 / us / 1e6:.2f} TB/s")
