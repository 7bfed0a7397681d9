#!/usr/bin/env python3
"""up2_relu_head timing at the pipeline's shape (batch 64: 256 images of 32 x 32 x 128 -> 15 planes of 64 x 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egorear_amd import hip
B, V, J, G = 64, 4, 15, 2
lo = hip.Img(torch.randn(B * V, 32, 32, 128, device="cuda"))
w = torch.randn(G, J, 128, device="cuda") * 0.1
b = torch.randn(G, J, device="cuda")
out = torch.empty(B, V, J, 64, 64, device="cuda")
plane = J * 64 * 64
f = lambda: hip.up2_relu_head(lo, w, b, out, hip.NMap(B, V * plane, plane), 2 * plane, groups=G)
for _ in range(3): f()
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
print(f"up2_relu_head {e0.elapsed_time(e1) / 20 * 1e3:.1f} us   checksum {float(out.double().sum()):.6f}")
