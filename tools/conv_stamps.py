#!/usr/bin/env python3
"""Diagnostic: per-workgroup phase times of the conv kernel from s_memtime stamps (cycles of the 100 MHz-ish
constant clock are NOT used; s_memtime counts shader cycles).  python tools/conv_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egorear_amd import hip

SH = [(128, 64, 64, 128, 128, 1, 1, "1x1 K=128 N=128"), (128, 64, 64, 256, 128, 1, 1, "1x1 K=256 N=128"),
      (128, 16, 16, 256, 256, 3, 1, "3x3 layer3"), (128, 64, 64, 128, 128, 3, 1, "3x3 fpn")]
hip.lib.egr_conv_debug_stamps.argtypes = [ctypes.c_void_p]
X6 = "--x6" in sys.argv   # weights in the bf16x3 format: the split-bf16 kernel
for (n, h, w, cin, cout, k, s, label) in SH:
    x = torch.randn(n, h, w, cin, device="cuda"); wt = torch.randn(cout, k * k * cin, device="cuda") * 0.05
    if X6:
        wt = hip.pack_w6(wt)
    sh = torch.randn(cout, device="cuda"); out = hip.Img(torch.empty(n, h, w, cout, device="cuda"))
    M = n * h * w; blocks = (M // 128) * (cout // 128)
    buf = torch.zeros(blocks * 8, dtype=torch.int64, device="cuda")
    run = lambda: hip.conv2d(hip.Img(x), wt, cout, k, k, s, k // 2, shift=sh, act=1, out=out)
    for _ in range(3): run()
    torch.cuda.synchronize()
    hip.lib.egr_conv_debug_stamps(ctypes.c_void_p(buf.data_ptr())); run(); torch.cuda.synchronize()
    hip.lib.egr_conv_debug_stamps(None)
    t = buf.view(blocks, 8).cpu().double()
    d = (t[:, 1:6] - t[:, 0:5])
    tot = (t[:, 5] - t[:, 0])
    span = (t[:, 5].max() - t[:, 0].min())
    names = ["decode", "1st DMA", "k-loop", "stage", "store issue"]
    print(f"{label:18s} blocks {blocks:5d}  kernel span {span:10.0f} cyc | per block: total {tot.mean():8.0f} | " +
          " ".join(f"{nm} {d[:, i].mean():7.0f}" for i, nm in enumerate(names)))
    # co-residency: group blocks by physical CU (XCC id, SE/SH/CU bits of HW_ID) and measure how much of the time at least
    # one resident block is inside its K loop (MFMA can be busy) vs none (MFMA certainly idle)
    ids = buf.view(blocks, 8)[:, 6].cpu().numpy()
    cu = (ids >> 16) * 4096 + ((ids >> 8) & 0x7f)
    import numpy as np
    tt = t.numpy()
    busy = idle = 0.0; both = 0.0
    for c in np.unique(cu):
        sel = tt[cu == c]
        ev = []
        for r in sel:
            ev.append((r[2], +1)); ev.append((r[3], -1))       # K loop interval [first chunk landed, k loop done]
        ev.sort()
        lo, hi = sel[:, 0].min(), sel[:, 5].max()
        cur, last, b1, b2 = 0, lo, 0.0, 0.0
        for (x, dlt) in ev:
            if cur >= 1: b1 += x - last
            if cur >= 2: b2 += x - last
            cur += dlt; last = x
        busy += b1; both += b2; idle += (hi - lo) - b1
    print(f"{'':18s} CUs seen {len(np.unique(cu))}; per-CU time with >=1 block in its K loop {busy / (busy + idle):.2f}, with 2 in K loop {both / (busy + idle):.2f}, none {idle / (busy + idle):.2f}")
    # concurrency: sum of block lifetimes / span / 512 slots
    print(f"{'':18s} occupancy of 512 slots: {tot.sum() / span / 512:.2f}; start-to-start gap on a slot ~ {span * 512 / blocks - tot.mean():8.0f} cyc")
