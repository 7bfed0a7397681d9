#!/usr/bin/env python3
"""Per-tile phase times of the persistent split-bf16 conv launches (s_memtime stamps, egr_conv_debug_stamps):
    [0] tile's turn begins  [2] first stage staged  [3] K loop done  [4] accumulators staged  [5] stores issued
python tools/conv_stamps_p.py [cin ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egorear_amd import hip
hip.lib.egr_conv_debug_stamps.argtypes = [ctypes.c_void_p]
for cin in ([int(v) for v in sys.argv[1:]] or [128, 64, 256]):
    n, h, w, cout = 128, 64, 64, 128
    x = torch.randn(n, h, w, cin, device="cuda"); wt = hip.pack_w6(torch.randn(cout, cin, device="cuda") * 0.05)
    sh = torch.randn(cout, device="cuda"); out = hip.Img(torch.empty(n, h, w, cout, device="cuda"))
    tiles = n * h * w // 128
    buf = torch.zeros(tiles * 8, dtype=torch.int64, device="cuda")
    run = lambda: hip.conv2d(hip.Img(x), wt, cout, 1, 1, 1, 0, shift=sh, act=1, out=out)
    for _ in range(3): run()
    torch.cuda.synchronize()
    hip.lib.egr_conv_debug_stamps(ctypes.c_void_p(buf.data_ptr())); run(); torch.cuda.synchronize(); hip.lib.egr_conv_debug_stamps(None)
    t = buf.view(tiles, 8).double().cpu()
    ok = (t[:, 0] > 0) & (t[:, 5] > t[:, 0]) & (t[:, 2] > 0)
    t = t[ok]
    names = [("wait+stage0", 0, 2), ("k-loop", 2, 3), ("acc->lds", 3, 4), ("stores", 4, 5), ("tile total", 0, 5)]
    print(f"cin {cin}: {int(ok.sum())} tiles stamped | " + " ".join(f"{nm} {float((t[:, b] - t[:, a]).median()):8.0f}" for nm, a, b in names))
