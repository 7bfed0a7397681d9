#!/usr/bin/env python3
"""Stem timing: stem, stem + maxpool, stem_pool at (B, 4 views, 256 x 256).   python tools/stem_micro.py [--batch 64] [--reps 20]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egorear_amd import hip

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=64); ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
B, V, H, W, G = a.batch, 4, 256, 256, 2
img = torch.randn(B, V, 3, H, W, device="cuda")
wp = torch.zeros(G, 64, 148, device="cuda"); wp[:, :, :147] = torch.randn(G, 64, 147, device="cuda") * 0.1
sc, sh = torch.rand(G, 64, device="cuda") + 0.5, torch.randn(G, 64, device="cuda")


def timed(f):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps * 1e3


fl = 2.0 * B * V * (H // 2) * (W // 2) * 64 * 147
t = timed(lambda: hip.stem(img, 0, 2, wp, sc, sh, groups=G)); print(f"stem            {t:8.1f} us  {fl / t / 1e6:6.1f} TF")
y = hip.stem(img, 0, 2, wp, sc, sh, groups=G)
t2 = timed(lambda: hip.maxpool(y, 3, 2, 1)); print(f"maxpool         {t2:8.1f} us   (sum {t + t2:.1f})")
if hasattr(hip, "stem_pool"):
    t = timed(lambda: hip.stem_pool(img, 0, 2, wp, sc, sh, groups=G)); print(f"stem_pool       {t:8.1f} us  {fl / t / 1e6:6.1f} TF")
if hasattr(hip, "stem_x6"):
    w6 = hip.pack_stem_w6(wp)
    t = timed(lambda: hip.stem_x6(img, 0, 2, w6, sc, sh, groups=G)); print(f"stem_x6         {t:8.1f} us  {fl / t / 1e6:6.1f} TF")
    t = timed(lambda: hip.stem_x6(img, 0, 2, w6, sc, sh, groups=G, pool=True)); print(f"stem_x6 pool    {t:8.1f} us  {fl / t / 1e6:6.1f} TF")
if hasattr(hip, "pack_stem_wh2"):
    wh2, wds = hip.pack_stem_wh2(wp)
    t = timed(lambda: hip.stem_x6(img, 0, 2, wh2, sc, sh, groups=G, w_descale=wds)); print(f"stem_h2         {t:8.1f} us  {fl / t / 1e6:6.1f} TF")
    t = timed(lambda: hip.stem_x6(img, 0, 2, wh2, sc, sh, groups=G, pool=True, w_descale=wds)); print(f"stem_h2 pool    {t:8.1f} us  {fl / t / 1e6:6.1f} TF")
