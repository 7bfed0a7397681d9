#!/usr/bin/env python3
"""Micro-benchmark of egr_conv1x1_chain_f32 on the three chains of the forward (batch 64), against the two launches each replaces.
    python tools/chain_micro.py            (EGR_LIB=<variant> for the elimination builds: tools/build_variant.py NAME -DCHAIN_EXP_... --src egr_conv_chain.hip)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egorear_amd import hip
DEV = "cuda:0"
torch.manual_seed(0)
SHAPES = [("fpn lateral0 -> fuse0 (64 -> 128 -> 128 @64x64, up-sampled residual)", 128, 64, 64, 2, 3),
          ("fpn lateral1 -> fuse1 (128 -> 128 -> 128 @32x32, up-sampled residual)", 128, 32, 128, 2, 3),
          ("refiners' frame_feat_refined_proj (128 -> 128 -> 128 @32x32)", 64, 32, 128, 4, 0)]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, n, hw, cin, G, res_mode in SHAPES:
    x = torch.relu(torch.randn(G * n, hw, hw, cin, device=DEV))
    rec = torch.zeros(64, dtype=torch.int32, device=DEV)
    hip.absmax_record(x, rec)
    xin = hip.Img(x, amax=rec)
    def mk(co, ci):
        w = torch.randn(G, co, ci, device=DEV) / math.sqrt(ci)
        return hip.add_wh2(hip.pack_w6(w.contiguous()))
    p1, p2 = mk(128, cin), mk(128, 128)
    b1, b2 = torch.randn(G, 128, device=DEV), torch.randn(G, 128, device=DEV)
    res = hip.Img(torch.randn(G * n, hw // 2, hw // 2, 128, device=DEV)) if res_mode == 3 else None
    out = hip.Img(torch.empty(G * n, hw, hw, 128, device=DEV))
    mid = hip.Img(torch.empty(G * n, hw, hw, 128, device=DEV))
    r1 = torch.zeros(64, dtype=torch.int32, device=DEV)
    r2 = torch.zeros(64, dtype=torch.int32, device=DEV)
    t_chain = timeit(lambda: hip.conv1x1_chain(xin, p1, p2, 128, 128, shift1=b1, shift2=b2, act1=1, act2=1, res=res, res_mode=res_mode, groups=G, out=out, amax_out=r2))
    def two():
        m = hip.conv2d(xin, p1, 128, 1, 1, 1, 0, shift=b1, act=1, groups=G, out=mid, amax_out=r1)
        hip.conv2d(m, p2, 128, 1, 1, 1, 0, shift=b2, act=1, res=res, res_mode=res_mode, groups=G, out=out, amax_out=r2)
    t_two = timeit(two)
    px = G * n * hw * hw
    hbm = 4.0 * px * (cin + 128) + (4.0 * px / 4 * 128 if res_mode == 3 else 0)
    print(f"{name}: chain {t_chain:7.1f} us ({hbm / t_chain / 1e6:5.2f} TB/s of input + residual + output), two launches {t_two:7.1f} us")
