#!/usr/bin/env python3
"""Per-launch time of small GEMM launches inside a hipGraph (the latency-bound launches of the heads): a chain of dependent launches,
so the figure is the full launch-to-launch period.   python tools/small_launch_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egorear_amd import engine, hip, hip_train as T

dev = torch.device("cuda")
st = engine.State(dev)


def chain_time(fn, n=64, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / n * 1e3


x0 = torch.zeros(1024, device=dev)
print(f"fill 4 KB (launch floor)          {chain_time(lambda: T.fill(x0)):6.2f} us")
for (M, N, K) in [(2048, 128, 128), (2048, 128, 512), (2048, 512, 128), (512, 128, 128), (64, 128, 2048), (16, 128, 128), (8192, 128, 64)]:
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    p = engine.pack_linears([(w, b)])
    y = [x]

    def f():
        engine.linear(st, x, p, 0)
    t = chain_time(f)
    # the same launch forced onto the fp16-scheme kernel (record on the rows, size rule off)
    rec = torch.zeros(64, dtype=torch.int32, device=dev)
    hip.absmax_record(x, rec)
    saved = hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS
    hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = 0, 0.0
    xi = hip.Img(x.view(M, 1, 1, K), amax=rec)

    def f_split(im):
        hip.conv2d(im, p.w, p.cout, 1, 1, 1, 0, shift=p.shift, workspace=st.workspace, split_k=0)
    try:
        t2 = chain_time(lambda: f_split(xi))                                    # fp16 scheme (the rows carry a record)
        t3 = chain_time(lambda: f_split(hip.Img(x.view(M, 1, 1, K))))           # bf16 scheme (no record)
        kern = hip.lib.egr_conv_last_kernel()
    finally:
        hip.X6_MIN_ROWS, hip.X6_MIN_FLOPS = saved
    print(f"linear M{M} N{N} K{K}: fp32 {t:6.2f} us ({2.0 * M * N * K / t / 1e6:6.2f} TF)   fp16 scheme {t2:6.2f} us   bf16 scheme {t3:6.2f} us")
