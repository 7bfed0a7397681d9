#!/usr/bin/env python3
"""Where the waves of conv_tapx_kernel spend their cycles (s_memtime sums per wave).  Needs the diagnostic build:
    python tools/build_variant.py stamps -DTAPX_STAMPS --src egr_conv_tapx.hip
    EGR_LIB=egorear_amd/csrc/libegorear_hip_stamps.so python tools/tapx_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from egorear_amd import hip
SHAPES = [(128, 64, 256, 256, 0, 0, "1x1 256->256 @64"), (256, 32, 512, 128, 0, 0, "1x1 512->128 @32"), (128, 64, 128, 128, 1, 0, "fpn 3x3 128->128 @64"), (128, 64, 64, 64, 1, 1, "layer1 64->64 (+res)"), (256, 32, 128, 128, 1, 1, "layer2 (+res)"),
          (512, 16, 256, 256, 1, 1, "layer3 (+res)"), (64, 64, 256, 512, 2, 0, "refiner s2 256->512")]
for (n, hw, cin, cout, stride, res, label) in SHAPES:
    k = 1 if stride == 0 else 3
    stride = max(stride, 1)
    ho = hw // stride
    x = torch.randn(n, hw, hw, cin, device="cuda")
    wt = hip.add_wh2(hip.pack_w6(torch.randn(cout, k * k * cin, device="cuda") * 0.05))
    rec = torch.zeros(64, dtype=torch.int32, device="cuda"); rec[0] = x.abs().max().reshape(1).view(torch.int32)[0]
    r = torch.randn(n, ho, ho, cout, device="cuda") if res else None
    buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
    hip.H2 = True
    def run():
        return hip.conv2d(hip.Img(x, amax=rec), wt, cout, k, k, stride, k // 2, act=1, res=hip.Img(r) if res else None, res_mode=1 if res else 0)
    for _ in range(3): run()
    torch.cuda.synchronize()
    hip.lib.egr_conv_debug_stamps(C.c_void_p(buf.data_ptr()))
    run(); torch.cuda.synchronize()
    hip.lib.egr_conv_debug_stamps(None)
    assert hip.lib.egr_conv_last_kernel() == 6, hip.lib.egr_conv_last_kernel()
    b = buf.view(256, 8, 8).double().cpu()
    m, l = b[:, :4].reshape(-1, 8), b[:, 4:].reshape(-1, 8)
    m, l = m[m[:, 3] > 0], l[l[:, 3] > 0]
    mul, bar, park, tot, chunks = [float(m[:, i].mean()) for i in range(5)]
    spread = f"per-workgroup total: min {float(m[:, 3].min()):.0f} / mean {tot:.0f} / max {float(m[:, 3].max()):.0f} (+{100 * (float(m[:, 3].max()) / tot - 1):.1f} %)"
    tiles_m = n * ho * ho // (cout if False else 1)      # (unused)
    bound = 9 * 3 * 4 * 32 * (2 if os.environ.get("EGR_CONV_TAPX_FN", "0") == "2" else (1 if os.environ.get("EGR_CONV_TAPX_FN", "0") == "1" else 0))
    print(f"{label:24s} multiplying waves {len(m)}: total {tot:9.0f} cyc = multiply {mul:9.0f} ({100*mul/tot:.1f} %) + chunk-barrier wait {bar:8.0f} ({100*bar/tot:.1f} %) + "
          f"park/hand-over {park:8.0f} ({100*park/tot:.1f} %); {chunks:.0f} chunks -> {mul/chunks:.0f} cyc per chunk (MFMA-bound: {bound if bound else '3456 / 6912'})")
    print(f"{'':24s} {spread}")
    work, lbar, hand, ltot, _, drain = [float(l[:, i].mean()) for i in range(6)]
    print(f"{'':24s} loading waves     {len(l)}: total {ltot:9.0f} cyc = work {work:9.0f} ({100*work/ltot:.1f} %, of it epilogue slices {drain:8.0f}) + barrier wait {lbar:8.0f} "
          f"({100*lbar/ltot:.1f} %) + hand-over {hand:8.0f} ({100*hand/ltot:.1f} %); work per chunk {work/chunks:.0f} cyc", flush=True)
