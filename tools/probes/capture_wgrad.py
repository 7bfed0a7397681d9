#!/usr/bin/env python3
"""Which launch of the weight-gradient path breaks hipStreamEndCapture when it runs on a forked stream?  (child process per case)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CASES = ["absmax_side", "wgrad_small_side", "wgrad_split_f32_side", "wgrad_split_h2_side", "wgrad_split_h2_main", "conv_side", "wgrad_split_h2_side_prealloc"]


def run(name):
    import torch
    from egorear_amd import hip
    dev = "cuda:0"
    side = torch.cuda.Stream(device=dev)
    ws_main = torch.empty(1 << 24, device=dev)
    ws_side = torch.empty(1 << 24, device=dev)
    arena = hip.AmaxArena(dev, records=64)
    n, h, w, cin, cout = 8, 64, 64, 128, 32
    x = torch.randn(n, h, w, cin, device=dev)
    dy = torch.randn(n, h, w, cout, device=dev)
    out = {}
    pre = torch.empty(cout, cin, device=dev)

    def work(ws):
        if name.startswith("absmax"):
            rec = arena.new()
            hip.absmax_record(x, rec)
            out["r"] = rec
        elif name.startswith("wgrad_small"):
            xs, ds = hip.Img(x[:1, :8].contiguous()), hip.Img(dy[:1, :8].contiguous())
            out["r"], _ = hip.conv2d_wgrad(xs, ds, 1, 1, 1, 0, ws, x6=False)
        elif name.startswith("wgrad_split_f32"):
            out["r"], _ = hip.conv2d_wgrad(hip.Img(x), hip.Img(dy), 1, 1, 1, 0, ws, x6="force")
        elif name.startswith("wgrad_split_h2"):
            xi, di = hip.Img(x), hip.Img(dy)
            xi.amax = di.amax = None
            out["r"], _ = hip.conv2d_wgrad(xi, di, 1, 1, 1, 0, ws, x6="force", amax_arena=arena, want_bias=True,
                                           dw=pre if name.endswith("prealloc") else None)
        elif name.startswith("conv"):
            out["r"] = x * 2

    def body():
        arena.begin()
        main = torch.cuda.current_stream()
        a = x * 1.0
        if name.endswith("main"):
            work(ws_main)
        else:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                work(ws_side)
            b = a + 1
            main.wait_stream(side)
        out["s"] = out["r"].float().sum()

    for _ in range(2):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="relaxed"):
        body()
    g.replay()
    torch.cuda.synchronize()
    print(name, "ok", float(out["s"]))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True)
            print(c, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1][:80], (r.stderr.strip().splitlines() or [""])[-1][:100] if r.returncode else "")
