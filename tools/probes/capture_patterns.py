#!/usr/bin/env python3
"""Which two-stream patterns survive hipStreamEndCapture on this stack?  Each pattern in its own process (a crash is a segfault).
    python tools/probes/capture_patterns.py            -> runs all patterns as child processes, prints rc per pattern
    python tools/probes/capture_patterns.py P2         -> runs one pattern in this process"""
import subprocess
import sys

PATTERNS = ["P1_double_join", "P2_fork_twice_no_join_between", "P3_fork_join_fork_join", "P4_join_then_new_fork", "P5_side_alloc_freed_in_capture",
            "P6_main_consumes_side_tensor_then_join_again", "P7_side_last_then_join"]


def run(name):
    import torch
    dev = "cuda:0"
    side = torch.cuda.Stream(device=dev)
    x = torch.randn(1 << 20, device=dev)
    y = torch.randn(1 << 20, device=dev)
    out = {}

    def body():
        main = torch.cuda.current_stream()
        a = x * 2.0
        if name.startswith("P1"):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                b = y * 3.0
            main.wait_stream(side)
            c = a + b
            main.wait_stream(side)
            out["r"] = c + 1
        elif name.startswith("P2"):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                b = y * 3.0
            a2 = a + 1
            side.wait_stream(main)
            with torch.cuda.stream(side):
                b2 = b + a2
            main.wait_stream(side)
            out["r"] = b2 + a
        elif name.startswith("P3"):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                b = y * 3.0
            main.wait_stream(side)
            c = a + b
            side.wait_stream(main)
            with torch.cuda.stream(side):
                d = c * 2
            main.wait_stream(side)
            out["r"] = d + 1
        elif name.startswith("P4"):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                b = y * 3.0
            main.wait_stream(side)
            c = a + b
            e = c * 5
            side.wait_stream(main)
            with torch.cuda.stream(side):
                d = e * 2
            f = e + 1
            main.wait_stream(side)
            out["r"] = d + f
        elif name.startswith("P5"):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                b = y * 3.0
                t = b * 2          # temporary allocated on the side stream and freed during capture
                b = t + 1
                del t
                t2 = b * 4
                b = t2 - 1
                del t2
            main.wait_stream(side)
            out["r"] = a + b
        elif name.startswith("P6"):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                b = y * 3.0
            main.wait_stream(side)
            c = a + b
            del b                  # side-allocated tensor dies on the main stream's watch
            d = c * 2
            main.wait_stream(side)
            out["r"] = d + 1
        elif name.startswith("P7"):
            c = a + 1
            side.wait_stream(main)
            with torch.cuda.stream(side):
                b = c * 3.0
                b = b + 1
            d = c * 2              # main goes on
            d = d + 5
            main.wait_stream(side)
            out["r"] = d + b

    for _ in range(2):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="relaxed"):
        body()
    g.replay()
    torch.cuda.synchronize()
    print(name, "ok", float(out["r"].sum()))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for p in PATTERNS:
            r = subprocess.run([sys.executable, __file__, p], capture_output=True, text=True)
            print(p, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1][:80])
