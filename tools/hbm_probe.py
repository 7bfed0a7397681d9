#!/usr/bin/env python3
"""What the HBM system of this box sustains for pure streaming: write-only (fill), copy (read + write), read-only (sum), with
torch's own elementwise kernels on 2 GiB tensors.   python tools/hbm_probe.py"""
import torch
n = 1 << 29      # floats: 2 GiB
a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")


def timed(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


t = timed(lambda: a.fill_(1.0)); print(f"write-only  fill_  : {4 * n / t / 1e9:6.2f} TB/s ({t:.3f} ms)")
t = timed(lambda: b.copy_(a)); print(f"read + write copy_ : {8 * n / t / 1e9:6.2f} TB/s total ({t:.3f} ms)")
t = timed(lambda: a.sum()); print(f"read-only   sum    : {4 * n / t / 1e9:6.2f} TB/s ({t:.3f} ms)")
t = timed(lambda: torch.add(a, 1.0, out=b)); print(f"read + write add   : {8 * n / t / 1e9:6.2f} TB/s total ({t:.3f} ms)")
