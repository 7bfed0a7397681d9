#!/usr/bin/env python3
"""mlp_pred[0] (2048 x 32768) at B rows: the weight-stream launch vs the fp32 split-K launch.   python tools/wstream_micro.py [--rows 64]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egorear_amd import engine, hip

ap = argparse.ArgumentParser(); ap.add_argument("--rows", type=int, nargs="*", default=[64, 32, 1]); ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
N, K = 2048, 32768
w = torch.randn(N, K, device="cuda") / K ** 0.5
bias = torch.randn(N, device="cuda")
img, ds = hip.pack_wstream(w)
p = engine.pack_linears([(w, bias)])
st = engine.State(torch.device("cuda"))
ws = st.workspace


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


for B in a.rows:
    x = torch.randn(B, K, device="cuda")
    rec = torch.zeros(64, dtype=torch.int32, device="cuda"); hip.absmax_record(x, rec)
    xr = x.clone(); xr._egr_amax = rec
    t0 = timed(lambda: engine.linear(st, x, p, 2))
    t1 = timed(lambda: hip.linear_wstream(xr, img, ds, bias, 2, ws))
    print(f"rows {B:3d}: fp32 split-K {t0:7.1f} us ({4.0 * N * K / t0 / 1e6:5.2f} TB/s)   weight stream {t1:7.1f} us ({4.0 * N * K / t1 / 1e6:5.2f} TB/s)", flush=True)
