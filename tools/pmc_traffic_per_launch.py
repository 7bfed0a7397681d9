#!/usr/bin/env python3
"""HBM traffic of every implicit-GEMM launch of ONE forward against its algorithmic bytes (input once + output once + weights once), per
launch shape: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; FETCH x2 on gfx950) of the eager bench command, joined by dispatch
order with the tags of tools/launch_list.py (same launch sequence).
    python tools/pmc_traffic_per_launch.py <fetch_dir> <write_dir> <launch_list.txt> [forwards=12] > out.txt"""
import collections, csv, glob, re, sys


def per_dispatch(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    val = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        nm = r["Kernel_Name"]
        if r["Counter_Name"] != counter or not ("conv_igemm" in nm or "conv_pw_x6" in nm or "conv_tapx_kernel" in nm or "conv_pw_chain_kernel" in nm or "conv_pw2_kernel" in nm or "linear_small_kernel" in nm):
            continue
        k = int(r["Dispatch_Id"])
        val[k] = val.get(k, 0.0) + float(r["Counter_Value"])
    return [v * 1024.0 for _, v in sorted(val.items())]          # bytes, dispatch order


fetch, write = per_dispatch(sys.argv[1], "FETCH_SIZE"), per_dispatch(sys.argv[2], "WRITE_SIZE")
forwards = int(sys.argv[4]) if len(sys.argv) > 4 else 12
tags = []
for l in open(sys.argv[3]):
    m = re.match(r"\s*\d+\s+([\d.]+) us\s+(?:egr_conv2d_nhwc_f32|egr_conv1x1_chain_f32)\s+(.*)", l)
    if m:
        tags.append((float(m.group(1)), m.group(2).strip()))
per = len(tags)
assert len(fetch) == len(write) and len(fetch) % per == 0 and len(fetch) // per == forwards, (len(fetch), len(write), per, forwards)
print(f"{per} implicit-GEMM launches per forward, {forwards} forwards profiled; forwards 2-5 (the timed ones) averaged; FETCH_SIZE x 2 (gfx950)")
print(f"{'#':>3s} {'alg MB':>8s} {'fetch MB':>9s} {'write MB':>9s} {'HBM/alg':>8s} {'us':>7s}  launch")
tot_a = tot_h = 0.0
for i, (us, tag) in enumerate(tags):
    m = re.search(r"G(\d+) M(\d+) N(\d+) K(\d+) k(\d)s(\d) cin(\d+)", tag)
    G, M, N, K, k, s_, cin = map(int, m.groups())
    mid = re.search(r" mid(\d+)", tag)           # chained 1x1 pair: both weight matrices, no intermediate tensor
    alg = 4.0 * G * (M * N + M * s_ * s_ * cin + N * K + (int(mid.group(1)) * (K + N) - N * K if mid else 0))
    fs = [2.0 * fetch[f_ * per + i] for f_ in range(2, 6)]
    ws = [write[f_ * per + i] for f_ in range(2, 6)]
    fm, wm = sum(fs) / len(fs), sum(ws) / len(ws)
    tot_a += alg
    tot_h += fm + wm
    print(f"{i:3d} {alg / 1e6:8.1f} {fm / 1e6:9.1f} {wm / 1e6:9.1f} {(fm + wm) / alg:8.2f} {us:7.1f}  {tag}")
print(f"sum {tot_a / 1e6:8.1f} MB algorithmic, {tot_h / 1e6:.1f} MB measured: {tot_h / tot_a:.3f}")
