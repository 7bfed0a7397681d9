#!/usr/bin/env python3
"""Reduce two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) of
`python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline` into HBM bytes per launch of the conv kernel.
    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [batch] [forwards] [commit]
`forwards` = forwards the profiled command runs (12 for the command above: first + warm-up + 2 timed + the instrumented one + 7 of the
from-raw-frames leg) -> launches_per_forward, which bench.py compares with its own run before it reports the figure.
gfx950 corrections (guide §HBM): FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads -> x2; both are in KB."""
import collections, csv, glob, json, re, sys

def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        nm = r["Kernel_Name"]
        # conv_igemm_x6 / x6p / tap / tap2 / conv_pw_x6 kernels = the split launches (last template argument: 3 = bf16 x 3 planes,
        # 2 = fp16 x 2 planes), conv_igemm_kernel the fp32 ones
        tapx = "conv_tapx_kernel" in nm or "conv_pw_chain_kernel" in nm or "conv_pw2_kernel" in nm    # role-split persistent workgroups / chained 1x1 pairs: fp16 scheme only
        split = "conv_igemm_x6" in nm or "conv_igemm_tap" in nm or "conv_pw_x6" in nm or tapx
        h2 = tapx or (split and re.search(r",\s*2>\(", nm.replace(") ", ")")) is not None)
        k = (("conv_h2" if h2 else "conv_x6") if split else "conv") if ("conv_igemm" in nm or "conv_pw_x6" in nm or tapx) else "other"
        per[k] += float(r["Counter_Value"]); n[(k, r["Dispatch_Id"])] += 1
    launches = collections.Counter(k for (k, _d) in n)
    return per, launches

fetch, lf = load(sys.argv[1], "FETCH_SIZE")
write, lw = load(sys.argv[2], "WRITE_SIZE")
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 64
forwards = int(sys.argv[5]) if len(sys.argv) > 5 else 12
commit = sys.argv[6] if len(sys.argv) > 6 else None
def entry(key, label):
    n = lf[key]
    if not n:
        return None
    return {"kernel": label, "launches_counted": n, "launches_per_forward": n // forwards if n % forwards == 0 else n / forwards,
            "fetch_bytes_per_launch_raw_x1024": fetch[key] * 1024 / n, "fetch_correction": 2.0,
            "write_bytes_per_launch": write[key] * 1024 / lw[key],
            "hbm_bytes_per_launch": (2.0 * fetch[key] * 1024) / n + write[key] * 1024 / lw[key]}

h2 = entry("conv_h2", "conv_tapx / conv_igemm_x6 / x6p / tap / tap2 / conv_pw_x6 kernels, two fp16 planes (egr_conv2d_nhwc_ex_f32, EGR_W_F16X2 launches)")
x6 = entry("conv_x6", "conv_igemm_x6 / x6p / tap / tap2 / conv_pw_x6 kernels, three bf16 planes (egr_conv2d_nhwc_f32, EGR_W_BF16X3 launches)")
f32 = entry("conv", "conv_igemm_kernel (egr_conv2d_nhwc_f32, fp32-matrix-core launches)")
main = h2 or x6 or f32
out = {"batch": batch, "forwards_profiled": forwards, "commit": commit, **main, "by_format": {"f16x2": h2, "bf16x3": x6, "f32": f32},
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-train"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
