#!/usr/bin/env python3
"""Reduce two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) of
`python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline` into HBM bytes per launch of the conv kernel.
    python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [batch]
gfx950 corrections (guide §HBM): FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads -> x2; both are in KB."""
import collections, csv, glob, json, sys

def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        nm = r["Kernel_Name"]
        # conv_igemm_x6 / x6p / tap / tap2 / conv_pw_x6 kernels = the split-bf16 launches, conv_igemm_kernel the fp32 ones
        k = ("conv_x6" if ("conv_igemm_x6" in nm or "conv_igemm_tap" in nm or "conv_pw_x6" in nm) else "conv") if ("conv_igemm" in nm or "conv_pw_x6" in nm) else "other"
        per[k] += float(r["Counter_Value"]); n[(k, r["Dispatch_Id"])] += 1
    launches = collections.Counter(k for (k, _d) in n)
    return per, launches

fetch, lf = load(sys.argv[1], "FETCH_SIZE")
write, lw = load(sys.argv[2], "WRITE_SIZE")
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 64
def entry(key, label):
    n = lf[key]
    if not n:
        return None
    return {"kernel": label, "launches_counted": n, "fetch_bytes_per_launch_raw_x1024": fetch[key] * 1024 / n, "fetch_correction": 2.0,
            "write_bytes_per_launch": write[key] * 1024 / lw[key],
            "hbm_bytes_per_launch": (2.0 * fetch[key] * 1024) / n + write[key] * 1024 / lw[key]}

x6 = entry("conv_x6", "conv_igemm_x6 / x6p / tap / tap2 / conv_pw_x6 kernels (egr_conv2d_nhwc_f32, EGR_W_BF16X3 launches)")
f32 = entry("conv", "conv_igemm_kernel (egr_conv2d_nhwc_f32, fp32-matrix-core launches)")
main = x6 or f32
out = {"batch": batch, **main, "by_format": {"bf16x3": x6, "f32": f32},
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-train"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
