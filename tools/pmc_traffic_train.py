#!/usr/bin/env python3
"""HBM bytes per launch of the training step's kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, as
MI355X_MICROARCH.md prescribes; FETCH_SIZE x2 on gfx950) of `python3 tools/train_bench.py --batch 32 --steps 2 --warmup 1` (eager).
    python tools/pmc_traffic_train.py <fetch_dir> <write_dir> <out.json> [batch] [commit]"""
import collections, csv, glob, json, re, sys


def _planes(nm: str) -> str:
    """The split kernels carry their plane count as the last template argument: 2 = the fp16 scheme, 3 = the bf16 scheme."""
    m = re.search(r"<([^<>]*)>", nm)
    last = m.group(1).split(",")[-1].strip() if m else "3"
    return "f16x2" if last == "2" else "bf16x3"


def classify(nm: str) -> str:
    if "conv_tapx" in nm or "conv_pw_chain" in nm or "conv_pw2" in nm: return "conv_igemm_f16x2"      # role-split / chained kernels: fp16 scheme only
    if "conv_igemm_x6" in nm or "conv_igemm_tap" in nm or "conv_pw_x6" in nm: return "conv_igemm_" + _planes(nm)
    if "conv_igemm" in nm: return "conv_igemm_f32"
    if "conv_wgrad_x6" in nm or "conv_wgrad3_x6" in nm: return "conv_wgrad_" + _planes(nm)
    if "conv_wgrad" in nm: return "conv_wgrad_f32"
    for k in ("bn_partial", "bn_bwd_apply", "bn_stats", "scale_shift", "adamw", "msda_gather_bwd", "stem_wgrad", "stem_x6_kernel", "stem_kernel", "repack", "upsample2x_bwd"):
        if k in nm: return k
    return "other"


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(float); seen = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        k = classify(r["Kernel_Name"])
        per[k] += float(r["Counter_Value"]); seen[k].add(r["Dispatch_Id"])
    return per, {k: len(v) for k, v in seen.items()}


fetch, lf = load(sys.argv[1], "FETCH_SIZE")
write, lw = load(sys.argv[2], "WRITE_SIZE")
out = {"batch": int(sys.argv[4]) if len(sys.argv) > 4 else 32, "commit": sys.argv[5] if len(sys.argv) > 5 else None, "fetch_correction": 2.0, "kernels": {},
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 tools/train_bench.py --batch 32 --steps 2 --warmup 1"}
for k in sorted(lf, key=lambda k: -(2 * fetch[k] + write.get(k, 0))):
    n, nw = lf[k], lw.get(k, 0)
    out["kernels"][k] = {"launches_counted": n, "fetch_bytes_per_launch_raw_x1024": fetch[k] * 1024 / n,
                         "write_bytes_per_launch": write.get(k, 0.0) * 1024 / max(nw, 1),
                         "hbm_bytes_per_launch": 2.0 * fetch[k] * 1024 / n + write.get(k, 0.0) * 1024 / max(nw, 1)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out["kernels"].items()}))
