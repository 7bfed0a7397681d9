#!/usr/bin/env python3
"""Per-kernel MFMA-pipe utilisation and effective clock from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass."""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); seen = set(); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"][:60]
    per[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[n] += 1
for n in per:
    busy, gui = per[n]["SQ_VALU_MFMA_BUSY_CYCLES"], per[n]["GRBM_GUI_ACTIVE"]
    cyc = gui / 8.0
    print(f"{n:60s} launches {cnt[n]:4d}  avg {dur[n]/cnt[n]/1e3:9.1f} us  mfma_util {busy/(cyc*1024.0+1e-9):.3f}  clock {cyc/max(dur[n],1):.2f} GHz")
