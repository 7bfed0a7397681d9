#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel: python tools/pmc_show.py DIR [substring]"""
import csv, collections, glob, sys
sub = sys.argv[2] if len(sys.argv) > 2 else "conv_igemm"
for f in sorted(glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set); dur = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "")[:60]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in cnt[k]:
            cnt[k].add(r["Dispatch_Id"]); dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k, v in per.items():
        if sub in k:
            print(f"{k}  launches {len(cnt[k])}  avg {dur[k] / len(cnt[k]) / 1e3:.1f} us")
            wc = v.get("SQ_WAVE_CYCLES", 0)
            for c, x in sorted(v.items()):
                print(f"    {c:28s} {x:16.0f}" + (f"  {x / wc:.3f} of WAVE_CYCLES" if wc and c.startswith("SQ_") else ""))
