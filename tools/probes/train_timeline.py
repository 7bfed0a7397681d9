#!/usr/bin/env python3
"""Where the captured training step's wall-clock goes: from a rocprofv3 kernel trace of `tools/train_bench.py --graph`, the LAST step's
kernels as intervals - per hardware queue busy time, the time no kernel runs at all, the time exactly one small (< 60 us) kernel runs alone.
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/train_bench.py --batch 32 --graph --steps 4 --warmup 3
    python tools/probes/train_timeline.py DIR"""
import collections, csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "stem_wgrad_kernel" in r["Kernel_Name"]]
# the instrumented eager step comes last in train_bench: take the last GRAPH step = between the 3rd and 2nd last stem weight gradients
a, b = marks[-3] + 1, marks[-2] + 1
step = rows[a:b]
t0 = min(int(r["Start_Timestamp"]) for r in step); t1 = max(int(r["End_Timestamp"]) for r in step)
ev = []
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    small = (e - s) < 60000
    ev.append((s, 1, small)); ev.append((e, -1, small))
ev.sort()
busy = idle = alone_small = 0
n = ns = 0
last = t0
for t, d, small in ev:
    dt = t - last
    if n == 0: idle += dt
    else:
        busy += dt
        if n == 1 and ns == 1: alone_small += dt
    n += d; ns += d if small else 0
    last = t
q = collections.Counter(); qn = collections.Counter()
for r in step:
    q[r.get("Queue_Id", "?")] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); qn[r.get("Queue_Id", "?")] += 1
print(f"step span {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels; some kernel running {busy / 1e6:.3f} ms, NO kernel running {idle / 1e6:.3f} ms, "
      f"one small (< 60 us) kernel running alone {alone_small / 1e6:.3f} ms")
for k, v in q.most_common():
    print(f"  queue {k}: {qn[k]} kernels, {v / 1e6:.3f} ms of kernel time")
