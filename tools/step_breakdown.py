#!/usr/bin/env python3
"""Developer tool: fold a rocprofv3 kernel trace (csv) of bench.py into ONE steady-state step -- the launches between the last two
stem kernels -- and print per-kernel time and the idle gaps between launches.
    python tools/step_breakdown.py <kernel_trace.csv> [marker substring, default stem_s2]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "stem_s2"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = marks[-2], marks[-1]
step = rows[a:b]
t0, t1 = int(step[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
busy = defaultdict(lambda: [0, 0.0])
gap = 0.0
prev_end = None
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    k = r["Kernel_Name"]
    k = k[k.find("::") + 2:] if k.startswith("void (anonymous") else k
    busy[k[:90]][0] += 1
    busy[k[:90]][1] += (e - s) / 1e3
    if prev_end is not None and s > prev_end:
        gap += (s - prev_end) / 1e3
    prev_end = max(prev_end or 0, e)
print(f"step: {len(step)} launches, {(t1 - t0) / 1e6:.3f} ms wall, {sum(v[1] for v in busy.values()) / 1e3:.3f} ms in kernels, {gap / 1e3:.3f} ms idle between launches")
for k, (n, us) in sorted(busy.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{us / 1e3:8.3f} ms {n:4d}  {k}")
