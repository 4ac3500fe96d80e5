#!/usr/bin/env python3
"""One step of a rocprofv3 kernel trace (tools/profile_bench.sh): launches, kernel time, gaps, and launches per kernel.
usage: tools/step_launches.py <kernel_trace.csv> <substring of the kernel that starts a step>"""
import csv, sys
from collections import Counter

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [(int(r["Start_Timestamp"]) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"]) for r in rows]
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r[2]]
a, b = idx[-2], idx[-1]
step = rows[a:b]
gaps = [(step[i + 1][0] - (step[i][0] + step[i][1]), step[i][2], step[i + 1][2]) for i in range(len(step) - 1)]
gaps.append((rows[b][0] - (step[-1][0] + step[-1][1]), step[-1][2], "(next step)"))
short = lambda n: n.replace("void ", "").replace("bhip::", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
print(f"one step (second to last of the run): {len(step)} launches, span {rows[b][0] - rows[a][0]:.1f} us, kernels {sum(r[1] for r in step):.1f} us, "
      f"gaps {sum(g[0] for g in gaps):.1f} us (under rocprofv3: launches cost more than in a plain run)")
print("largest gaps (us): after -> before")
for g in sorted(gaps, reverse=True)[:12]:
    print(f"  {g[0]:7.1f}  {short(g[1])} -> {short(g[2])}")
print("launches | total us | kernel")
c = Counter(short(r[2]) for r in step)
for k, v in sorted(c.items(), key=lambda kv: -sum(r[1] for r in step if short(r[2]) == kv[0])):
    print(f"  {v:3d} | {sum(r[1] for r in step if short(r[2]) == k):8.1f} | {k}")
