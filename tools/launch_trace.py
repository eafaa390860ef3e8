#!/usr/bin/env python3
"""Per-launch start and duration of one frame from a rocprofv3 --kernel-trace run:
   python tools/launch_trace.py <trace dir> <first launch> <count>"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [r for r in rows if any(k in r["Kernel_Name"] for k in ("intersect_kernel", "shade_kernel", "resolve_kernel"))]
first, count = int(sys.argv[2]), int(sys.argv[3])
fr = ks[first:first + count]
t0 = int(fr[0]["Start_Timestamp"])
for r in fr:
    n = r["Kernel_Name"]
    n = n[n.index("::") + 2:][:36]
    print(f"{n:38s} start {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  duration {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us")
