#!/usr/bin/env python3
"""Kernels of ONE optimizer step from a rocprofv3 kernel trace (…_kernel_trace.csv): the launches between two consecutive
adamw_kernel dispatches in the middle of the run -- exact count, busy time, idle gaps: python scripts/step_kernels.py trace.csv"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
k = len(idx) // 2
seg = rows[idx[k] + 1:idx[k + 1] + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
gaps = [max(0, int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"])) for i in range(len(seg) - 1)]
print(f"step {k} of {len(idx)}: {len(seg)} kernels, span {(t1 - t0) / 1e3:.1f} us, sum of durations {busy / 1e3:.1f} us, idle between kernels {sum(gaps) / 1e3:.1f} us")
names = collections.Counter(r["Kernel_Name"].split("(")[0][:70] for r in seg)
for n, c in names.most_common(12):
    print(f"  {c:4d}  {n}")
