#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --stats kernel_stats.csv:  python tools/kstats.py <dir-or-csv> [rows]"""
import csv
import glob
import os
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{path}: total {tot/1e6:.2f} ms")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{r['Name'][:72]:72s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
