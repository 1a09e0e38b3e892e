#!/usr/bin/env python3
"""Mean per-dispatch counter values of kernels whose name contains a substring, from rocprofv3 --pmc output directories.
    python tools/pmc_summary.py <substring> <dir> [<dir> ...]"""
import collections
import csv
import glob
import os
import sys

sub = sys.argv[1]
for d in sys.argv[2:]:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                a = agg[r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
        for k, (n, v) in sorted(agg.items()):
            print(f"{os.path.basename(d)} {k} n={n} mean={v / n:.6g}")
