#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch of
each counter for each kernel.  python profiles/show_pmc.py <dir-or-csv>..."""
import collections
import csv
import glob
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in sys.argv[1:]:
    files = [p] if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            per[(r["Kernel_Name"][:28], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (k, d, c), v in per.items():
            acc[k][c].append(v)
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:28s} n={len(v):3d} mean={sum(v) / len(v):.4g}")
