#!/usr/bin/env python3
"""Steps of a rocprofv3 kernel trace as a timeline:  python profiles/show_timeline.py <kernel_trace.csv> [step [nsteps]]

A step starts at each launch of the contraction kernel (score3_kernel); printing starts at the `step`-th launch
from the end (default 2, i.e. a warmed-up one) and covers `nsteps` of them (default 1; with two lanes print
three or more from the timed region -- the last ~10 launches of a bench run are its one-lane isolation steps --
to see how the lanes' kernels interleave and wait for each other).  Per kernel: stream/queue, start and end in
microseconds after the first launch printed."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void score3_kernel") or "score3_kernel<" in r["Kernel_Name"]]
if len(starts) < back + 1:
    sys.exit("too few steps in the trace")
a, b = starts[-back - 1], starts[min(len(starts) - 1, len(starts) - back - 1 + nsteps)]
# the sparse pass of the same step is launched just before the contraction kernel: include up to 2 launches before
a0 = a
while a0 > 0 and a - a0 < 3 and "s3_t3" in rows[a0 - 1]["Kernel_Name"]:
    a0 -= 1
t0 = int(rows[a0]["Start_Timestamp"])
for r in rows[a0:b + 12]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"q{r.get('Queue_Id', '?'):>3s} {s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:60]}")
