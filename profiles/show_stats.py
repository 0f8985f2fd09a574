#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: python profiles/show_stats.py <csv>"""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:44]:44s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs']) / 1e3:10.1f} "
          f"total_ms={float(r['TotalDurationNs']) / 1e6:8.2f}")
