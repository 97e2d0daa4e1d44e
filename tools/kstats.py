#!/usr/bin/env python
"""Print per-kernel averages (ms) from a rocprofv3 kernel_stats.csv, optionally filtered by substrings."""
import csv
import glob
import sys
d = sys.argv[1]
pats = sys.argv[2:]
f = sorted(glob.glob(d + '/*/*kernel_stats.csv') + glob.glob(d + '/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if not pats or any(p in n for p in pats):
        short = n.replace('(anonymous namespace)::', '').split('(')[0][:60]
        print("%-60s calls %5s  avg %9.4f ms  total %9.3f ms" % (short, r['Calls'], float(r['AverageNs']) / 1e6, float(r['TotalDurationNs']) / 1e6))
