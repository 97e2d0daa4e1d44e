#!/usr/bin/env python
"""Summarise k_sweep_round launch durations / gaps of the LAST step in a rocprofv3 kernel_trace.csv."""
import csv
import glob
import sys
import numpy as np
f = sorted(glob.glob(sys.argv[1] + '/*/*kernel_trace.csv'))[-1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = [r for r in csv.DictReader(open(f)) if 'k_sweep_round' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows) // steps
last = rows[(steps - 1) * n:]
dur = np.array([(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in last])
gap = np.array([(int(last[i + 1]['Start_Timestamp']) - int(last[i]['End_Timestamp'])) / 1e3 for i in range(len(last) - 1)])
print(n, 'launches; kernel %.2f ms; gaps %.2f ms' % (dur.sum() / 1e3, gap.sum() / 1e3))
for a, b in [(0, 10), (10, 50), (50, 100), (100, 200), (200, 300), (300, 500), (500, n)]:
    if a < n:
        print('%4d-%4d kernel %6.2f ms  gaps %5.2f ms  avg dur %7.1f us  avg gap %5.1f us' % (
            a, min(b, n), dur[a:b].sum() / 1e3, gap[a:b].sum() / 1e3, dur[a:b].mean(), gap[a:min(b, len(gap))].mean()))
