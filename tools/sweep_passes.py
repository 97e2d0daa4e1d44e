#!/usr/bin/env python
"""Per-pass durations and gaps of the tile sweep (k_sweep_tiles*, k_tiles_of_frontier) of the LAST step in a
rocprofv3 kernel_trace.csv:  python tools/sweep_passes.py <rocprof dir> [steps]"""
import csv
import glob
import sys
f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sw = [i for i, r in enumerate(rows) if any(k in r['Kernel_Name'] for k in ('k_sweep_tiles', 'k_sweep_first', 'k_sweep_dense', 'k_compact_tiles', 'k_compact_build', 'k_tile_scan', 'k_sweep_sym', 'k_sym_finish'))]
n = len(sw) // steps
last = sw[(steps - 1) * n:]
t0 = int(rows[last[0]]['Start_Timestamp'])
prev_end = t0
tot = 0
for q, i in enumerate(last):
    r = rows[i]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    g = int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0)
    print('pass %3d  start %8.1f us  dur %8.1f us  gap %6.1f us  grid %8d  %s' % (q + 1, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, g, r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '')[:28]))
    prev_end = e
    tot += e - s
print('passes %d  kernel time %.2f ms  wall %.2f ms' % (n, tot / 1e6, (prev_end - t0) / 1e6))
