#!/usr/bin/env python
"""Idle time of the GPU inside a bench step, from a rocprofv3 --kernel-trace CSV: the union of the kernels' intervals over the last
step (from its k_stencil_march to its k_twi) and the largest gaps with the kernels on either side.   step_gaps.py <trace dir> [n]"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
starts = [i for i, r in enumerate(rows) if 'k_stencil_march' in r[2]]
ends = [i for i, r in enumerate(rows) if r[2].startswith('k_twi') or '::k_twi' in r[2]]
i0, i1 = starts[-1], ends[-1]
if i0 > i1:
    i0 = [s for s in starts if s < i1][-1]
seg = rows[i0:i1 + 1]
t0, t1 = seg[0][0], max(r[1] for r in seg)
busy, cur_end, gaps = 0, seg[0][0], []
prev = seg[0]
for s, e, name in seg:
    if s > cur_end:
        gaps.append((s - cur_end, prev[2], name))
        busy += 0
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
        prev = (s, e, name)
print('step window %.3f ms, GPU busy %.3f ms, idle %.3f ms in %d gaps' % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps)))


def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:44]


for g, a, b in sorted(gaps, reverse=True)[:top]:
    print('%8.1f us  after %-44s before %s' % (g / 1e3, short(a), short(b)))
