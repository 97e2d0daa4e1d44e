#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection CSVs into per-kernel means per dispatch.

    pmc_aggregate.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> > profiles/rNN_pmc_fetch_write_<size>.csv

Output columns: kernel, counter, dispatches, mean_per_dispatch_KB (raw counter units: KiB), which is
what bench.py's pmc_traffic() reads.  FETCH_SIZE needs the x2 correction on gfx950 (profiles/README.md)."""
import csv
import glob
import sys
from collections import defaultdict

rows = []
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = defaultdict(lambda: defaultdict(float))     # (kernel, counter) -> dispatch id -> value (summed over XCDs/instances)
        for r in csv.DictReader(open(f)):
            acc[(r['Kernel_Name'], r['Counter_Name'])][r['Dispatch_Id']] += float(r['Counter_Value'])
        for (k, c), per in acc.items():
            rows.append((k, c, len(per), sum(per.values()) / len(per)))
w = csv.writer(sys.stdout)
w.writerow(['kernel', 'counter', 'dispatches', 'mean_per_dispatch_KB'])
for k, c, n, v in sorted(rows, key=lambda x: -x[3]):
    w.writerow([k, c, n, '%.1f' % v])
