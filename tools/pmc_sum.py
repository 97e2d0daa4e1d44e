#!/usr/bin/env python
"""Sum rocprofv3 --pmc counters per kernel over ALL dispatches:  pmc_sum.py <dir> [<dir> ...]"""
import csv
import glob
import sys
from collections import defaultdict
acc = defaultdict(float)
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r['Kernel_Name'][:60], r['Counter_Name'])] += float(r['Counter_Value'])
for (k, c), v in sorted(acc.items()):
    if v > 0:
        print('%-62s %-24s %16.0f' % (k, c, v))
