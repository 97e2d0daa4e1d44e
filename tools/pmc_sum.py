#!/usr/bin/env python
"""Totals of rocprofv3 --pmc passes per kernel (summed over dispatches and counter instances):
    pmc_sum.py <dir> [<dir> ...] > table.csv       columns: kernel (short), counter, dispatches, total"""
import csv
import glob
import re
import sys
from collections import defaultdict

tot = defaultdict(float)
ndisp = defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
            k = re.sub(r'^void ', '', k).split('(')[0]
            tot[(k, r['Counter_Name'])] += float(r['Counter_Value'])
            ndisp[(k, r['Counter_Name'])].add(r['Dispatch_Id'])
w = csv.writer(sys.stdout)
w.writerow(['kernel', 'counter', 'dispatches', 'total'])
for (k, c) in sorted(tot):
    w.writerow([k, c, len(ndisp[(k, c)]), '%.6g' % tot[(k, c)]])
