#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: dispatches, sum, average.
usage: pmc_summary.py <counter_collection.csv | directory>... > summary.csv   (a directory is searched for *counter_collection.csv)"""
import collections
import csv
import glob
import os
import sys

paths = []
for a in sys.argv[1:]:
    paths += sorted(glob.glob(os.path.join(a, "**", "*counter_collection.csv"), recursive=True)) if os.path.isdir(a) else [a]

print("kernel,counter,dispatches,sum,avg_per_dispatch")
for path in paths:
    tot = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"].split("(")[0].replace(",", ";"), r["Counter_Name"])
        tot[k][0] += 1
        tot[k][1] += float(r["Counter_Value"])
    for (kn, cn), (n, s) in sorted(tot.items()):
        if kn.startswith("__amd_rocclr"):
            continue
        print('"%s",%s,%d,%.6g,%.6g' % (kn, cn, n, s, s / n))
