#!/usr/bin/env python3
"""Mean per-launch value of every counter in a rocprofv3 --pmc counter_collection.csv for kernels matching a substring.
usage: pmc_kernel.py <counter_collection.csv> <kernel substring>"""
import collections
import csv
import sys

d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(d):
    v = d[k]
    print(f"{k:28s} launches {len(v):4d}  mean {sum(v) / len(v):16.1f}")
