#!/usr/bin/env python3
"""Mean per launch of every counter of one kernel in a rocprofv3 --pmc counter_collection.csv.
usage: pmc_kernel.py <counter_collection.csv> <kernel name substring>"""
import collections
import csv
import sys

d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(d):
    print(f"{k:32s} {sum(d[k]) / len(d[k]):16.0f}   ({len(d[k])} launches)")
