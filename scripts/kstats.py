#!/usr/bin/env python3
"""Short table of a rocprofv3 *_kernel_stats.csv: calls and average duration per kernel (name shortened).
usage: kstats.py <kernel_stats.csv> [substring ...]"""
import csv
import re
import sys

for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+)_kernel", r["Name"])
    name = (m.group(1) if m else r["Name"][:40]) + ("<" + re.search(r"<(\d+)>", r["Name"]).group(1) + ">" if re.search(r"<(\d+)>", r["Name"]) else "")
    if len(sys.argv) > 2 and not any(s in name for s in sys.argv[2:]):
        continue
    print(f"{name:28s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:8.2f} us  total {float(r['TotalDurationNs']) / 1e6:8.3f} ms  {float(r['Percentage']):5.1f} %")
