#!/usr/bin/env python3
"""Where the queue ran empty: gaps (> 1 us) between consecutive kernels of a rocprofv3 kernel trace, by (previous, next) kernel.
usage: trace_gaps.py <kernel_trace.csv> [from_fraction to_fraction]   (the fractions select a slice of the run, default 0 1)"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0


def nm(r):
    m = re.search(r"(\w+)_kernel", r["Kernel_Name"])
    return m.group(1) if m else r["Kernel_Name"][:30]


prev = None
gaps = {}
for r in rows[int(lo * len(rows)):int(hi * len(rows))]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev is not None:
        g = (s - prev[1]) / 1e3
        if g > 1.0:
            gaps.setdefault((prev[0], nm(r)), []).append(g)
    prev = (nm(r), e)
out = []
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    out.append(f"{k[0]:>24s} -> {k[1]:24s} {len(v):4d} x  avg {sum(v) / len(v):9.1f} us")
sys.stdout.write("\n".join(out[:12]) + "\n")
