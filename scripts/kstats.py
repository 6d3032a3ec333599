"""Print a rocprofv3 *_kernel_stats.csv compactly: python scripts/kstats.py <csv> [rows]."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for r in rows[:limit]:
    m = re.search(r"(\w+_kernel|__amd_\w+)", r["Name"])
    name = m.group(1) if m else r["Name"][:40]
    print(f"{name:32s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e3:8.2f} us  total {float(r['TotalDurationNs']) / 1e3:9.1f} us"
          f"  min {float(r['MinNs']) / 1e3:7.2f}  max {float(r['MaxNs']) / 1e3:7.2f}")
