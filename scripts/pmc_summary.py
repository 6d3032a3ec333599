#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (collected separately, as MI355X_MICROARCH.md section HBM
prescribes) into per-kernel HBM bytes per launch.

gfx950 corrections applied (guide): FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream, i.e. half the
bytes - doubled here; WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Units of both counters: KiB.

usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.csv> [<out.json>]"""
import collections
import csv
import json
import re
import sys


def agg(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        d[m.group(1) if m else r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
    return d


def main():
    f, w = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE")
    rows = []
    for n in sorted(set(f) | set(w)):
        fv, wv = f.get(n, [0.0]), w.get(n, [0.0])
        fa, wa = sum(fv) / len(fv), sum(wv) / len(wv)
        rows.append((n, len(fv), fa, wa, (2 * fa + wa) * 1024))
    with open(sys.argv[3], "w") as out:
        out.write("kernel,launches,FETCH_SIZE_KiB_avg,WRITE_SIZE_KiB_avg,hbm_bytes_per_launch_corrected\n")
        for r in rows:
            out.write(f"{r[0]},{r[1]},{r[2]:.1f},{r[3]:.1f},{r[4]:.0f}\n")
    if len(sys.argv) > 4:
        lin = next(r for r in rows if r[0] == "ba_linearize_kernel")
        json.dump({"ba_linearize_hbm_bytes_per_launch": round(lin[4]),
                   "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                             "(gfx950: FETCH_SIZE reports half of a wide coalesced read)",
                   "fetch_size_kib": lin[2], "write_size_kib": lin[3], "workload": "bench.py --config 3 (1M observations)"},
                  open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
