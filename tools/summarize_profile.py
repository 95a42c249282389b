#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory: per-kernel time table from the kernel trace and
FETCH_SIZE / WRITE_SIZE per launch of each kernel (raw counter values; see DESIGN.md for the gfx950
correction of FETCH_SIZE)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    return name.split("(")[0].replace("void ", "").replace("sh::", "")[:60]


rows = defaultdict(list)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("== kernel trace (us) ==")
print("%-62s %8s %12s %12s %12s" % ("kernel", "calls", "total_us", "avg_us", "max_us"))
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print("%-62s %8d %12.1f %12.1f %12.1f" % (k, len(v), sum(v), sum(v) / len(v), max(v)))

for ctr, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == ctr:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    print("\n== %s per launch (counter units: KiB) ==" % ctr)
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print("%-62s launches %5d  avg %16.1f  max %16.1f" % (k, len(v), sum(v) / len(v), max(v)))
