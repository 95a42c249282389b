#!/usr/bin/env python3
"""Condenses one tools/profile.sh pass directory: per-kernel time table from the kernel trace, FETCH_SIZE / WRITE_SIZE
per launch of each kernel, and <dir>/traffic.json in the form bench.py quotes (profiles/r02_traffic.json).

FETCH_SIZE / WRITE_SIZE are in KiB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports half of the bytes of
wide (16 B per lane) coalesced reads and is uncalibrated for other widths; these kernels read 4 B per lane, so the raw
sum is given as `bytes_per_launch` and the sum with FETCH doubled as an upper bound.

  summarize_profile.py <dir>                      summary on stdout, <dir>/traffic.json
  summarize_profile.py --merge a.json b.json      union of the `kernels` of several passes on stdout
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    """kernel name as the library reports it in spasm_hip_schur_stats.kernel"""
    name = name.replace("void ", "").replace("sh::", "").replace("(anonymous namespace)::", "")
    name = name.split("(")[0]
    return re.sub(r"\s+", "", name)[:80]


def main():
    if sys.argv[1] == "--merge":
        merged = None
        for p in sys.argv[2:]:
            if not os.path.exists(p):
                continue
            t = json.load(open(p))
            if merged is None:
                merged = t
            elif (t.get("workload"), t.get("rows")) == (merged.get("workload"), merged.get("rows")):
                for k, v in t["kernels"].items():
                    merged["kernels"].setdefault(k, v)
        print(json.dumps(merged, indent=1))
        return
    out = sys.argv[1]
    rows = defaultdict(list)
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("== kernel trace (us) ==")
    print("%-62s %8s %12s %12s %12s" % ("kernel", "calls", "total_us", "avg_us", "max_us"))
    for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        print("%-62s %8d %12.1f %12.1f %12.1f" % (k, len(v), sum(v), sum(v) / len(v), max(v)))
    per = {}
    for ctr, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write"), ("TCC_ATOMIC_sum", "pmc_atomic")):
        acc = defaultdict(list)
        for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == ctr:
                    acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        print("\n== %s per launch (%s) ==" % (ctr, "64-byte atomic requests" if ctr.startswith("TCC") else "counter units: KiB"))
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            # bench.py sizes its row pool by probing: launches that ran out of pool stop early and move fewer bytes.
            # The per-launch figure is the mean over the full-size launches (within 10 % of the largest).
            full = [x for x in v if x >= 0.9 * max(v)]
            print("%-62s launches %5d  avg %16.1f  max %16.1f  full-size %5d  avg %16.1f" % (k, len(v), sum(v) / len(v), max(v), len(full), sum(full) / len(full)))
            per.setdefault(k, {})[ctr] = sum(full) / len(full) * (1.0 if ctr.startswith("TCC") else 1024.0)
            per[k]["launches_" + ctr] = len(full)
    bench = {}
    try:
        for line in open(os.path.join(out, "bench_trace.json")):
            if line.startswith("{"):
                bench = json.loads(line)
    except (OSError, ValueError):
        pass
    cfg = bench.get("config", {})
    traffic = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `bench.py --steps 3 --warmup 1 "
                         "--no-cpu-baseline --no-extras`, tools/profile.sh; raw counter sums (KiB * 1024)",
               "workload": (cfg.get("workload") or "").split(" ")[0], "rows": cfg.get("rows_per_step"),
               "bench_line_under_trace": {k: bench.get(k) for k in ("value", "ms_per_step")},
               "kernels": {}}
    for k, v in per.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            traffic["kernels"][k] = {"fetch_bytes_raw": v["FETCH_SIZE"], "write_bytes_raw": v["WRITE_SIZE"],
                                     "bytes_per_launch": v["FETCH_SIZE"] + v["WRITE_SIZE"],
                                     "bytes_per_launch_fetch_doubled": 2 * v["FETCH_SIZE"] + v["WRITE_SIZE"],
                                     "launches": v["launches_FETCH_SIZE"], "atomic_requests_per_launch": v.get("TCC_ATOMIC_sum"),
                                     "avg_us_in_trace_full_size": (lambda t: sum(t) / len(t))([x for x in rows[k] if x >= 0.9 * max(rows[k])]) if k in rows else None}
    json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
