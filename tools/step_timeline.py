#!/usr/bin/env python3
"""The kernels of the LAST bench step in a rocprofv3 kernel trace, in order, with the idle time before each: where the step's
wall time goes between the kernels.  python tools/step_timeline.py <dir given to rocprofv3 -d>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("sh::", "").split("(")[0])
            for r in csv.DictReader(open(f)))
builds = [i for i, e in enumerate(ev) if e[2].startswith("backsolve_kernel")]
a = builds[-1]
while a > 0 and ev[a][0] - ev[a - 1][1] < 200000 and not ev[a - 1][2].startswith("bs_expand"):
    a -= 1
b = max(i for i, e in enumerate(ev) if e[2].startswith("bs_expand"))
prev_end = ev[a - 1][1] if a > 0 else ev[a][0]
print("%-40s %10s %10s" % ("kernel", "idle_us", "run_us"))
for s, e, k in ev[a:b + 1]:
    print("%-40s %10.1f %10.1f" % (k[:40], (s - prev_end) / 1e3, (e - s) / 1e3))
    prev_end = max(prev_end, e)
print("step: %.1f us from the first kernel to the last, %.1f us in kernels" % ((ev[b][1] - ev[a][0]) / 1e3, sum(e - s for s, e, _ in ev[a:b + 1]) / 1e3))
if len(builds) >= 2:
    print("previous build -> this build: %.1f us" % ((ev[builds[-1]][0] - ev[builds[-2]][0]) / 1e3))
