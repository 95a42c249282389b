#!/bin/bash
# spasm_hip_echelonize on a stand-in under rocprofv3 (kernel trace + stats): which kernels the two calls of
# tools/probe_e2e_verbose.py spend their device time in.  Writes gpurun_out/prof_e2e_<tag>_<name>/summary.txt.
set -u
TAG=${1:-r05}
NAME=${2:-ch8-8.b5}
OUT=gpurun_out/prof_e2e_${TAG}_$NAME
cd "$(dirname "$0")/.." || exit 1
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/probe_e2e_verbose.py $NAME 1 > $OUT/run.log 2> $OUT/trace.log
python3 - "$OUT" "$NAME" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, name = sys.argv[1], sys.argv[2]
ktime = defaultdict(list)
t_lo, t_hi = None, None
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("sh::", "").split("(")[0]
        ktime[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = ["rocprofv3 --kernel-trace of tools/probe_e2e_verbose.py %s (two calls of spasm_hip_echelonize)" % name,
         "%-72s %7s %12s %10s" % ("kernel", "calls", "total_us", "avg_us")]
tot = sum(sum(v) for v in ktime.values())
for k, v in sorted(ktime.items(), key=lambda kv: -sum(kv[1]))[:25]:
    lines.append("%-72s %7d %12.1f %10.1f" % (k[:72], len(v), sum(v), sum(v) / len(v)))
lines.append("all kernels: %.1f ms over both calls, %d launches" % (tot / 1e3, sum(len(v) for v in ktime.values())))
lines += [l.rstrip()[:400] for l in open(os.path.join(out, "run.log")) if l.startswith("rank")]
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
