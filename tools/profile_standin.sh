#!/bin/bash
# kernel trace of one spasm_hip_echelonize call on a GL7d19-class stand-in (tools/probe_standins.py)
set -u
NAME=${1:-ch8-8.b5}
OUT=gpurun_out/prof_standin
cd "$(dirname "$0")/.." || exit 1
mkdir -p $OUT
rm -rf $OUT/trace
export TMPDIR=/tmp
export PROBE_CALLS=1
export SPASM_HIP_EXPERIMENT=1
export SPASM_HIP_RREF_TIMING=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/probe_standins.py $NAME > $OUT/out.txt 2> $OUT/err.txt
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $OUT/kernel_stats_head.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-78s %7s %11s %11s %6s" % ("kernel", "calls", "total_ms", "avg_us", "%"))
for r in rows[:16]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)[:78]
    print("%-78s %7d %11.3f %11.1f %6.2f" % (name, int(r["Calls"]), int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
cat $OUT/kernel_stats_head.txt
grep -v "^\[rref/hip\]" $OUT/out.txt | tail -30
