#!/bin/bash
# The device pivot search under rocprofv3 (kernel trace + stats): tools/probe_pivot_labels.py on a stand-in (mk15.b4 by
# default) runs spasm_hip_pivots_extract_structural three times -- labelled search, ticket search alone, labelled search --
# so the trace holds both kernels on the same matrix.  Writes gpurun_out/prof_pivots_<tag>/summary.txt; copy it to
# profiles/<tag>_pivot_search_summary.txt.
set -u
TAG=${1:-r05}
NAME=${2:-mk15.b4}
OUT=gpurun_out/prof_pivots_$TAG
cd "$(dirname "$0")/.." || exit 1
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/probe_pivot_labels.py $NAME > $OUT/run.log 2> $OUT/trace.log
python3 - "$OUT" "$NAME" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, name = sys.argv[1], sys.argv[2]
ktime = defaultdict(list)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ktime[r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("sh::", "").split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
lines = ["rocprofv3 --kernel-trace of tools/probe_pivot_labels.py %s (labelled search, ticket search alone, labelled search)" % name,
         "%-64s %6s %12s %12s" % ("kernel", "calls", "total_us", "avg_us")]
for k, v in sorted(ktime.items(), key=lambda kv: -sum(kv[1]))[:12]:
    lines.append("%-64s %6d %12.1f %12.1f" % (k[:64], len(v), sum(v), sum(v) / len(v)))
lines.append("")
lines += [l.rstrip()[:1200] for l in open(os.path.join(out, "run.log")) if l.startswith("==") or "[pivots] device" in l]
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:16]))
PY
