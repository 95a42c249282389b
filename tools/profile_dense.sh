#!/bin/bash
# Dense tail under rocprofv3 on the GPU box: kernel trace of tools/bench_dense.py (4096 x 32768 RREF mod 42013), then
# SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE in their own pass.  Writes gpurun_out/prof_dense_<tag>/dense_tail.json
# (copy it to profiles/<tag>_dense_tail.json: bench.py quotes mfma_busy_pct from there) and a text summary.
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_dense_$TAG
cd "$(dirname "$0")/.." || exit 1
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_dense.py --mfma-only > $OUT/bench.json 2> $OUT/trace.log
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python3 tools/bench_dense.py --mfma-only > /dev/null 2> $OUT/pmc.log
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
bench = {}
for line in open(os.path.join(out, "bench.json")):
    if line.startswith("{"):
        bench = json.loads(line)
ktime = defaultdict(list)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ktime[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sh::", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
ctr = defaultdict(lambda: defaultdict(float))
for f in glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ctr[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sh::", "")][r["Counter_Name"]] += float(r["Counter_Value"])
lines = ["%-40s %6s %12s %10s" % ("kernel", "calls", "total_us", "avg_us")]
for k, v in sorted(ktime.items(), key=lambda kv: -sum(kv[1]))[:16]:
    lines.append("%-40s %6d %12.1f %10.1f" % (k[:40], len(v), sum(v), sum(v) / len(v)))
upd = ctr.get("rref_update_mfma_multi", {})
busy, active = upd.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), upd.get("GRBM_GUI_ACTIVE", 0.0)
# GRBM_GUI_ACTIVE is summed over the 8 XCDs; the chip has 256 CUs x 4 SIMDs issuing MFMA
pct = 100.0 * busy / (active / 8.0 * 1024.0) if active > 0 else None
lines.append("")
lines.append("rref_update_mfma_multi: SQ_VALU_MFMA_BUSY_CYCLES %.4g, GRBM_GUI_ACTIVE %.4g (8 XCDs) -> %.1f %% of the MFMA issue slots while it runs" % (busy, active, pct or 0.0))
open(os.path.join(out, "summary.txt"), "w").write("\n".join(lines) + "\n")
json.dump({"source": "tools/profile_dense.sh: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE of tools/bench_dense.py --mfma-only",
           "shape": bench.get("shape"), "ms_total": bench.get("ms_total_untimed"), "update_Tmacs_per_s": (bench.get("mfma_i8") or {}).get("update_Tmacs_per_s"),
           "mfma_busy_pct": pct, "kernel": "rref_update_mfma_multi"}, open(os.path.join(out, "dense_tail.json"), "w"), indent=1)
print("\n".join(lines))
PY
