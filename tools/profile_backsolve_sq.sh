#!/bin/bash
# What the build of the dense image waits for: wave-level SQ counters of backsolve_kernel on the bench workload (one rocprofv3 --pmc
# pass, no other trace with it).  Writes gpurun_out/prof_backsolve_sq_<tag>/sq.json; copy it to profiles/<tag>_backsolve_sq.json.
set -u
TAG=${1:-r06}
OUT=gpurun_out/prof_backsolve_sq_$TAG
cd "$(dirname "$0")/.." || exit 1
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
CTRS="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
python3 $ARGS > /dev/null 2>&1
rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/run.log 2> $OUT/pmc_sq.log
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 $ARGS > /dev/null 2> $OUT/pmc_sq2.log
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for d in ("pmc_sq", "pmc_sq2"):
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("sh::", "").split("(")[0]
            if k.startswith("backsolve_kernel") or k.startswith("bs_apply") or k.startswith("bs_expand"):
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"source": "rocprofv3 --pmc (two passes) -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras (tools/profile_backsolve_sq.sh)", "kernels": {}}
for k, d in acc.items():
    per = {c: sum(v) / len(v) for c, v in d.items()}
    res["kernels"][k] = {"launches": max(len(v) for v in d.values()), "per_launch": per}
json.dump(res, open(os.path.join(out, "sq.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
