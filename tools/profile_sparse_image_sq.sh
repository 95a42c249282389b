#!/bin/bash
# What the kernels of the sparse image wait for: wave-level SQ counters of sp_build_kernel / sp_apply_kernel / sp_gather_kernel
# on a stand-in (mk15.b4 by default) -- instructions by class, cycles with an instruction in flight, cycles parked on a
# wait count -- in ONE rocprofv3 --pmc pass (no other trace with it: gpurun refuses the combination).  Per launch and per
# (row, segment) pair of the apply kernel.  Writes gpurun_out/prof_spimage_sq_<tag>/sq.json; copy it to
# profiles/<tag>_at_scale_sparse_image_sq.json.
set -u
TAG=${1:-r05}
NAME=${2:-mk15.b4}
POOL=${3:-1.2e9}
OUT=gpurun_out/prof_spimage_sq_$TAG
cd "$(dirname "$0")/.." || exit 1
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="tools/probe_sparse_image.py --workload $NAME --steps 2 --paths sparse --no-check --pool $POOL ${EXTRA:-}"
CTRS="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"
python3 $ARGS > $OUT/warm.log 2>&1
rocprofv3 --pmc $CTRS --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/run.log 2> $OUT/pmc_sq.log
python3 - "$OUT" "$NAME" "$CTRS" <<'PY'
import csv, glob, json, os, re, sys
from collections import defaultdict
out, name, ctrs = sys.argv[1], sys.argv[2], sys.argv[3].split()
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_sq", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("sh::", "").split("(")[0]
        if k.startswith("sp_"):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
log = open(os.path.join(out, "run.log")).read()
m = re.search(r"(\d+) rows to reduce", log)
res = {"source": "rocprofv3 --pmc %s -- python3 tools/probe_sparse_image.py --workload %s --steps 2 --paths sparse (tools/profile_sparse_image_sq.sh)" % (" ".join(ctrs), name),
       "workload": name, "rows": int(m.group(1)) if m else None, "kernels": {}}
for k, d in acc.items():
    launches = max(len(v) for v in d.values())
    per = {c: sum(v) / len(v) for c, v in d.items()}          # per launch
    o = {"launches": launches, "per_launch": per}
    wc = per.get("SQ_WAVE_CYCLES", 0.0)
    if wc > 0:
        o["fraction_parked_on_waitcnt"] = per.get("SQ_WAIT_ANY", 0.0) / wc
        o["fraction_with_an_instruction_in_flight"] = per.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
        o["fraction_waiting_to_issue"] = per.get("SQ_WAIT_INST_ANY", 0.0) / wc
    res["kernels"][k] = o
json.dump(res, open(os.path.join(out, "sq.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
