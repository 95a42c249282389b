#!/bin/bash
# The sparse path on the GL7d19-class stand-in (mk14.b4) under rocprofv3: kernel trace, then FETCH_SIZE / WRITE_SIZE /
# TCC_ATOMIC_sum in separate passes (gpurun refuses pmc together with other traces).  Every launch of the row-group kernel
# in these runs is the full round-0 batch (tools/probe_sparse.py, 24 GB of accumulator slices as in a one-shot call).
# Writes gpurun_out/prof_sparse_<tag>/{summary.txt,traffic.json}; copy them to profiles/<tag>_sparse_*.
set -u
TAG=${1:-r03}
OUT=gpurun_out/prof_sparse_$TAG
cd "$(dirname "$0")/.." || exit 1
mkdir -p $OUT
export TMPDIR=/tmp
export SPASM_HIP_SCRATCH_GB=24
ARGS="tools/probe_sparse.py mk14.b4 --steps 2"
python3 $ARGS > $OUT/warm.log 2>&1          # (the round-0 pivots are cached in /tmp by the first run: the traced runs reduce the same rows)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/run_trace.log 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > /dev/null 2> $OUT/pmc_write.log
rocprofv3 --pmc TCC_ATOMIC_sum --output-format csv -d $OUT/pmc_atomic -- python3 $ARGS > /dev/null 2> $OUT/pmc_atomic.log
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
python3 - "$OUT" <<'PY'
import json, os, re, sys
out = sys.argv[1]
t = json.load(open(os.path.join(out, "traffic.json")))
log = open(os.path.join(out, "run_trace.log")).read()
m = re.search(r"(\d+) rows to reduce", log)
t["workload"] = "mk14.b4"
t["rows"] = int(m.group(1)) if m else None
t["source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_ATOMIC_sum (separate passes) of `tools/probe_sparse.py mk14.b4 --steps 2`, tools/profile_sparse.sh"
t["run_under_trace"] = [l.strip() for l in log.splitlines() if "ms total" in l][-1:]
json.dump(t, open(os.path.join(out, "traffic.json"), "w"), indent=1)
PY
cat $OUT/summary.txt | head -40
cat $OUT/traffic.json
