#!/bin/bash
# The sparse image on the GL7d19-class stand-in (mk14.b4 by default) under rocprofv3: kernel trace, then FETCH_SIZE and
# WRITE_SIZE in passes of their own (gpurun refuses pmc together with other traces).  Every step of tools/probe_sparse_image.py
# forgets the image: a traced run holds steps + 1 launches of sp_build_kernel, sp_apply_kernel and sp_gather_kernel, each on
# the full round-0 batch.  Writes gpurun_out/prof_spimage_<tag>/{summary.txt,traffic.json}; copy them to
# profiles/<tag>_sparse_image_*.
set -u
TAG=${1:-r04}
NAME=${2:-mk14.b4}
POOL=${3:-1.2e9}
OUT=gpurun_out/prof_spimage_$TAG
cd "$(dirname "$0")/.." || exit 1
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="tools/probe_sparse_image.py --workload $NAME --steps 2 --paths sparse --no-check --pool $POOL ${EXTRA:-}"
python3 $ARGS > $OUT/warm.log 2>&1          # (the round-0 pivots are cached in /tmp by the first run: the traced runs reduce the same rows)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/run_trace.log 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > /dev/null 2> $OUT/pmc_write.log
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
python3 - "$OUT" "$NAME" <<'PY'
import json, os, re, sys
out, name = sys.argv[1], sys.argv[2]
t = json.load(open(os.path.join(out, "traffic.json")))
log = open(os.path.join(out, "run_trace.log")).read()
m = re.search(r"(\d+) rows to reduce", log)
t["workload"] = name
t["rows"] = int(m.group(1)) if m else None
t["source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `tools/probe_sparse_image.py --workload %s --steps 2 --paths sparse`, tools/profile_sparse_image.sh" % name
t["run_under_trace"] = [l.strip() for l in log.splitlines() if l.startswith("sparse") or "fill of R" in l][-2:]
json.dump(t, open(os.path.join(out, "traffic.json"), "w"), indent=1)
PY
head -30 $OUT/summary.txt
cat $OUT/traffic.json
