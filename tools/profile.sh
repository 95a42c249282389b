#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then the two HBM
# counters in separate passes (gpurun refuses pmc together with other traces).
# Summaries land in gpurun_out/prof_<tag>/ ; copy what you want judged into profiles/.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
ARGS="${BENCH_ARGS:-bench.py --steps 3 --warmup 1 --no-cpu-baseline}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > /dev/null 2> $OUT/pmc_write.log
find $OUT -name "*.csv" | head -20
python3 tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
