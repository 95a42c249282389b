#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then the two HBM counters in separate passes (gpurun
# refuses pmc together with other traces), once per elimination path (default: back-substituted image; then
# SPASM_HIP_BACKSOLVE=0: the row-by-row kernels).  Every launch of the elimination kernels in these runs is the
# bench batch (--no-extras).  Summaries land in gpurun_out/prof_<tag>/ ; copy what you want judged into profiles/:
#   summary_*.txt -> profiles/<tag>_*_summary.txt,  traffic.json -> profiles/<tag>_traffic.json (bench.py quotes it).
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
cd "$(dirname "$0")/.." || exit 1
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="${BENCH_ARGS:-bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras}"
for MODE in default rows; do
	if [ $MODE = rows ]; then export SPASM_HIP_BACKSOLVE=0; else unset SPASM_HIP_BACKSOLVE; fi
	D=$OUT/$MODE
	mkdir -p $D
	rocprofv3 --kernel-trace --stats --output-format csv -d $D/trace -- python3 $ARGS > $D/bench_trace.json 2> $D/trace.log
	rocprofv3 --pmc FETCH_SIZE --output-format csv -d $D/pmc_fetch -- python3 $ARGS > /dev/null 2> $D/pmc_fetch.log
	rocprofv3 --pmc WRITE_SIZE --output-format csv -d $D/pmc_write -- python3 $ARGS > /dev/null 2> $D/pmc_write.log
	# no-return atomics that reach the memory side (round 1's row-group kernel is bound by them; the back-substituted path has none)
	rocprofv3 --pmc TCC_ATOMIC_sum --output-format csv -d $D/pmc_atomic -- python3 $ARGS > /dev/null 2> $D/pmc_atomic.log
	python3 tools/summarize_profile.py $D > $OUT/summary_$MODE.txt 2>&1
	cat $OUT/summary_$MODE.txt
done
python3 tools/summarize_profile.py --merge $OUT/default/traffic.json $OUT/rows/traffic.json > $OUT/traffic.json
cat $OUT/traffic.json
