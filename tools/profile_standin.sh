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
export SPASM_HIP_RREF_TIMING=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/probe_standins.py $NAME > $OUT/out.txt 2> $OUT/err.txt
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
head -25 "$f" | cut -d, -f1-5 | cut -c1-200 > $OUT/kernel_stats_head.txt
cat $OUT/kernel_stats_head.txt
grep -v "^\[rref/hip\]" $OUT/out.txt | tail -30
