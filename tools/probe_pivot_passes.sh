#!/bin/bash
# GPU box: the passes of the labelled pivot search (pivots_device.hip) on the generated stand-ins -- second labelled pass always
# / by the library's rule / with another cap: time of the greedy search and the counters of every pass.
# bash tools/probe_pivot_passes.sh [name ...]
cd "$(dirname "$0")/.." || exit 1
NAMES=${@:-mk13.b5 ch8-8.b5 mk14.b4 mk15.b4}
for cfg in "SPASM_HIP_PIVOT_SECOND_PASS_ALWAYS=1 SPASM_HIP_PIVOT_SECOND_PASS_MANY_ROWS=0" "X=1" "SPASM_HIP_PIVOT_LABEL_PASSES=1"; do
for n in $NAMES; do
echo "== $cfg $n"
env $cfg SPASM_HIP_EXPERIMENT=1 SPASM_HIP_VERBOSE=3 SPASM_HIP_PIVOT_STATS=1 timeout 600 python tools/probe_e2e.py $n 3 0.01 2>&1 | grep -E "greedy search [0-9]|pass [12] on|ticket search on|labels of the final" | sed -E 's/: [0-9]+ searches in flight.*cascades:/: cascades:/; s/\(.* thrown away.*\), ([0-9]+ rows deferred)/ \1/' | cut -c1-230 | tail -5
done; done
