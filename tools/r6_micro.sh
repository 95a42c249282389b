#!/bin/bash
# round 6: A/B of several builds on the same box -- sparse image at scale (stage times)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
TAG=${1:-x}
shift
{
for rep in 1 2; do
for lib in "$@"; do
  echo "== ${lib}"
  SPASM_HIP_LIB=$lib timeout 600 python tools/probe_sparse_image.py --workload mk15.b4 --steps 5 --paths sparse --no-check --fixed-pivots --pool 3.0e9 2>&1 | grep "sparse total" | sed 's/([^)]*)//g'
  SPASM_HIP_LIB=$lib timeout 600 python tools/probe_sparse_image.py --workload mk15.b4 --steps 5 --paths sparse --no-check --pool 1.5e9 2>&1 | grep "sparse total" | sed 's/([^)]*)//g'
done
done
} > gpurun_out/r6_micro_$TAG.log 2>&1
cat gpurun_out/r6_micro_$TAG.log
