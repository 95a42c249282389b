#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
TAG=${1:-x}
{
for w in mk15.b4; do
  SPASM_HIP_SPARSE_IMAGE_PROFILE=1 timeout 600 python tools/probe_sparse_image.py --workload $w --steps 2 --paths sparse --no-check --fixed-pivots --pool 3.0e9 2>&1 | grep -v "^\[factor\|^\[sparse image\] R:" | tail -12
done
} > gpurun_out/r6_prof_$TAG.log 2>&1
tail -40 gpurun_out/r6_prof_$TAG.log
