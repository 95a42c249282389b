#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
TAG=${1:-x}
{
for w in mk14.b4 mk15.b4; do
for v in "" "SPASM_HIP_SP_HEADS=0" "SPASM_HIP_SP_TICKET_AHEAD=0" "SPASM_HIP_SP_HEADS=0 SPASM_HIP_SP_TICKET_AHEAD=0" "SPASM_HIP_LIB=tools/ab/libspasm_hip_r05.so"; do
  echo "== $w $v"
  env $v timeout 600 python tools/probe_sparse_image.py --workload $w --steps 4 --paths sparse --no-check --fixed-pivots --pool 3.0e9 2>&1 | grep "sparse total" | sed 's/(.*rows of S/ rows of S/; s/(.*scan/ scan/; s/(.*//'
done
done
} > gpurun_out/r6_build_$TAG.log 2>&1
cat gpurun_out/r6_build_$TAG.log
