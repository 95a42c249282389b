#!/bin/bash
# round 6 iteration loop for the sparse image: parity first (smoke + the sparse-image tests), then the kernels' times on the fixed pivot
# sets, this build and round 5's (tools/ab/libspasm_hip_r05.so) on the same box
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
TAG=${1:-x}
{
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_gpu_sparse_image.py -x -q 2>&1 | tail -3
for w in mk14.b4 mk15.b4; do
  for lib in "" tools/ab/libspasm_hip_r05.so; do
    [ -n "$lib" ] && [ ! -f "$lib" ] && continue
    echo "== $w ${lib:-this build} (fixed pivot set)"
    SPASM_HIP_LIB=$lib timeout 600 python tools/probe_sparse_image.py --workload $w --steps 3 --paths sparse --no-check --fixed-pivots --pool 3.0e9 2>&1 | grep "sparse total"
  done
done
for lib in "" tools/ab/libspasm_hip_r05.so; do
  [ -n "$lib" ] && [ ! -f "$lib" ] && continue
  echo "== mk15.b4 ${lib:-this build} (device pivot search)"
  SPASM_HIP_LIB=$lib timeout 600 python tools/probe_sparse_image.py --workload mk15.b4 --steps 3 --paths sparse --no-check --pool 1.5e9 2>&1 | grep "sparse total"
done
} > gpurun_out/r6_sp_$TAG.log 2>&1
cat gpurun_out/r6_sp_$TAG.log
