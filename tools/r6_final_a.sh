#!/bin/bash
# round 6, after the bounded hand-offs: dense tests, the dense-tail profile (counters serialise the launches: the probe must switch
# the lookahead off and the pass must finish), the pivot-search profile, stage cycles of the sparse image's kernels
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
echo "=== dense tests"; timeout 900 python -m pytest tests/test_gpu_dense.py -x -q 2>&1 | tail -5
echo "=== dense tail"; bash tools/profile_dense.sh r06 2>&1 | tail -25
grep -h "side by side" gpurun_out/prof_dense_r06/*.log | head -3
echo "=== pivot search"; bash tools/profile_pivots.sh r06 2>&1 | tail -25
echo "=== stage cycles, device-search pivots"; SPASM_HIP_EXPERIMENT=1 SPASM_HIP_SPARSE_IMAGE_PROFILE=1 SPASM_HIP_VERBOSE=1 timeout 600 python3 tools/probe_sparse_image.py --workload mk15.b4 --steps 2 --paths sparse --no-check --pool 1.5e9 2>&1 | grep -v "^\[factor\|^\[sparse image\] R:" | tail -30
echo "=== stage cycles, fixed set"; SPASM_HIP_EXPERIMENT=1 SPASM_HIP_SPARSE_IMAGE_PROFILE=1 SPASM_HIP_VERBOSE=1 timeout 600 python3 tools/probe_sparse_image.py --workload mk15.b4 --fixed-pivots --steps 2 --paths sparse --no-check --pool 3.0e9 2>&1 | grep -v "^\[factor\|^\[sparse image\] R:" | tail -30
} > gpurun_out/r6_final_a.log 2>&1
tail -120 gpurun_out/r6_final_a.log
