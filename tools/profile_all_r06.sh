#!/bin/bash
# round 6: every profile the documents quote, in one go (GPU box).  Results under gpurun_out/; copy the summaries to profiles/.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
echo "=== bench paths (trace + HBM counters)"; bash tools/profile.sh r06 2>&1 | tail -40
echo "=== sparse image at scale, fixed pivot set (trace + HBM counters)"; EXTRA=--fixed-pivots bash tools/profile_sparse_image.sh r06 mk15.b4 3.0e9 2>&1 | tail -40
echo "=== sparse image at scale, fixed pivot set (SQ counters)"; EXTRA=--fixed-pivots bash tools/profile_sparse_image_sq.sh r06 mk15.b4 3.0e9 > gpurun_out/prof_spimage_sq_r06.out 2>&1; tail -5 gpurun_out/prof_spimage_sq_r06.out
echo "=== sparse image at scale, pivots of the device search (SQ counters: instructions per (row, segment) pair against round 5)"; bash tools/profile_sparse_image_sq.sh r06dev mk15.b4 1.5e9 > gpurun_out/prof_spimage_sq_r06dev.out 2>&1; tail -3 gpurun_out/prof_spimage_sq_r06dev.out
echo "=== dense tail"; bash tools/profile_dense.sh r06 2>&1 | tail -25
echo "=== pivot search"; bash tools/profile_pivots.sh r06 2>&1 | tail -25
} > gpurun_out/profile_all_r06.log 2>&1
tail -150 gpurun_out/profile_all_r06.log
