mkdir -p gpurun_out
timeout 1500 bash tools/profile.sh r04 > gpurun_out/r04_profile_default.log 2>&1
tail -5 gpurun_out/r04_profile_default.log
timeout 900 bash tools/profile_dense.sh r04 > gpurun_out/r04_profile_dense.log 2>&1
tail -4 gpurun_out/r04_profile_dense.log
timeout 900 bash tools/profile_sparse_image.sh r04 mk14.b4 1.2e9 > gpurun_out/r04_profile_spimage.log 2>&1
tail -4 gpurun_out/r04_profile_spimage.log
