export SPASM_HIP_EXPERIMENT=1
timeout 600 bash tools/profile_dense.sh r04big > gpurun_out/r04big_profile.log 2>&1
head -24 gpurun_out/prof_dense_r04big/summary.txt 2>/dev/null || tail -20 gpurun_out/r04big_profile.log
