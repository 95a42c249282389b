mkdir -p gpurun_out
export SPASM_HIP_EXPERIMENT=1
timeout 400 python tools/probe_pivot_waves.py mk15.b4 8 > gpurun_out/r04b_pivots_mk15b4.log 2>&1
grep -E "device:|==" gpurun_out/r04b_pivots_mk15b4.log | cut -c1-900
timeout 400 python tools/probe_pivot_waves.py mk14.b4 4 > gpurun_out/r04b_pivots_mk14b4.log 2>&1
grep -E "device:|==" gpurun_out/r04b_pivots_mk14b4.log | cut -c1-900
timeout 400 python tools/probe_pivot_waves.py ch8-8.b5 4 > gpurun_out/r04b_pivots_ch8.log 2>&1
grep -E "device:|==" gpurun_out/r04b_pivots_ch8.log | cut -c1-900
