mkdir -p gpurun_out
export SPASM_HIP_EXPERIMENT=1
timeout 900 python -m pytest tests/test_gpu_pivots.py -x -q --timeout 600 > gpurun_out/r04b_pivot_tests.log 2>&1
tail -3 gpurun_out/r04b_pivot_tests.log
for bw in 1 0; do
export SPASM_HIP_PIVOT_BACKWARD=$bw
echo "#### backward=$bw"
timeout 400 python tools/probe_pivot_waves.py mk15.b4 8 > gpurun_out/r04b_pivots_mk15b4_bw$bw.log 2>&1
grep -E "device:|==" gpurun_out/r04b_pivots_mk15b4_bw$bw.log | cut -c1-700
timeout 400 python tools/probe_pivot_waves.py mk14.b4 4 > gpurun_out/r04b_pivots_mk14b4_bw$bw.log 2>&1
grep -E "device:|==" gpurun_out/r04b_pivots_mk14b4_bw$bw.log | cut -c1-700
timeout 400 python tools/probe_pivot_waves.py ch8-8.b5 4 > gpurun_out/r04b_pivots_ch8_bw$bw.log 2>&1
grep -E "device:|==" gpurun_out/r04b_pivots_ch8_bw$bw.log | cut -c1-700
timeout 400 python tools/probe_pivot_waves.py mk13.b5 4 > gpurun_out/r04b_pivots_mk13b5_bw$bw.log 2>&1
grep -E "device:|==" gpurun_out/r04b_pivots_mk13b5_bw$bw.log | cut -c1-700
timeout 400 python tools/probe_long_rows.py 300000 310000 19 device > gpurun_out/r04b_pivots_long_bw$bw.log 2>&1
grep -E "device:|==" gpurun_out/r04b_pivots_long_bw$bw.log | cut -c1-700
done
