mkdir -p gpurun_out
export SPASM_HIP_EXPERIMENT=1
timeout 900 python -m pytest tests/test_gpu_pivots.py -x -q --timeout 600 > gpurun_out/r04g_pivot_tests.log 2>&1
tail -3 gpurun_out/r04g_pivot_tests.log
timeout 400 python tools/probe_long_rows.py 300000 310000 19 device > gpurun_out/r04g_pivots_long.log 2>&1
grep -E "device:|==" gpurun_out/r04g_pivots_long.log | cut -c1-600
SPASM_HIP_PIVOT_BITS=global timeout 400 python tools/probe_long_rows.py 300000 310000 19 device > gpurun_out/r04g_pivots_long_gb.log 2>&1
grep -E "device:|==" gpurun_out/r04g_pivots_long_gb.log | cut -c1-600
timeout 400 python tools/probe_pivot_waves.py mk15.b4 8 > gpurun_out/r04g_pivots_mk15b4.log 2>&1
grep -E "device:|==" gpurun_out/r04g_pivots_mk15b4.log | cut -c1-600
