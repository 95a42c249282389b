mkdir -p gpurun_out
export SPASM_HIP_EXPERIMENT=1
timeout 900 python -m pytest tests/test_gpu_pivots.py -x -q --timeout 600 > gpurun_out/r04b_pivot_tests.log 2>&1
tail -3 gpurun_out/r04b_pivot_tests.log
for set in 4096 8192 2048 0; do
export SPASM_HIP_PIVOT_SET=$set
echo "### set=$set"
timeout 400 python tools/probe_pivot_waves.py mk15.b4 8 4 > gpurun_out/r04d_pivots_mk15b4_$set.log 2>&1
grep -E "device:|==" gpurun_out/r04d_pivots_mk15b4_$set.log | cut -c1-560
done
export SPASM_HIP_PIVOT_SET=4096
export SPASM_HIP_PIVOT_BITS=global
for w in mk14.b4 ch8-8.b5 mk13.b5; do
timeout 400 python tools/probe_pivot_waves.py $w 8 > gpurun_out/r04d_pivots_${w}_gb.log 2>&1
grep -E "device:|==" gpurun_out/r04d_pivots_${w}_gb.log | cut -c1-560
done
timeout 400 python tools/probe_long_rows.py 300000 310000 19 device > gpurun_out/r04d_pivots_long_gb.log 2>&1
grep -E "device:|==" gpurun_out/r04d_pivots_long_gb.log | cut -c1-560
