mkdir -p gpurun_out
export SPASM_HIP_EXPERIMENT=1
timeout 1200 python -m pytest tests/test_gpu_dense.py -x -q -k "rref" --timeout 600 > gpurun_out/r04e_rref_tests.log 2>&1
tail -3 gpurun_out/r04e_rref_tests.log
timeout 300 python tools/probe_dense_real.py mk13.b5 > gpurun_out/r04e_dense_real.log 2>&1
grep -E "^\{|rank" gpurun_out/r04e_dense_real.log | cut -c1-600
SPASM_HIP_RREF_BIG_TILES=0 timeout 300 python tools/probe_dense_real.py mk13.b5 > gpurun_out/r04e_dense_real_old.log 2>&1
grep -E "^\{" gpurun_out/r04e_dense_real_old.log | cut -c1-600
