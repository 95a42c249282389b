mkdir -p gpurun_out
export SPASM_HIP_EXPERIMENT=1
timeout 1500 python -m pytest tests/test_gpu_dense.py -x -q --timeout 600 > gpurun_out/r04h_dense_tests.log 2>&1
tail -3 gpurun_out/r04h_dense_tests.log
timeout 300 python tools/probe_dense_real.py mk13.b5 > gpurun_out/r04h_dense_real.log 2>&1
grep -E "^\{" gpurun_out/r04h_dense_real.log | cut -c1-330
SPASM_HIP_RREF_RETIRE=0 timeout 300 python tools/probe_dense_real_only.py mk13.b5 2>&1 | grep -E "^\{" | cut -c1-330
