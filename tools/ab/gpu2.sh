mkdir -p gpurun_out
export SPASM_HIP_EXPERIMENT=1
timeout 600 python -m pytest tests/test_gpu_dense.py -x -q -k "echelon_extend" --timeout 300 > gpurun_out/r04b_extend_tests.log 2>&1
tail -3 gpurun_out/r04b_extend_tests.log
SPASM_HIP_VERBOSE=2 timeout 300 python tools/probe_dense_real.py mk13.b5 > gpurun_out/r04b_dense_real2.log 2>&1
tail -12 gpurun_out/r04b_dense_real2.log
