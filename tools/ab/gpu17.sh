export SPASM_HIP_EXPERIMENT=1
SPASM_HIP_RREF_TIMING=1 timeout 900 python -m pytest tests/test_gpu_dense.py -x -q -k "tries_fail" --timeout 600 -s 2>&1 | grep -E "passed|failed|end of the panels|Error|assert" | tail -12 | cut -c1-250
SPASM_HIP_RREF_TIMING=1 timeout 300 python tools/probe_dense_real_only.py mk13.b5 2>&1 | grep -E "^\{|end of the panels" | cut -c1-300 | tail -3
timeout 1500 python -m pytest tests/test_gpu_dense.py tests/test_gpu_echelonize.py -x -q --timeout 600 > gpurun_out/r04j_tests.log 2>&1
tail -3 gpurun_out/r04j_tests.log
