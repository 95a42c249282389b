export SPASM_HIP_EXPERIMENT=1
SPASM_HIP_RREF_TIMING=1 timeout 300 python tools/probe_dense_real_only.py mk13.b5 2>&1 | grep -E "^\{|end of the panels|super-panel|regular way" | cut -c1-300 | tail -12
timeout 300 python tools/probe_dense_deficient.py 2>&1 | grep -E "rank|end of the panels" | cut -c1-260 | tail -4
timeout 1500 python -m pytest tests/test_gpu_dense.py -x -q --timeout 600 > gpurun_out/r04i_dense_tests.log 2>&1
tail -3 gpurun_out/r04i_dense_tests.log
timeout 300 python tools/probe_dense_real.py mk13.b5 2>&1 | grep -E "^\{" | cut -c1-300
