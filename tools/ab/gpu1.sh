mkdir -p gpurun_out
export SPASM_HIP_EXPERIMENT=1
timeout 300 python tools/probe_sparse_image.py --workload mk13.b4 > gpurun_out/r04b_probe_mk13b4.log 2>&1
timeout 300 python tools/probe_dense_real.py mk13.b5 > gpurun_out/r04b_dense_real.log 2>&1
timeout 300 python tools/probe_image.py mk13.b5 > gpurun_out/r04b_image.log 2>&1
tail -5 gpurun_out/r04b_probe_mk13b4.log; tail -8 gpurun_out/r04b_dense_real.log; tail -30 gpurun_out/r04b_image.log
