export SPASM_HIP_EXPERIMENT=1
SPASM_HIP_VERBOSE=3 timeout 300 python tools/probe_image.py mk13.b5 > gpurun_out/r04f_image.log 2>&1
grep -E "factor image|image " gpurun_out/r04f_image.log | tail -45
