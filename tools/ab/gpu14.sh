export SPASM_HIP_EXPERIMENT=1
SPASM_HIP_RREF_TIMING=1 timeout 300 python tools/probe_dense_real_only.py mk13.b5 2>&1 | grep -E "^\{|end of the panels|super-panel|regular way" | cut -c1-330 | tail -16
