export SPASM_HIP_EXPERIMENT=1
timeout 300 python tools/probe_dense_deficient.py 2>&1 | grep -E "rank|end of the panels" | cut -c1-260
SPASM_HIP_RREF_RETIRE=0 timeout 300 python tools/probe_dense_deficient.py 2>&1 | grep -E "rank|end of the panels" | cut -c1-260
