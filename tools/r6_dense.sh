#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
TAG=${1:-x}
{
timeout 1500 python -m pytest tests/test_gpu_dense.py -x -q 2>&1 | tail -3
for v in "SPASM_HIP_RREF_LOOKAHEAD=1" "SPASM_HIP_RREF_LOOKAHEAD=0"; do
  echo "== $v"
  env SPASM_HIP_EXPERIMENT=1 $v timeout 300 python -c "
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import torch, spasm_amd, bench
dev=torch.device('cuda:0')
d=bench.dense_tail_probe(torch, spasm_amd, dev)
print({k:d[k] for k in ('shape','rank','ms','update_kernels_ms_serialised')})
d=bench.dense_tail_probe(torch, spasm_amd, dev, n=8192, m=16384)
print({k:d[k] for k in ('shape','rank','ms','update_kernels_ms_serialised')})
" 2>&1 | tail -2
  env SPASM_HIP_EXPERIMENT=1 $v timeout 300 python tools/probe_dense_real_only.py 2>&1 | tail -1
done
} > gpurun_out/r6_dense_$TAG.log 2>&1
cat gpurun_out/r6_dense_$TAG.log
