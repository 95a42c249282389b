#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
TAG=${1:-x}
{
for v in "SPASM_HIP_RREF_LOOKAHEAD=1" "SPASM_HIP_RREF_LOOKAHEAD=1 SPASM_HIP_RREF_CU_MASK=0" "SPASM_HIP_RREF_LOOKAHEAD=0" "SPASM_HIP_RREF_LOOKAHEAD=0 SPASM_HIP_RREF_CU_MASK=0"; do
  echo "== $v"
  env SPASM_HIP_EXPERIMENT=1 $v timeout 300 python -c "
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import torch, spasm_amd, bench
dev=torch.device('cuda:0')
for shape in ((4096,32768),(8192,16384),(4096,8192),(6144,12288)):
    d=bench.dense_tail_probe(torch, spasm_amd, dev, n=shape[0], m=shape[1])
    print({k:d[k] for k in ('shape','rank','ms')})
" 2>&1 | tail -4
  env SPASM_HIP_EXPERIMENT=1 $v timeout 300 python tools/probe_dense_real_only.py 2>&1 | tail -1 | sed 's/.*ms_first/ms_first/'
done
} > gpurun_out/r6_dense2_$TAG.log 2>&1
cat gpurun_out/r6_dense2_$TAG.log
