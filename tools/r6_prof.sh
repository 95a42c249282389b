#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
TAG=${1:-x}
{
SPASM_HIP_SPARSE_IMAGE_PROFILE=1 timeout 600 python tools/probe_sparse_image.py --workload mk15.b4 --steps 1 --paths sparse --no-check --fixed-pivots --pool 3.0e9 2>&1 | grep "profile\|sparse total" | tail -3
bash tools/profile_sparse_image_sq.sh r06$TAG mk15.b4 1.5e9 2>&1 | python3 -c "
import json,sys
t=sys.stdin.read()
d=json.loads(t[t.index('{'):])
for k,v in d['kernels'].items():
    if k.startswith('sp_apply') or k.startswith('sp_build_kernel') or k.startswith('sp_gather'):
        p=v['per_launch']; print(k, {c:'%.3g'%x for c,x in p.items()}, 'parked %.2f'%v.get('fraction_parked_on_waitcnt',0))
print('rows',d['rows'])
"
} > gpurun_out/r6_prof_$TAG.log 2>&1
tail -40 gpurun_out/r6_prof_$TAG.log
