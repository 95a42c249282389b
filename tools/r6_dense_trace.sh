#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cat > /tmp/dense_once.py <<'P'
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import torch, spasm_amd, bench
dev=torch.device('cuda:0')
d=bench.dense_tail_probe(torch, spasm_amd, dev)
print({k:d[k] for k in ('shape','rank','ms')})
P
for v in 1 0; do
  rm -rf gpurun_out/dtrace_$v
  SPASM_HIP_EXPERIMENT=1 SPASM_HIP_RREF_LOOKAHEAD=$v rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dtrace_$v -- python3 /tmp/dense_once.py > gpurun_out/dtrace_$v.log 2>&1
  python3 - $v <<'P'
import csv,glob,sys
v=sys.argv[1]
f=glob.glob('gpurun_out/dtrace_%s/**/*kernel_trace.csv'%v, recursive=True)[0]
rows=list(csv.DictReader(open(f)))
ks=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in rows if 'rref_' in r['Kernel_Name']]
ks.sort()
tr=[k for k in ks if 'try_inverse' in k[2]]
# take the third drref call: last 64 tries
tr=tr[-64-64:-64] if len(tr)>=192 else tr[-64:]
per=[(tr[i+1][0]-tr[i][0])/1e3 for i in range(len(tr)-1)]
per.sort()
print('lookahead',v,'tries',len(tr),'start-to-start us: median %.1f p10 %.1f p90 %.1f'%(per[len(per)//2],per[len(per)//10],per[9*len(per)//10]),'try duration us %.1f'%(sum(t[1]-t[0] for t in tr)/len(tr)/1e3))
t0=tr[8][0]
for k in ks:
    if t0-2000 <= k[0] <= t0+260000:
        print('  %8.1f +%6.1f %s'%((k[0]-t0)/1e3,(k[1]-k[0])/1e3,k[2][:40]))
P
done
