#!/bin/bash
# kernel totals of spasm_hip_echelonize on a stand-in: tools/r6_prof_e2e_any.sh <name> [dense-threshold]
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_any
rm -rf $OUT; mkdir -p $OUT
SPASM_HIP_VERBOSE=2 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/probe_e2e.py $1 2 $2 > $OUT/run.log 2> $OUT/trace.log
python3 - <<'PY'
import csv,glob,os,re
from collections import defaultdict
f=sorted(glob.glob('gpurun_out/prof_any/trace/**/*kernel_trace.csv', recursive=True), key=os.path.getmtime)[-1]
kt=defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].replace('void ','').replace('sh::','').replace('(anonymous namespace)::','')
    n=re.split(r'[<(]',n)[0]
    kt[n].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(kt.items(), key=lambda kv:-sum(kv[1]))[:16]:
    print('%-40s %6d %10.1f %8.1f'%(k[:40],len(v),sum(v),sum(v)/len(v)))
print('all', sum(sum(v) for v in kt.values())/1e3,'ms', sum(len(v) for v in kt.values()))
PY
grep -E "echelon rows\]|== |echelonize/dense" $OUT/run.log | cut -c1-260 | tail -24
