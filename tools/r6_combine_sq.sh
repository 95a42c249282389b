#!/bin/bash
# SQ / TCC counters of the combination kernels of a driver call on mk15.b4 (two rocprofv3 --pmc passes, nothing else traced)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_combine
rm -rf $OUT; mkdir -p $OUT
SPASM_HIP_VERBOSE=0 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/a -- python3 tools/probe_e2e.py mk15.b4 1 > $OUT/a.log 2>&1
SPASM_HIP_VERBOSE=0 timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCC_ATOMIC_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 tools/probe_e2e.py mk15.b4 1 > $OUT/b.log 2>&1
python3 - <<'PY'
import csv,glob,re
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(float)); n=defaultdict(set)
for d in ('a','b'):
    for f in glob.glob('gpurun_out/prof_combine/%s/**/*counter_collection.csv'%d, recursive=True):
        for r in csv.DictReader(open(f)):
            k=re.split(r'[<(]',r['Kernel_Name'].replace('void ','').replace('sh::','').replace('(anonymous namespace)::',''))[0]
            if 'combine' in k or 'fl_census' in k or 'rows_nonpivotal' in k:
                acc[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k].add(r['Dispatch_Id'])
for k,v in acc.items():
    print(k, 'dispatches', len(n[k])//2 or len(n[k]))
    for c,x in sorted(v.items()): print('   %-22s %.4g'%(c,x))
PY
