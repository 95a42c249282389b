import os, sys, time
sys.path.insert(0, "/root/repo"); os.environ["SPASM_HIP_VERBOSE"]="0"
import numpy as np, torch, bench, spasm_amd
A, rows, F = bench.build_workload("mk13.b5")
dA = spasm_amd.DeviceCsr.from_host(A); dF = spasm_amd.DeviceFact(F)
n = len(rows)
cases = [("every 16th", rows[::16]), ("all", rows)]
for name, sub in cases:
    sub = np.ascontiguousarray(sub)
    W = spasm_amd.SchurWorkspace(len(sub), A.m, 4 * A.nnz + (1 << 28))
    d = torch.from_numpy(sub).cuda()
    for _ in range(2):
        S, st = spasm_amd.dschur(dA, d, dF, W, fetch=False)
    print("%-14s rows %6d groups %4d: group kernel %.1f ms, elim %.3e, wave-pivots %.3e, streamed %.3e, eff %.2f" % (name, len(sub), (len(sub)+63)//64, st.ms_group, st.eliminations, st.group_pivots, st.entries_streamed, st.eliminations / (64.0 * max(1, st.group_pivots))), flush=True)
    W.close()
