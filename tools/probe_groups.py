"""Row-group kernel: run time against the number of row groups (mk13.b5, every k-th row), for 1 and 4 waves per
group.  With few groups the run time is the chain of level rounds of one group, not the throughput of the chip.
Run on a GPU box: python tools/probe_groups.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SPASM_HIP_VERBOSE"] = "0"
import numpy as np, torch, bench, spasm_amd

A, rows, F = bench.build_workload("mk13.b5")
dA = spasm_amd.DeviceCsr.from_host(A)
dF = spasm_amd.DeviceFact(F)
for step in (1, 2, 3, 4, 8, 16, 64):
    sub = np.ascontiguousarray(rows[::step])
    d = torch.from_numpy(sub).cuda()
    line = "every %2d. row: %6d rows %4d groups:" % (step, len(sub), (len(sub) + 63) // 64)
    for waves in (1, 2, 4):
        os.environ["SPASM_HIP_GROUP_WAVES"] = str(waves)
        W = spasm_amd.SchurWorkspace(len(sub), A.m, 4 * A.nnz + (1 << 28))
        for _ in range(3):
            S, st = spasm_amd.dschur(dA, d, dF, W, fetch=False)
        line += "  %d wave(s) %6.2f ms" % (waves, st.ms_group if st.used_group_kernel else st.ms_eliminate)
        W.close()
    print(line + "  (lane efficiency %.2f)" % (st.eliminations / (64.0 * max(1, st.group_pivots))), flush=True)
