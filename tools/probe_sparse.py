#!/usr/bin/env python3
"""GPU box: the round-0 Schur complement of a workload on the row-by-row path, with every counter the library keeps:
which kernels ran (row-group kernel / per-row tiers), whether the row-group kernel gave up, lane efficiency, times."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")          # (this script names kernel variants and debugging aids)
os.environ.setdefault("SPASM_HIP_VERBOSE", "0")
ap = argparse.ArgumentParser()
ap.add_argument("workloads", nargs="*", default=["mk14.b4"])
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--rows", type=int, default=0, help="only the first ROWS rows of the batch")
args = ap.parse_args()

import torch
import spasm_amd
import workloads

for name in args.workloads:
    A, rows, F, source = workloads.round0(name, 42013, threads=0 if name.startswith("mk14") else 1)
    if args.rows:
        rows = rows[:args.rows]
    print("%s: %d x %d, %d nnz; %d pivots, %d rows to reduce, %d non-pivotal columns" % (name, A.n, A.m, A.nnz, F.U.n, len(rows), A.m - F.U.n), flush=True)
    dev = torch.device("cuda:0")
    dA = spasm_amd.DeviceCsr.from_host(A, dev)
    drows = torch.from_numpy(np.ascontiguousarray(rows)).to(dev)
    os.environ["SPASM_HIP_BACKSOLVE"] = "0"
    dF = spasm_amd.DeviceFact(F)
    print("  factor: %d levels" % dF.levels, flush=True)
    pool = 1 << 28
    while True:
        W = spasm_amd.SchurWorkspace(len(rows), A.m, pool)
        S, st = spasm_amd.dschur(dA, drows, dF, W, fetch=False)
        if st.status == 0:
            break
        W.close()
        pool *= 2
    for _ in range(args.steps):
        _, st = spasm_amd.dschur(dA, drows, dF, W, fetch=False)
        eff = st.eliminations / (64.0 * st.group_pivots) if st.group_pivots else 0.0
        print("  %s: %.1f ms total (group %.1f, tier0 %.1f, tier1 %.1f, tier2 %.1f, finalize %.1f); group kernel %d aborted %d; rows lds %d / big %d / dense-or-group %d; "
              "eliminations %.3e, streamed %.3e, group pivots %.3e, lane efficiency %.3f; nnz(S) %d density %.4f; %.2f M rows/s"
              % (st.kernel.decode(), st.ms_total, st.ms_group, st.ms_tier0, st.ms_tier1, st.ms_tier2, st.ms_finalize, st.used_group_kernel, st.group_aborted,
                 st.rows_lds, st.rows_lds_big, st.rows_dense, st.eliminations, st.entries_streamed, st.group_pivots, eff, st.nnz,
                 st.nnz / (len(rows) * max(A.m - F.U.n, 1)), len(rows) / st.ms_total / 1e3), flush=True)
    W.close()
    dF.close()
