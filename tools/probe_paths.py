#!/usr/bin/env python3
"""GPU box: the round-0 Schur complement of a workload through both elimination paths (back-substituted image / row by row),
same result checked, device times printed; then spasm_hip_echelonize end to end with its time split."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")          # (this script names kernel variants and debugging aids)
os.environ.setdefault("SPASM_HIP_VERBOSE", "0")
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="mk13.b4")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--no-e2e", action="store_true")
args = ap.parse_args()

import torch
import spasm_amd
import workloads

A, rows, F, source = workloads.round0(args.workload, 42013)
print("%s: %d x %d, %d nnz; %d pivots, %d rows to reduce, %d non-pivotal columns" % (args.workload, A.n, A.m, A.nnz, F.U.n, len(rows), A.m - F.U.n), flush=True)
dev = torch.device("cuda:0")
dA = spasm_amd.DeviceCsr.from_host(A, dev)
drows = torch.from_numpy(np.ascontiguousarray(rows)).to(dev)
ref = None
for mode in ("1", "0"):
    os.environ["SPASM_HIP_BACKSOLVE"] = mode
    dF = spasm_amd.DeviceFact(F)
    pool = 1 << 28
    while True:
        W = spasm_amd.SchurWorkspace(len(rows), A.m, pool)
        S, st = spasm_amd.dschur(dA, drows, dF, W, fetch=True)
        if st.status == 0:
            break
        W.close()
        pool *= 2
    if mode == "1" and not st.used_backsolve:
        print("  back-substituted image: the factor is not eligible")
        W.close()
        dF.close()
        continue
    if ref is None:
        ref = S
    else:
        same = torch.equal(S.p, ref.p) and torch.equal(S.j[:st.nnz], ref.j[:st.nnz]) and torch.equal(S.x[:st.nnz], ref.x[:st.nnz])
        print("  same S on both paths:", bool(same))
    best = None
    for _ in range(args.steps):
        dF.forget()
        _, st = spasm_amd.dschur(dA, drows, dF, W, fetch=False)
        best = st.ms_total if best is None else min(best, st.ms_total)
    print("  %-24s %8.2f ms per step (%s%s), nnz(S) %d, density %.4f" % ("back-substituted image" if st.used_backsolve else "row by row", best, st.kernel.decode(),
          (" + " + st.kernel_other.decode()) if st.used_backsolve else "", st.nnz, st.nnz / (len(rows) * max(A.m - F.U.n, 1))), flush=True)
    W.close()
    dF.close()
    if mode == "0":
        del S
os.environ.pop("SPASM_HIP_BACKSOLVE", None)
del ref
if not args.no_e2e:
    for env in ({}, {"SPASM_HIP_BACKSOLVE": "0", "SPASM_HIP_DEVICE_FINISH": "0"}):
        os.environ.update(env)
        spasm_amd.echelonize(A)
        t0 = time.perf_counter()
        fact = spasm_amd.echelonize(A)
        print("  echelonize %s: rank %d in %.2f s %s" % (env or "(defaults)", fact.U.n, time.perf_counter() - t0, spasm_amd.echelonize_profile()), flush=True)
        for k in env:
            os.environ.pop(k)
