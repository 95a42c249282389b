#!/usr/bin/env python3
"""GPU box: the data behind the path rule.  For a family of generated matrices: the round-0 Schur
complement on both elimination paths, with the split of the back-substituted path (build of R / apply / expansion) and the
elimination count of the row-by-row path; prints one JSON line per matrix."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")          # (this script names kernel variants and debugging aids)
os.environ.setdefault("SPASM_HIP_VERBOSE", "0")
import torch
import spasm_amd
import workloads

names = sys.argv[1:] or ["mk13.b3", "ch7-8.b3", "ch8-8.b3", "mk12.b3", "ch7-7.b3", "mk11.b4", "ch6-7.b4", "mk13.b5", "ch7-7.b4", "ch7-8.b4", "mk12.b4",
                         "ch8-8.b4", "mk13.b4"]
dev = torch.device("cuda:0")
for name in names:
    A, rows, F, _ = workloads.round0(name, 42013)
    dA = spasm_amd.DeviceCsr.from_host(A, dev)
    drows = torch.from_numpy(np.ascontiguousarray(rows)).to(dev)
    out = {"name": name, "r": int(F.U.n), "rows": len(rows), "Sm": int(A.m - F.U.n), "nnz_U": int(F.U.nnz)}
    for mode in ("1", "0"):
        os.environ["SPASM_HIP_BACKSOLVE"] = mode
        dF = spasm_amd.DeviceFact(F)
        pool = 1 << 28
        while True:
            W = spasm_amd.SchurWorkspace(len(rows), A.m, pool)
            S, st = spasm_amd.dschur(dA, drows, dF, W, fetch=False)
            if st.status == 0:
                break
            W.close()
            pool *= 2
        best = None
        for _ in range(3):
            dF.forget()
            _, st = spasm_amd.dschur(dA, drows, dF, W, fetch=False)
            if best is None or st.ms_total < best.ms_total:
                best = st
        st = best
        if mode == "1" and st.used_backsolve:
            out.update({"bs_ms": st.ms_total, "build_ms": st.ms_backsolve, "apply_ms": st.ms_apply, "expand_ms": st.ms_expand,
                        "pivotal_entries": int(st.eliminations), "nnz_S": int(st.nnz), "levels": dF.levels})
        elif mode == "0":
            out.update({"rows_ms": st.ms_total, "eliminations": int(st.eliminations), "group_pivots": int(st.group_pivots),
                        "kernel": st.kernel.decode(), "nnz_S": int(st.nnz)})
        W.close()
        dF.close()
    os.environ.pop("SPASM_HIP_BACKSOLVE", None)
    out["density"] = out["nnz_S"] / (out["rows"] * max(out["Sm"], 1))
    print(json.dumps(out), flush=True)
