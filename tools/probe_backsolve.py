#!/usr/bin/env python3
"""GPU box: times the back-substituted path on a BASELINE workload for every kernel variant
(variant = <packed><shape>: packed 1 = 16-bit entries of R, two per word (p < 2^16), 0 = 32-bit; shape 0 = 128-byte slab rows
and 16 waves, 1 = 128 B and 8 waves, 2 = 64 B and 8 waves)
and checks that all of them produce the same S."""
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
ap.add_argument("--workload", default="mk13.b5")
ap.add_argument("--variants", default="10,11,12,00,01,02")
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--no-check", action="store_true")
args = ap.parse_args()

import torch
import spasm_amd
import workloads

A, rows, F, source = workloads.round0(args.workload, 42013)
dev = torch.device("cuda:0")
dA = spasm_amd.DeviceCsr.from_host(A, dev)
drows = torch.from_numpy(np.ascontiguousarray(rows)).to(dev)
ref = None
for v in args.variants.split(","):
    os.environ["SPASM_HIP_BS_SHAPE"] = v[1]
    dF = spasm_amd.DeviceFact(F)
    pool = 1 << 30
    W = spasm_amd.SchurWorkspace(len(rows), A.m, pool)
    S, st = spasm_amd.dschur(dA, drows, dF, W, fetch=True)
    assert st.status == 0, "pool too small"
    if ref is None:
        ref = S
    elif not args.no_check:
        assert torch.equal(S.p, ref.p) and torch.equal(S.j[:st.nnz], ref.j[:st.nnz]) and torch.equal(S.x[:st.nnz], ref.x[:st.nnz]), "variant %s differs" % v
    tb, ta, tf, tt = [], [], [], []
    for _ in range(args.steps):
        dF.forget()
        _, st = spasm_amd.dschur(dA, drows, dF, W, fetch=False)
        tb.append(st.ms_backsolve); ta.append(st.ms_apply); tf.append(st.ms_finalize); tt.append(st.ms_total)
    print("variant %s: backsolve %.3f ms, apply %.3f ms, finalize %.3f ms, total %.3f ms (min of %d); nnz %d" %
          (v, min(tb), min(ta), min(tf), min(tt), args.steps, st.nnz), flush=True)
    W.close()
    dF.close()
    if v != args.variants.split(",")[0]:
        del S
os.environ.pop("SPASM_HIP_BS_SHAPE", None)
