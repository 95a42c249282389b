#!/usr/bin/env python3
"""GPU box: spasm_hip_echelonize on a generated stand-in, N calls, with the log of the driver (which path every Schur
complement took, the time split).  python tools/probe_e2e.py <name> [calls] [dense-threshold]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")          # (this script names kernel variants and debugging aids)
os.environ.setdefault("SPASM_HIP_VERBOSE", "2")
import spasm_amd
import workloads

name = sys.argv[1] if len(sys.argv) > 1 else "mk14.b4"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 2
t0 = time.time()
A, src = workloads.load_matrix(name)
print("== %s: %d x %d, %d nnz (%s, %.1f s to build)" % (name, A.n, A.m, A.nnz, src, time.time() - t0), flush=True)
o = spasm_amd.default_opts()
if len(sys.argv) > 3:
    o.sparsity_threshold = float(sys.argv[3])
ranks = []
for k in range(calls):
    t0 = time.time()
    F = spasm_amd.echelonize(A, o)
    ranks.append(F.U.n)
    print("== %s call %d: rank %d, %.2f s, %s %s" % (name, k, F.U.n, time.time() - t0, spasm_amd.echelonize_profile(),
                                                    {a: b for a, b in spasm_amd.echelonize_counters().items() if not a.startswith("pivot")}), flush=True)
print("ranks agree:", len(set(ranks)) == 1, ranks)
