#!/usr/bin/env python3
"""GPU box: a stand-in near the size of GL7d19 end to end (default mk14.b5: 945,945 x 945,945, 5.7 M entries -- wide enough
that the pivot search keeps its reached-bits in HBM), twice, then its transpose: the three ranks must agree."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")          # (this script names kernel variants and debugging aids)
os.environ.setdefault("SPASM_HIP_VERBOSE", "1")
import spasm_amd
import workloads

name = sys.argv[1] if len(sys.argv) > 1 else "mk14.b5"
t0 = time.time()
A, src = workloads.load_matrix(name)
print("== %s: %d x %d, %d nnz (%s, %.1f s to build)" % (name, A.n, A.m, A.nnz, src, time.time() - t0), flush=True)
o = spasm_amd.default_opts()
if len(sys.argv) > 2:
    o.sparsity_threshold = float(sys.argv[2])
ranks = []
for k in range(2):
    t0 = time.time()
    F = spasm_amd.echelonize(A, o)
    ranks.append(F.U.n)
    print("== %s call %d: rank %d, %.2f s, %s" % (name, k, F.U.n, time.time() - t0, spasm_amd.echelonize_profile()), flush=True)
t0 = time.time()
Ft = spasm_amd.echelonize(spasm_amd.transpose(A), o)
ranks.append(Ft.U.n)
print("== %s transposed: rank %d, %.2f s, %s" % (name, Ft.U.n, time.time() - t0, spasm_amd.echelonize_profile()), flush=True)
print("ranks agree:", len(set(ranks)) == 1, ranks)
