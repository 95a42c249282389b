"""GL7d19-class stand-ins (chessboard complexes) through spasm_hip_echelonize with the GL7d19 options
(--dense-threshold 0.01), verbose: which rounds run, on which kernels, for how long."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")          # (this script names kernel variants and debugging aids)
os.environ.setdefault("SPASM_HIP_VERBOSE", "1")
import workloads          # noqa: E402
import spasm_amd          # noqa: E402

names = sys.argv[1:] or ["ch7-8.b5", "ch8-8.b5"]
for name in names:
    thr = 0.01
    if "@" in name:
        name, thr = name.split("@")
        thr = float(thr)
    t = time.time()
    A, src = workloads.load_matrix(name)
    print("== %s: %d x %d, %d nnz (%s, %.1f s to build)" % (name, A.n, A.m, A.nnz, src, time.time() - t), flush=True)
    o = spasm_amd.default_opts()
    o.sparsity_threshold = thr
    for k in range(int(os.environ.get("PROBE_CALLS", "2"))):
        t = time.time()
        F = spasm_amd.echelonize(A, o)
        print("== %s thr %.3f call %d: rank %d, %.2f s, %s" % (name, thr, k, F.U.n, time.time() - t, spasm_amd.echelonize_profile()), flush=True)
