"""host pivot search (spasm_hip_pivots_extract_structural) against the thread count, on the box's cores."""
import os
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")          # (this script names kernel variants and debugging aids)
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["SPASM_HIP_VERBOSE"] = "0"
import workloads          # noqa: E402
import spasm_amd          # noqa: E402

names = sys.argv[1:] or ["mk13.b5", "ch7-8.b5", "ch8-8.b5"]
print("hardware threads:", os.cpu_count(), flush=True)
for name in names:
    A, _ = workloads.load_matrix(name)
    for T in (8, 16, 32, 64, 128, 256):
        if T > 2 * (os.cpu_count() or 1):
            break
        os.environ["SPASM_HIP_THREADS"] = str(T)
        best = None
        for rep in range(2):
            t = time.time()
            npiv, perm, F = spasm_amd.pivots_extract_structural(A, spasm_amd.empty_fact(A.m, 42013))
            dt = time.time() - t
            best = dt if best is None else min(best, dt)
        print("%s: %3d threads %.3f s, %d pivots" % (name, T, best, npiv), flush=True)
