import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")          # (this script names kernel variants and debugging aids)
os.environ.setdefault("SPASM_HIP_VERBOSE", "1")
import spasm_amd, workloads
A, src = workloads.load_matrix("mk14.b5")
for w in sys.argv[1:]:
    os.environ["SPASM_HIP_PIVOT_WAVES_PER_CU"] = w
    t0 = time.time()
    npiv, perm, F = spasm_amd.pivots_extract_structural(A, spasm_amd.empty_fact(A.m, A.prime))
    print("== waves/CU %s: %d pivots, %.2f s" % (w, npiv, time.time() - t0), flush=True)
