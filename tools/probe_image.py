"""how long does the factor image of a workload take to build (host plan + upload), and where?  SPASM_HIP_VERBOSE=2 prints the split."""
import os
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")          # (this script names kernel variants and debugging aids)
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("SPASM_HIP_VERBOSE", "2")
import torch
import spasm_amd
import workloads
for name in sys.argv[1:] or ["mk13.b5"]:
    A, rows, F, _ = workloads.round0(name, 42013, threads=0 if name.startswith("mk14") else 1)
    print(name, "pivots", F.U.n, "nnz(U)", F.U.nnz, flush=True)
    for _ in range(3):
        t0 = time.perf_counter()
        dF = spasm_amd.DeviceFact(F)
        torch.cuda.synchronize()
        print("  image %.1f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
        dF.close()
