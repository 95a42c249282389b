import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
os.environ["SPASM_HIP_VERBOSE"] = "2"
import workloads, spasm_amd
A, _ = workloads.load_matrix("mk13.b5")
spasm_amd.echelonize(A)
print("=========== second call", flush=True)
t = time.time(); F = spasm_amd.echelonize(A); print("rank", F.U.n, time.time() - t, spasm_amd.echelonize_profile())
