import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
os.environ["SPASM_HIP_VERBOSE"] = "1"
import workloads, spasm_amd
A, rows, F, src = workloads.round0("mk14.b4", 42013)
for k in range(2):
    t = time.time(); S, p_out = spasm_amd.schur(A, rows, F); print("host schur %.2f s, nnz %d" % (time.time() - t, S.nnz), flush=True)
