import os
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")
import sys, time, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
os.environ["SPASM_HIP_VERBOSE"]="1"
import workloads, spasm_amd
A,_ = workloads.load_matrix("mk13.b4")
o = spasm_amd.default_opts()
o.sparsity_threshold = 0.1          # let the sparse round run (estimated density 0.05)
for k in range(2):
    t=time.time(); F = spasm_amd.echelonize(A, o); print("rank", F.U.n, "%.2f s" % (time.time()-t), spasm_amd.echelonize_profile(), flush=True)
