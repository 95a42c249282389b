import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
os.environ["SPASM_HIP_EXPERIMENT"]="1"; os.environ["SPASM_HIP_VERBOSE"]="1"; os.environ["SPASM_HIP_PIVOT_STATS"]="1"
import spasm_amd, workloads
name=sys.argv[1]
A,_=workloads.load_matrix(name)
for w in sys.argv[2:]:
    os.environ["SPASM_HIP_PIVOT_WAVES_PER_CU"]=w
    t=time.time()
    npiv, perm, F = spasm_amd.pivots_extract_structural(A, spasm_amd.empty_fact(A.m, A.prime))
    print("== %s waves/CU %s: %d pivots, %.3f s" % (name, w, npiv, time.time()-t), flush=True)
