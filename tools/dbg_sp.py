import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.environ["SPASM_HIP_SPARSE_IMAGE"] = "1"
os.environ["SPASM_HIP_VERBOSE"] = "3"
import numpy as np, torch
import spasm_amd
from oracle import oracle
import test_gpu_sparse_image as T
p = 42013
rng = np.random.default_rng(3)
npiv = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n, m, ti, tj, tx = T._triangular_system(rng, p, npiv=npiv, nnon=9000, nred=800, deps=lambda k: 3, reach=300, np_per_row=4, red_entries=6)
A = oracle.compress(p, n, m, ti, tj, tx)
npiv, perm, F = oracle.pivots_extract_structural(A, oracle.empty_fact(A.n, A.m, p))
rows = perm[npiv:]
want, _, _ = oracle.schur(A, rows, F)
t = time.time()
S, st, W, dF, dA, drows = T._stats_of_device_call(A, rows, F, 1 << 22)
print("full", st.status, st.used_sparse_image, st.sparse_image_built, st.kernel, st.nnz, want.nnz, "%.2f s" % (time.time() - t), "build ms", st.ms_sparse_build, flush=True)
H = S.to_host()
print("same", oracle.same_matrix(oracle.CSR(H.n, H.m, H.p, H.j, H.x, p), want))
