"""Host-pointer entry points: same names, argument meaning and results as the reference's C API."""
import ctypes as C

import numpy as np

from ._lib import lib, require_gpu
from .matrix import Csr, Fact, CLu, EchelonizeOpts, view_csr, copy_csr

_libc = C.CDLL(None)
_libc.fopen.restype = C.c_void_p
_libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
_libc.fclose.argtypes = [C.c_void_p]


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def load(path, prime, transpose_if_wide=False):
    """spasm_triplet_load + spasm_compress (spasm_io.c:60, spasm_triplet.c:97) of an SMS / MatrixMarket file."""
    L = lib()
    f = _libc.fopen(path.encode(), b"r")
    if not f:
        raise OSError("cannot open %s" % path)
    try:
        T = L.spasm_hip_triplet_load(f, prime, None)
    finally:
        _libc.fclose(f)
    if transpose_if_wide and T.contents.n < T.contents.m:
        L.spasm_hip_triplet_transpose(T)
    A = L.spasm_hip_compress(T)
    out = copy_csr(A)
    L.spasm_hip_csr_free(A)
    L.spasm_hip_triplet_free(T)
    return out


def compress(prime, n, m, ti, tj, tx):
    """triplets (0-based) -> Csr through spasm_hip_add_entry / spasm_hip_compress."""
    L = lib()
    T = L.spasm_hip_triplet_alloc(n, m, max(len(ti), 1), prime, True)
    for a, b, c in zip(np.asarray(ti).tolist(), np.asarray(tj).tolist(), np.asarray(tx).tolist()):
        L.spasm_hip_add_entry(T, a, b, c)
    A = L.spasm_hip_compress(T)
    out = copy_csr(A)
    L.spasm_hip_csr_free(A)
    L.spasm_hip_triplet_free(T)
    return out


def transpose(A):
    L = lib()
    a = view_csr(A)
    t = L.spasm_hip_transpose(C.byref(a), 1)
    out = copy_csr(t)
    L.spasm_hip_csr_free(t)
    return out


def empty_fact(m, prime):
    return Fact(Csr(0, m, np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32), prime),
                np.full(m, -1, np.int32))


def _lu_for(F, extra_rows, extra_nz):
    """library-owned struct spasm_lu initialised from a Fact, with room to grow."""
    L = lib()
    U = F.U
    up = L.spasm_hip_csr_alloc(U.n + extra_rows, U.m, max(U.nnz + extra_nz, 1), U.prime, True)
    s = up.contents
    C.memmove(s.p, U.p.ctypes.data, 8 * (U.n + 1))
    if U.nnz:
        C.memmove(s.j, U.j.ctypes.data, 4 * U.nnz)
        C.memmove(s.x, U.x.ctypes.data, 4 * U.nnz)
    up.contents.n = U.n
    qinv = F.qinv.copy()
    lu = CLu()
    lu.r = U.n
    lu.complete = False
    lu.L = None
    lu.U = up
    lu.qinv = _ip(qinv)
    lu.p = None
    lu.Ltmp = None
    return lu, up, qinv


def pivots_extract_structural(A, F, greedy=True):
    """spasm_pivots_extract_structural (spasm_pivots.c:369): returns (npiv, p, F + new pivotal rows)."""
    L = lib()
    a = view_csr(A)
    lu, up, qinv = _lu_for(F, A.n, A.nnz)
    opts = EchelonizeOpts()
    opts.enable_greedy_pivot_search = bool(greedy)
    p = np.zeros(max(A.n, 1), np.int32)
    npiv = L.spasm_hip_pivots_extract_structural(C.byref(a), None, C.byref(lu), _ip(p), C.byref(opts))
    out = Fact(copy_csr(lu.U), qinv)
    L.spasm_hip_csr_free(lu.U)
    return npiv, p[:A.n], out


def schur(A, p, F, p_in=None, want_L=False):
    """spasm_schur (spasm_schur.c:61) on the GPU: returns (S, p_out), or (S, p_out, (Li, Lj, Lx)) with
    want_L: the elimination coefficients as triplets (row, index of the pivot row in U, value)."""
    require_gpu("schur")
    L = lib()
    a = view_csr(A)
    lu, up, qinv = _lu_for(F, 0, 0)
    p = np.ascontiguousarray(p, np.int32)
    n = len(p)
    p_out = np.zeros(max(n, 1), np.int32)
    pin = _ip(np.ascontiguousarray(p_in, np.int32)) if p_in is not None else None
    T = L.spasm_hip_triplet_alloc(max(A.n, 1), max(F.U.n, 1), 16, A.prime, True) if want_L else None
    s = L.spasm_hip_schur(C.byref(a), _ip(p), n, C.byref(lu), -1.0, T, pin, _ip(p_out))
    S = copy_csr(s)
    L.spasm_hip_csr_free(s)
    L.spasm_hip_csr_free(up)
    if not want_L:
        return S, p_out[:n]
    t = T.contents
    nz = int(t.nz)
    trip = tuple(np.ctypeslib.as_array(arr, shape=(max(nz, 1),))[:nz].copy() for arr in (t.i, t.j, t.x))
    L.spasm_hip_triplet_free(T)
    return S, p_out[:n], trip


class ResidentSchur:
    """spasm_hip_schur_resident on fixed arguments, call after call: spasm_hip_schur the way spasm_hip_echelonize calls it between
    two rounds (entries of S left on the device, the installed communicator in force) -- what bench.py times on several GPUs.
    forget=True: the cached factor images drop R before every call, so that a call pays for all of spasm_schur."""

    def __init__(self, A, p, F):
        require_gpu("ResidentSchur")
        self._a = view_csr(A)
        self._lu, self._up, self._qinv = _lu_for(F, 0, 0)
        self._p = np.ascontiguousarray(p, np.int32)
        self._keep = (A, F)

    def __call__(self, forget=True, est_density=-1.0):
        L = lib()
        if forget:
            L.spasm_hip_forget_cached_images()
        return int(L.spasm_hip_schur_resident(C.byref(self._a), _ip(self._p), len(self._p), C.byref(self._lu), est_density))

    def close(self):
        if self._up is not None:
            lib().spasm_hip_csr_free(self._up)
            self._up = None


SPASM_DOUBLE, SPASM_FLOAT, SPASM_I64 = 0, 1, 2      # spasm_datatype (spasm.h:139)
_NP_OF = {SPASM_DOUBLE: np.float64, SPASM_FLOAT: np.float32, SPASM_I64: np.int64}


def schur_dense(A, p, F, p_in=None, datatype=SPASM_I64):
    """spasm_schur_dense (spasm_schur.c:257) on the GPU: returns (S [n, Sm], q, p_out)."""
    require_gpu("schur_dense")
    L = lib()
    a = view_csr(A)
    lu, up, qinv = _lu_for(F, 0, 0)
    p = np.ascontiguousarray(p, np.int32)
    n = len(p)
    Sm = A.m - F.U.n
    S = np.zeros(max(n * Sm, 1), _NP_OF[datatype])
    q = np.zeros(max(Sm, 1), np.int32)
    p_out = np.zeros(max(n, 1), np.int32)
    pin = _ip(np.ascontiguousarray(p_in, np.int32)) if p_in is not None else None
    L.spasm_hip_schur_dense(C.byref(a), _ip(p), n, pin, C.byref(lu), S.ctypes.data, datatype, _ip(q), _ip(p_out))
    L.spasm_hip_csr_free(up)
    return S[:n * Sm].reshape(n, Sm), q[:Sm], p_out[:n]


def ffpack_rref(prime, M, datatype=SPASM_I64):
    """spasm_ffpack_rref (spasm_ffpack.cpp:88) on the GPU, on a copy of M: returns (rank, R, qinv)."""
    require_gpu("ffpack_rref")
    L = lib()
    M = np.ascontiguousarray(M, _NP_OF[datatype]).copy()
    n, m = M.shape
    qinv = np.zeros(max(m, 1), np.uint64)
    r = L.spasm_hip_ffpack_rref(prime, n, m, M.ctypes.data, m, datatype, qinv.ctypes.data_as(C.POINTER(C.c_size_t)))
    return r, M, qinv[:m].astype(np.int64)


def ffpack_LU(prime, M, datatype=SPASM_I64):
    """spasm_ffpack_LU (spasm_ffpack.cpp:137) on the GPU, on a copy of M: returns (rank, packed LU, P, Qinv)."""
    require_gpu("ffpack_LU")
    L = lib()
    M = np.ascontiguousarray(M, _NP_OF[datatype]).copy()
    n, m = M.shape
    P = np.zeros(max(n, 1), np.uint64)
    Q = np.zeros(max(m, 1), np.uint64)
    r = L.spasm_hip_ffpack_LU(prime, n, m, M.ctypes.data, m, datatype, P.ctypes.data_as(C.POINTER(C.c_size_t)),
                              Q.ctypes.data_as(C.POINTER(C.c_size_t)))
    return r, M, P[:n].astype(np.int64), Q[:m].astype(np.int64)


def default_opts():
    o = EchelonizeOpts()
    lib().spasm_hip_echelonize_init_opts(C.byref(o))
    return o


def echelonize(A, opts=None):
    """spasm_echelonize (spasm_echelonize.c:473): returns Fact(U, qinv) with rank = U.n."""
    require_gpu("echelonize")
    L = lib()
    a = view_csr(A)
    lu = L.spasm_hip_echelonize(C.byref(a), C.byref(opts) if opts is not None else None)
    s = lu.contents
    U = copy_csr(s.U)
    qinv = np.ctypeslib.as_array(s.qinv, shape=(max(A.m, 1),))[:A.m].copy()
    F = Fact(U, qinv)
    F.L, F.Lp = None, None
    if bool(s.L):                       # opts.L: A == L * U, pivot j of L sits on row Lp[j]
        F.L = copy_csr(s.L)
        F.Lp = np.ctypeslib.as_array(s.p, shape=(max(U.n, 1),))[:U.n].copy()
    L.spasm_hip_lu_free(lu)
    return F


def echelonize_profile():
    """seconds of the last echelonize() call: dict(total, pivot_search, density_estimates, sparse_schur, dense_finish,
    sparse_rounds, structural_finish)."""
    out = (C.c_double * 8)()
    lib().spasm_hip_echelonize_profile(out)
    return {"total": out[0], "pivot_search": out[1], "density_estimates": out[2], "sparse_schur": out[3],
            "dense_finish": out[4], "sparse_rounds": int(out[5]), "structural_finish": out[6], "uploads_so_far": int(out[7])}


COUNTER_NAMES = ("pool_retries", "pools_sized_from_a_sample", "sparse_image_chunk_extensions", "sparse_image_build_aborts",
                 "block_cache_misses", "block_cache_miss_bytes", "factor_plans", "pivot_visits", "pivot_visits_of_searches_with_a_pivot",
                 "pivot_cascade_items", "pivot_rows_with_a_pivot", "pivot_rows_without", "pivots_accepted_on_labels_alone",
                 "pivot_rows_deferred_to_the_ticket_search", "schur_complements_kept_as_column_slabs", "column_slabs_gathered_into_whole_rows")


def echelonize_counters():
    """events since the last echelonize() call started that its time split does not show (spasm_hip_echelonize_counters):
    pool retries, extensions of the sparse image, block-cache misses, factor plans, visits of the pivot search by outcome."""
    out = (C.c_longlong * len(COUNTER_NAMES))()
    lib().spasm_hip_echelonize_counters(out, len(COUNTER_NAMES))
    return {k: int(out[t]) for t, k in enumerate(COUNTER_NAMES)}


def rref(F):
    """spasm_rref (spasm_rref.c:25): returns (R, Rqinv)."""
    require_gpu("rref")
    L = lib()
    lu, up, qinv = _lu_for(F, 0, 0)
    Rq = np.zeros(max(F.U.m, 1), np.int32)
    r = L.spasm_hip_rref(C.byref(lu), _ip(Rq))
    R = copy_csr(r)
    L.spasm_hip_csr_free(r)
    L.spasm_hip_csr_free(up)
    return R, Rq[:F.U.m]


def kernel(F):
    """spasm_kernel (spasm_kernel.c:9): basis of the right kernel, one vector per row."""
    require_gpu("kernel")
    L = lib()
    lu, up, qinv = _lu_for(F, 0, 0)
    k = L.spasm_hip_kernel(C.byref(lu))
    K = copy_csr(k)
    L.spasm_hip_csr_free(k)
    L.spasm_hip_csr_free(up)
    return K
