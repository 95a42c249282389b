"""ctypes front-end to the TEST-ONLY checkers under oracle/.

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module.  It wraps

  * oracle/liboracle.so           -- the C restatement (spasm_oracle.c)
  * oracle/_ref/libspasm_ref.so   -- the real reference, when it was built
                                     (see oracle/Makefile); `ref_available()`.

Matrices cross the boundary as the small `CSR` value class below (numpy
arrays, reference field names: n rows, m columns, p/j/x as in spasm.h:37-50).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class CSR:
    """n x m matrix mod prime in compressed-row form (host, numpy)."""

    def __init__(self, n, m, p, j, x, prime):
        self.n = int(n)
        self.m = int(m)
        self.p = np.ascontiguousarray(p, dtype=np.int64)
        self.j = np.ascontiguousarray(j, dtype=np.int32)
        self.x = np.ascontiguousarray(x, dtype=np.int32)
        self.prime = int(prime)
        assert self.p.shape[0] == self.n + 1

    @property
    def nnz(self):
        return int(self.p[self.n])

    def row(self, i):
        lo, hi = int(self.p[i]), int(self.p[i + 1])
        return self.j[lo:hi], self.x[lo:hi]

    def canonical(self):
        """rows as sorted (col, value mod p in [0,p)) -- order-free comparison form."""
        out = []
        P = self.prime
        for i in range(self.n):
            jj, xx = self.row(i)
            order = np.argsort(jj, kind="stable")
            out.append((jj[order].astype(np.int64), np.mod(xx[order].astype(np.int64), P)))
        return out

    def to_dense(self):
        D = np.zeros((self.n, self.m), dtype=np.int64)
        for i in range(self.n):
            jj, xx = self.row(i)
            D[i, jj] = np.mod(xx.astype(np.int64), self.prime)
        return D


def same_matrix(A, B):
    """True when A and B are equal as matrices mod p (row order kept, entry order free)."""
    if (A.n, A.m, A.prime) != (B.n, B.m, B.prime) or A.nnz != B.nnz:
        return False
    for (ja, xa), (jb, xb) in zip(A.canonical(), B.canonical()):
        if ja.shape != jb.shape or not np.array_equal(ja, jb) or not np.array_equal(xa, xb):
            return False
    return True


# --------------------------------------------------------------------------
# SMS reader for the test fixtures (format of spasm_io.c:60-170: "n m M",
# 1-based "i j x" triplets, terminated by "0 0 0").
# --------------------------------------------------------------------------
def read_sms(path):
    with open(path) as f:
        header = f.readline().split()
        n, m = int(header[0]), int(header[1])
        ti, tj, tx = [], [], []
        for line in f:
            parts = line.split()
            if len(parts) < 3:
                continue
            i, j, x = int(parts[0]), int(parts[1]), int(parts[2])
            if i == 0 and j == 0 and x == 0:
                break
            ti.append(i - 1)
            tj.append(j - 1)
            tx.append(x)
    return n, m, np.array(ti, dtype=np.int32), np.array(tj, dtype=np.int32), np.array(tx, dtype=np.int64)


# --------------------------------------------------------------------------
# liboracle.so
# --------------------------------------------------------------------------
class _OrcCsr(C.Structure):
    _fields_ = [("nzmax", C.c_int64), ("n", C.c_int), ("m", C.c_int),
                ("p", C.POINTER(C.c_int64)), ("j", C.POINTER(C.c_int)),
                ("x", C.POINTER(C.c_int32)), ("prime", C.c_int64)]


class _OrcLu(C.Structure):
    _fields_ = [("U", C.POINTER(_OrcCsr)), ("qinv", C.POINTER(C.c_int)), ("r", C.c_int),
                ("lnz", C.c_int64), ("lnzmax", C.c_int64),
                ("Li", C.POINTER(C.c_int)), ("Lj", C.POINTER(C.c_int)), ("Lx", C.POINTER(C.c_int32)),
                ("Lp", C.POINTER(C.c_int)), ("want_L", C.c_int)]


class OrcOpts(C.Structure):
    _fields_ = [("enable_greedy_pivot_search", C.c_int), ("enable_tall_and_skinny", C.c_int),
                ("enable_dense", C.c_int), ("enable_GPLU", C.c_int), ("L", C.c_int),
                ("complete", C.c_int), ("min_pivot_proportion", C.c_double), ("max_round", C.c_int),
                ("sparsity_threshold", C.c_double), ("dense_block_size", C.c_int),
                ("low_rank_ratio", C.c_double), ("tall_and_skinny_ratio", C.c_double),
                ("low_rank_start_weight", C.c_double)]


_lib = None


def build(force=False):
    """(re)build liboracle.so and, when /root/reference exists, _ref/libspasm_ref.so."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "spasm_oracle.c")
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src)
    ref_missing = os.path.isdir("/root/reference/src") and not os.path.exists(
        os.path.join(_HERE, "_ref", "libspasm_ref.so"))
    if force or stale or ref_missing:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(os.path.join(_HERE, "liboracle.so"))
    i64, i32, ci = C.c_int64, C.c_int32, C.c_int
    pcsr, plu = C.POINTER(_OrcCsr), C.POINTER(_OrcLu)
    pint, pi64, pi32 = C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    for name, args in [("orc_zp_add", [i64, i32, i32]), ("orc_zp_sub", [i64, i32, i32]),
                       ("orc_zp_mul", [i64, i32, i32]), ("orc_zp_inverse", [i64, i32]),
                       ("orc_zp_axpy", [i64, i32, i32, i32]), ("orc_zp_init", [i64, i64])]:
        getattr(L, name).restype = i32
        getattr(L, name).argtypes = args
    L.orc_csr_alloc.restype = pcsr
    L.orc_csr_alloc.argtypes = [ci, ci, i64, i64]
    L.orc_csr_free.argtypes = [pcsr]
    L.orc_compress.restype = pcsr
    L.orc_compress.argtypes = [i64, ci, ci, i64, pint, pint, pi64]
    L.orc_transpose.restype = pcsr
    L.orc_transpose.argtypes = [pcsr]
    L.orc_lu_alloc.restype = plu
    L.orc_lu_alloc.argtypes = [ci, ci, i64, i64, ci]
    L.orc_lu_free.argtypes = [plu]
    L.orc_sparse_triangular_solve.restype = ci
    L.orc_sparse_triangular_solve.argtypes = [pcsr, pcsr, ci, pint, pi32, pint]
    L.orc_pivots_extract_structural.restype = ci
    L.orc_pivots_extract_structural.argtypes = [pcsr, pint, plu, pint, ci]
    L.orc_schur_estimate_density.restype = C.c_double
    L.orc_schur_estimate_density.argtypes = [pcsr, pint, ci, pcsr, pint, ci, C.c_uint]
    L.orc_schur.restype = pcsr
    L.orc_schur.argtypes = [pcsr, pint, ci, plu, pint, pint]
    L.orc_schur_dense.argtypes = [pcsr, pint, ci, pint, plu, pi64, pint, pint]
    L.orc_dense_rref.restype = ci
    L.orc_dense_rref.argtypes = [i64, ci, ci, pi64, ci, pi64]
    L.orc_opts_init.argtypes = [C.POINTER(OrcOpts)]
    L.orc_echelonize.restype = plu
    L.orc_echelonize.argtypes = [pcsr, C.POINTER(OrcOpts)]
    L.orc_rref.restype = pcsr
    L.orc_rref.argtypes = [plu, pint]
    _lib = L
    return L


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _from_orc(ptr):
    s = ptr.contents
    n = s.n
    p = np.ctypeslib.as_array(s.p, shape=(n + 1,)).copy()
    nnz = int(p[n])
    j = np.ctypeslib.as_array(s.j, shape=(max(nnz, 1),))[:nnz].copy() if nnz else np.zeros(0, np.int32)
    x = np.ctypeslib.as_array(s.x, shape=(max(nnz, 1),))[:nnz].copy() if nnz else np.zeros(0, np.int32)
    return CSR(n, s.m, p, j, x, s.prime)


def _to_orc(A):
    """returns an owned orc_csr* holding a copy of A."""
    L = lib()
    ptr = L.orc_csr_alloc(A.n, A.m, max(A.nnz, 1), A.prime)
    s = ptr.contents
    C.memmove(s.p, A.p.ctypes.data, 8 * (A.n + 1))
    if A.nnz:
        C.memmove(s.j, A.j.ctypes.data, 4 * A.nnz)
        C.memmove(s.x, A.x.ctypes.data, 4 * A.nnz)
    return ptr


def compress(prime, n, m, ti, tj, tx):
    """orc_compress: triplets (0-based, int64 values) -> CSR."""
    L = lib()
    ti = np.ascontiguousarray(ti, np.int32)
    tj = np.ascontiguousarray(tj, np.int32)
    tx = np.ascontiguousarray(tx, np.int64)
    ptr = L.orc_compress(prime, n, m, len(ti), _ip(ti), _ip(tj),
                         tx.ctypes.data_as(C.POINTER(C.c_int64)))
    out = _from_orc(ptr)
    L.orc_csr_free(ptr)
    return out


def load_sms(path, prime):
    n, m, ti, tj, tx = read_sms(path)
    return compress(prime, n, m, ti, tj, tx)


def transpose(A):
    L = lib()
    a = _to_orc(A)
    t = L.orc_transpose(a)
    out = _from_orc(t)
    L.orc_csr_free(a)
    L.orc_csr_free(t)
    return out


class Fact:
    """U (CSR, pivot first in each row, unit pivots), qinv, optional L triplets."""

    def __init__(self, U, qinv, L=None, Lp=None):
        self.U = U
        self.qinv = np.ascontiguousarray(qinv, np.int32)
        self.L = L      # (Li, Lj, Lx) or None
        self.Lp = Lp

    @property
    def r(self):
        return self.U.n


def _lu_to_py(plu, n_rows_for_Lp):
    s = plu.contents
    U = _from_orc(s.U)
    qinv = np.ctypeslib.as_array(s.qinv, shape=(U.m,)).copy() if U.m else np.zeros(0, np.int32)
    Ltrip, Lp = None, None
    if s.want_L:
        k = int(s.lnz)
        if k:
            Ltrip = (np.ctypeslib.as_array(s.Li, shape=(k,)).copy(),
                     np.ctypeslib.as_array(s.Lj, shape=(k,)).copy(),
                     np.ctypeslib.as_array(s.Lx, shape=(k,)).copy())
        else:
            Ltrip = (np.zeros(0, np.int32),) * 3
        Lp = np.ctypeslib.as_array(s.Lp, shape=(max(n_rows_for_Lp, 1),))[:n_rows_for_Lp].copy()
    return Fact(U, qinv, Ltrip, Lp)


def _py_to_lu(F, n_rows, want_L=False):
    """owned orc_lu* initialised from a Fact (U rows + qinv)."""
    L = lib()
    U = F.U
    plu = L.orc_lu_alloc(max(n_rows, U.n), U.m, max(U.nnz, 1) + 16, U.prime, 1 if want_L else 0)
    s = plu.contents
    u = s.U.contents
    C.memmove(u.p, U.p.ctypes.data, 8 * (U.n + 1))
    if U.nnz:
        C.memmove(u.j, U.j.ctypes.data, 4 * U.nnz)
        C.memmove(u.x, U.x.ctypes.data, 4 * U.nnz)
    s.U.contents.n = U.n
    if U.m:
        C.memmove(s.qinv, F.qinv.ctypes.data, 4 * U.m)
    s.r = U.n
    return plu


def empty_fact(n, m, prime):
    return Fact(CSR(0, m, np.zeros(1, np.int64), np.zeros(0, np.int32), np.zeros(0, np.int32), prime),
                np.full(m, -1, np.int32))


def pivots_extract_structural(A, F, p_in=None, greedy=True, want_L=False):
    """returns (npiv, p, F') -- F' is F with the new pivotal rows appended."""
    L = lib()
    a = _to_orc(A)
    plu = _py_to_lu(F, A.n + F.U.n, want_L)
    p = np.zeros(max(A.n, 1), np.int32)
    pin = _ip(np.ascontiguousarray(p_in, np.int32)) if p_in is not None else None
    npiv = L.orc_pivots_extract_structural(a, pin, plu, _ip(p), 1 if greedy else 0)
    out = _lu_to_py(plu, A.n + F.U.n)
    L.orc_csr_free(a)
    L.orc_lu_free(plu)
    return npiv, p[:A.n], out


def schur(A, p, F, p_in=None, want_L=False):
    """orc_schur on rows p of A w.r.t. F.  returns (S, p_out, L-triplets or None)."""
    L = lib()
    a = _to_orc(A)
    plu = _py_to_lu(F, A.n + F.U.n, want_L)
    p = np.ascontiguousarray(p, np.int32)
    n = len(p)
    p_out = np.zeros(max(n, 1), np.int32)
    pin = _ip(np.ascontiguousarray(p_in, np.int32)) if p_in is not None else None
    s = L.orc_schur(a, _ip(p), n, plu, pin, _ip(p_out))
    S = _from_orc(s)
    f2 = _lu_to_py(plu, 0)
    L.orc_csr_free(s)
    L.orc_csr_free(a)
    L.orc_lu_free(plu)
    return S, p_out[:n], f2.L


def schur_dense(A, p, F, p_in=None):
    """orc_schur_dense.  returns (S int64 [n, Sm], q, p_out)."""
    L = lib()
    a = _to_orc(A)
    plu = _py_to_lu(F, A.n + F.U.n, False)
    p = np.ascontiguousarray(p, np.int32)
    n = len(p)
    Sm = A.m - F.U.n
    S = np.zeros((max(n, 1), max(Sm, 1)), np.int64)
    q = np.zeros(max(Sm, 1), np.int32)
    p_out = np.zeros(max(n, 1), np.int32)
    pin = _ip(np.ascontiguousarray(p_in, np.int32)) if p_in is not None else None
    Sflat = np.zeros(max(n * Sm, 1), np.int64)
    L.orc_schur_dense(a, _ip(p), n, pin, plu, Sflat.ctypes.data_as(C.POINTER(C.c_int64)), _ip(q), _ip(p_out))
    S = Sflat[:n * Sm].reshape(n, Sm) if n * Sm else np.zeros((n, Sm), np.int64)
    L.orc_csr_free(a)
    L.orc_lu_free(plu)
    return S, q[:Sm], p_out[:n]


def dense_rref(prime, M):
    """orc_dense_rref on a copy of the int64 matrix M.  returns (rank, R, qinv)."""
    L = lib()
    M = np.ascontiguousarray(M, np.int64).copy()
    n, m = M.shape
    qinv = np.zeros(max(m, 1), np.int64)
    r = L.orc_dense_rref(prime, n, m, M.ctypes.data_as(C.POINTER(C.c_int64)), m,
                         qinv.ctypes.data_as(C.POINTER(C.c_int64)))
    return r, M, qinv[:m]


def default_opts():
    o = OrcOpts()
    lib().orc_opts_init(C.byref(o))
    return o


def echelonize(A, opts=None):
    L = lib()
    a = _to_orc(A)
    plu = L.orc_echelonize(a, C.byref(opts) if opts is not None else None)
    out = _lu_to_py(plu, A.n)
    L.orc_csr_free(a)
    L.orc_lu_free(plu)
    return out


def rref(F):
    L = lib()
    plu = _py_to_lu(F, F.U.n)
    Rq = np.zeros(max(F.U.m, 1), np.int32)
    r = L.orc_rref(plu, _ip(Rq))
    R = _from_orc(r)
    L.orc_csr_free(r)
    L.orc_lu_free(plu)
    return R, Rq[:F.U.m]


def solve_row(U, qinv, B, k):
    """orc_sparse_triangular_solve: returns (pattern xj[top:m], dense x)."""
    L = lib()
    u = _to_orc(U)
    b = _to_orc(B)
    m = U.m
    xj = np.zeros(3 * max(m, 1), np.int32)
    x = np.zeros(max(m, 1), np.int32)
    q = np.ascontiguousarray(qinv, np.int32)
    top = L.orc_sparse_triangular_solve(u, b, k, _ip(xj), x.ctypes.data_as(C.POINTER(C.c_int32)), _ip(q))
    L.orc_csr_free(u)
    L.orc_csr_free(b)
    return xj[top:m].copy(), x[:m]


# --------------------------------------------------------------------------
# the real reference (oracle/_ref/libspasm_ref.so)
# --------------------------------------------------------------------------
class _RefField(C.Structure):
    _fields_ = [("p", C.c_int64), ("halfp", C.c_int64), ("mhalfp", C.c_int64), ("dinvp", C.c_double)]


class _RefCsr(C.Structure):       # struct spasm_csr, spasm.h:37-50
    _fields_ = [("nzmax", C.c_int64), ("n", C.c_int), ("m", C.c_int),
                ("p", C.POINTER(C.c_int64)), ("j", C.POINTER(C.c_int)),
                ("x", C.POINTER(C.c_int32)), ("field", _RefField)]


class _RefTriplet(C.Structure):   # struct spasm_triplet, spasm.h:52-61
    _fields_ = [("nzmax", C.c_int64), ("nz", C.c_int64), ("n", C.c_int), ("m", C.c_int),
                ("i", C.POINTER(C.c_int)), ("j", C.POINTER(C.c_int)),
                ("x", C.POINTER(C.c_int32)), ("field", _RefField)]


class _RefLu(C.Structure):        # struct spasm_lu, spasm.h:63-71
    _fields_ = [("r", C.c_int), ("complete", C.c_bool), ("L", C.POINTER(_RefCsr)),
                ("U", C.POINTER(_RefCsr)), ("qinv", C.POINTER(C.c_int)),
                ("p", C.POINTER(C.c_int)), ("Ltmp", C.POINTER(_RefTriplet))]


class _RefOpts(C.Structure):      # struct echelonize_opts, spasm.h:84-108
    _fields_ = [("enable_greedy_pivot_search", C.c_bool), ("enable_tall_and_skinny", C.c_bool),
                ("enable_dense", C.c_bool), ("enable_GPLU", C.c_bool), ("L", C.c_bool),
                ("complete", C.c_bool), ("min_pivot_proportion", C.c_double), ("max_round", C.c_int),
                ("sparsity_threshold", C.c_double), ("dense_block_size", C.c_int),
                ("low_rank_ratio", C.c_double), ("tall_and_skinny_ratio", C.c_double),
                ("low_rank_start_weight", C.c_double)]


_ref = None


def ref_path():
    return os.path.join(_HERE, "_ref", "libspasm_ref.so")


def ref_available():
    build()
    return os.path.exists(ref_path())


def ref():
    global _ref
    if _ref is not None:
        return _ref
    if not ref_available():
        raise RuntimeError("oracle/_ref/libspasm_ref.so was not built (no /root/reference here)")
    R = C.CDLL(ref_path())
    i64, i32, ci = C.c_int64, C.c_int32, C.c_int
    pc, pt, pl = C.POINTER(_RefCsr), C.POINTER(_RefTriplet), C.POINTER(_RefLu)
    pint = C.POINTER(C.c_int)
    pf = C.POINTER(_RefField)
    R.spasm_field_init.argtypes = [i64, pf]
    for name, nargs in [("spasm_ZZp_add", 2), ("spasm_ZZp_sub", 2), ("spasm_ZZp_mul", 2),
                        ("spasm_ZZp_inverse", 1), ("spasm_ZZp_axpy", 3)]:
        getattr(R, name).restype = i32
        getattr(R, name).argtypes = [pf] + [i32] * nargs
    R.spasm_ZZp_init.restype = i32
    R.spasm_ZZp_init.argtypes = [pf, i64]
    R.spasm_triplet_alloc.restype = pt
    R.spasm_triplet_alloc.argtypes = [ci, ci, i64, i64, C.c_bool]
    R.spasm_add_entry.argtypes = [pt, ci, ci, i64]
    R.spasm_triplet_free.argtypes = [pt]
    R.spasm_compress.restype = pc
    R.spasm_compress.argtypes = [pt]
    R.spasm_csr_alloc.restype = pc
    R.spasm_csr_alloc.argtypes = [ci, ci, i64, i64, C.c_bool]
    R.spasm_csr_free.argtypes = [pc]
    R.spasm_sparse_triangular_solve.restype = ci
    R.spasm_sparse_triangular_solve.argtypes = [pc, pc, ci, pint, C.POINTER(i32), pint]
    R.spasm_pivots_extract_structural.restype = ci
    R.spasm_pivots_extract_structural.argtypes = [pc, pint, pl, pint, C.POINTER(_RefOpts)]
    R.spasm_schur.restype = pc
    R.spasm_schur.argtypes = [pc, pint, ci, pl, C.c_double, pt, pint, pint]
    R.spasm_schur_dense.argtypes = [pc, pint, ci, pint, pl, C.c_void_p, ci, pint, pint]
    R.spasm_rref.restype = pc
    R.spasm_rref.argtypes = [pl, pint]
    R.spasm_prng_seed_simple.argtypes = [i64, C.c_uint64, C.c_uint32, C.c_void_p]
    R.spasm_prng_ZZp.restype = i32
    R.spasm_prng_ZZp.argtypes = [C.c_void_p]
    _ref = R
    return R


def ref_set_threads(k):
    """number of OpenMP threads the reference's parallel regions use."""
    C.CDLL("libgomp.so.1").omp_set_num_threads(int(k))


def ref_field(prime):
    F = _RefField()
    ref().spasm_field_init(prime, C.byref(F))
    return F


def _ref_from(ptr):
    s = ptr.contents
    n = s.n
    p = np.ctypeslib.as_array(s.p, shape=(n + 1,)).copy()
    nnz = int(p[n])
    if nnz:
        j = np.ctypeslib.as_array(s.j, shape=(nnz,)).copy()
        x = np.ctypeslib.as_array(s.x, shape=(nnz,)).copy()
    else:
        j = np.zeros(0, np.int32)
        x = np.zeros(0, np.int32)
    return CSR(n, s.m, p, j, x, s.field.p)


def _ref_to(A, extra_rows=0, extra_nz=0):
    R = ref()
    ptr = R.spasm_csr_alloc(A.n + extra_rows, A.m, max(A.nnz + extra_nz, 1), A.prime, True)
    s = ptr.contents
    C.memmove(s.p, A.p.ctypes.data, 8 * (A.n + 1))
    if A.nnz:
        C.memmove(s.j, A.j.ctypes.data, 4 * A.nnz)
        C.memmove(s.x, A.x.ctypes.data, 4 * A.nnz)
    ptr.contents.n = A.n
    return ptr


def ref_compress(prime, n, m, ti, tj, tx):
    """spasm_triplet_alloc + spasm_add_entry + spasm_compress of the real reference."""
    R = ref()
    devnull = _silence()
    try:
        T = R.spasm_triplet_alloc(n, m, max(len(ti), 1), prime, True)
        for a, b, c in zip(ti.tolist(), tj.tolist(), tx.tolist()):
            R.spasm_add_entry(T, a, b, c)
        Cp = R.spasm_compress(T)
        out = _ref_from(Cp)
        R.spasm_csr_free(Cp)
        R.spasm_triplet_free(T)
    finally:
        _unsilence(devnull)
    return out


def ref_load(path, prime):
    """spasm_triplet_load + spasm_compress of the real reference (spasm_io.c:60-160: SMS and MatrixMarket files)."""
    R = ref()
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    R.spasm_triplet_load.restype = C.POINTER(_RefTriplet)
    R.spasm_triplet_load.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    f = libc.fopen(path.encode(), b"r")
    if not f:
        raise OSError("cannot open %s" % path)
    saved = _silence()
    try:
        T = R.spasm_triplet_load(f, prime, None)
        Cp = R.spasm_compress(T)
        out = _ref_from(Cp)
        R.spasm_csr_free(Cp)
        R.spasm_triplet_free(T)
    finally:
        _unsilence(saved)
        libc.fclose(f)
    return out


def _silence():
    """the reference chats on stderr; park fd 2 on /dev/null around its calls."""
    import sys
    sys.stderr.flush()
    saved = os.dup(2)
    nul = os.open(os.devnull, os.O_WRONLY)
    os.dup2(nul, 2)
    os.close(nul)
    return saved


def _unsilence(saved):
    os.dup2(saved, 2)
    os.close(saved)


def _ref_lu(F, n_rows):
    """a struct spasm_lu (owned pieces returned for cleanup) initialised from a Fact."""
    R = ref()
    U = F.U
    up = _ref_to(U, extra_rows=n_rows, extra_nz=16)
    qinv = np.ascontiguousarray(F.qinv, np.int32).copy()
    lu = _RefLu()
    lu.r = U.n
    lu.complete = False
    lu.L = None
    lu.U = up
    lu.qinv = _ip(qinv)
    lu.p = None
    lu.Ltmp = None
    return lu, up, qinv


def ref_pivots_extract_structural(A, F, greedy=True):
    R = ref()
    a = _ref_to(A)
    lu, up, qinv = _ref_lu(F, A.n)
    # U must have room for every row of A
    opts = _RefOpts()
    opts.enable_greedy_pivot_search = bool(greedy)
    p = np.zeros(max(A.n, 1), np.int32)
    saved = _silence()
    try:
        npiv = R.spasm_pivots_extract_structural(a, None, C.byref(lu), _ip(p), C.byref(opts))
    finally:
        _unsilence(saved)
    Uo = _ref_from(lu.U)
    out = Fact(Uo, qinv.copy())
    R.spasm_csr_free(lu.U)
    R.spasm_csr_free(a)
    return npiv, p[:A.n], out


def ref_schur(A, p, F, threads=1):
    """the reference's spasm_schur (est_density=-1, no L).  returns (S, p_out)."""
    R = ref()
    ref_set_threads(threads)
    a = _ref_to(A)
    lu, up, qinv = _ref_lu(F, 0)
    p = np.ascontiguousarray(p, np.int32)
    n = len(p)
    p_out = np.zeros(max(n, 1), np.int32)
    saved = _silence()
    try:
        s = R.spasm_schur(a, _ip(p), n, C.byref(lu), 1.0 if n == 0 else -1.0, None, None, _ip(p_out))
    finally:
        _unsilence(saved)
    S = _ref_from(s)
    R.spasm_csr_free(s)
    R.spasm_csr_free(up)
    R.spasm_csr_free(a)
    return S, p_out[:n]


def ref_schur_dense(A, p, F):
    """the reference's spasm_schur_dense with the SPASM_I64 datatype.  (S, q, p_out)."""
    R = ref()
    a = _ref_to(A)
    lu, up, qinv = _ref_lu(F, 0)
    p = np.ascontiguousarray(p, np.int32)
    n = len(p)
    Sm = A.m - F.U.n
    Sflat = np.zeros(max(n * Sm, 1), np.int64)
    q = np.zeros(max(Sm, 1), np.int32)
    p_out = np.zeros(max(n, 1), np.int32)
    saved = _silence()
    try:
        R.spasm_schur_dense(a, _ip(p), n, None, C.byref(lu), Sflat.ctypes.data_as(C.c_void_p),
                            2, _ip(q), _ip(p_out))       # 2 == SPASM_I64 (spasm.h:139)
    finally:
        _unsilence(saved)
    R.spasm_csr_free(up)
    R.spasm_csr_free(a)
    return Sflat[:n * Sm].reshape(n, Sm), q[:Sm], p_out[:n]


def ref_solve_row(U, qinv, B, k):
    R = ref()
    u = _ref_to(U)
    b = _ref_to(B)
    m = U.m
    xj = np.zeros(3 * max(m, 1), np.int32)
    x = np.zeros(max(m, 1), np.int32)
    q = np.ascontiguousarray(qinv, np.int32)
    top = R.spasm_sparse_triangular_solve(u, b, k, _ip(xj), x.ctypes.data_as(C.POINTER(C.c_int32)), _ip(q))
    R.spasm_csr_free(u)
    R.spasm_csr_free(b)
    return xj[top:m].copy(), x[:m]


def ref_rref(F):
    R = ref()
    lu, up, qinv = _ref_lu(F, 0)
    Rq = np.zeros(max(F.U.m, 1), np.int32)
    ref_set_threads(1)
    saved = _silence()
    try:
        r = R.spasm_rref(C.byref(lu), _ip(Rq))
    finally:
        _unsilence(saved)
    out = _ref_from(r)
    R.spasm_csr_free(r)
    R.spasm_csr_free(up)
    return out, Rq[:F.U.m]
