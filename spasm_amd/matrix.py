"""Host-side value classes mirroring struct spasm_csr / spasm_lu / echelonize_opts (spasm.h:37-108)."""
import ctypes as C

import numpy as np


class Csr:
    """n x m sparse matrix mod prime, compressed rows; x holds balanced int32 representatives."""

    def __init__(self, n, m, p, j, x, prime):
        self.n = int(n)
        self.m = int(m)
        self.p = np.ascontiguousarray(p, dtype=np.int64)
        self.j = np.ascontiguousarray(j, dtype=np.int32)
        self.x = np.ascontiguousarray(x, dtype=np.int32)
        self.prime = int(prime)
        if self.p.shape[0] != self.n + 1:
            raise ValueError("row pointer array must have n + 1 entries")

    @property
    def nnz(self):
        return int(self.p[self.n])

    def row(self, i):
        lo, hi = int(self.p[i]), int(self.p[i + 1])
        return self.j[lo:hi], self.x[lo:hi]


class Fact:
    """the part of struct spasm_lu the path uses: U (pivot first, unit) and qinv."""

    def __init__(self, U, qinv):
        self.U = U
        self.qinv = np.ascontiguousarray(qinv, dtype=np.int32)

    @property
    def r(self):
        return self.U.n


# ---- ctypes images of the reference structs (layout of include/spasm_hip.h) ----
class CField(C.Structure):
    _fields_ = [("p", C.c_int64), ("halfp", C.c_int64), ("mhalfp", C.c_int64), ("dinvp", C.c_double)]


class CCsr(C.Structure):
    _fields_ = [("nzmax", C.c_int64), ("n", C.c_int), ("m", C.c_int),
                ("p", C.POINTER(C.c_int64)), ("j", C.POINTER(C.c_int)),
                ("x", C.POINTER(C.c_int32)), ("field", CField)]


class CTriplet(C.Structure):
    _fields_ = [("nzmax", C.c_int64), ("nz", C.c_int64), ("n", C.c_int), ("m", C.c_int),
                ("i", C.POINTER(C.c_int)), ("j", C.POINTER(C.c_int)),
                ("x", C.POINTER(C.c_int32)), ("field", CField)]


class CLu(C.Structure):
    _fields_ = [("r", C.c_int), ("complete", C.c_bool), ("L", C.POINTER(CCsr)),
                ("U", C.POINTER(CCsr)), ("qinv", C.POINTER(C.c_int)),
                ("p", C.POINTER(C.c_int)), ("Ltmp", C.POINTER(CTriplet))]


class EchelonizeOpts(C.Structure):
    _fields_ = [("enable_greedy_pivot_search", C.c_bool), ("enable_tall_and_skinny", C.c_bool),
                ("enable_dense", C.c_bool), ("enable_GPLU", C.c_bool), ("L", C.c_bool),
                ("complete", C.c_bool), ("min_pivot_proportion", C.c_double), ("max_round", C.c_int),
                ("sparsity_threshold", C.c_double), ("dense_block_size", C.c_int),
                ("low_rank_ratio", C.c_double), ("tall_and_skinny_ratio", C.c_double),
                ("low_rank_start_weight", C.c_double)]


class CDcsr(C.Structure):          # spasm_hip_dcsr
    _fields_ = [("n", C.c_int), ("m", C.c_int), ("nnz", C.c_int64),
                ("p", C.c_void_p), ("j", C.c_void_p), ("x", C.c_void_p)]


class CSchurStats(C.Structure):    # spasm_hip_schur_stats
    _fields_ = [("nnz", C.c_int64), ("eliminations", C.c_int64), ("entries_streamed", C.c_int64),
                ("input_entries", C.c_int64), ("group_pivots", C.c_int64), ("rows", C.c_int), ("rows_lds", C.c_int),
                ("rows_lds_big", C.c_int), ("rows_dense", C.c_int), ("status", C.c_int), ("used_group_kernel", C.c_int), ("group_aborted", C.c_int),
                ("ms_eliminate", C.c_float), ("ms_group", C.c_float), ("ms_tier0", C.c_float), ("ms_tier1", C.c_float),
                ("ms_tier2", C.c_float), ("ms_finalize", C.c_float), ("ms_total", C.c_float),
                ("used_backsolve", C.c_int), ("backsolve_built", C.c_int), ("ms_backsolve", C.c_float), ("ms_apply", C.c_float),
                ("bytes_backsolve", C.c_int64), ("bytes_apply", C.c_int64), ("kernel", C.c_char * 64),
                ("kernel_other", C.c_char * 64), ("ms_expand", C.c_float), ("ms_pad", C.c_float), ("bytes_expand", C.c_int64),
                ("bytes_staged", C.c_int64), ("kernel_expand", C.c_char * 64),
                ("group_slots", C.c_int), ("group_slots_wanted", C.c_int), ("group_waves", C.c_int), ("group_slot_bytes", C.c_int64),
                ("used_sparse_image", C.c_int), ("sparse_image_built", C.c_int), ("ms_sparse_build", C.c_float), ("ms_sparse_apply", C.c_float),
                ("ms_sparse_gather", C.c_float), ("sparse_image_levels", C.c_int), ("sparse_image_launches", C.c_int), ("sparse_image_pad", C.c_int),
                ("sparse_image_nnz", C.c_int64), ("sparse_image_ops_build", C.c_int64), ("sparse_image_ops_apply", C.c_int64),
                ("bytes_sparse_build", C.c_int64), ("bytes_sparse_apply", C.c_int64), ("bytes_sparse_gather", C.c_int64)]


def field_of(prime):
    F = CField()
    F.p = prime
    F.halfp = prime // 2
    F.mhalfp = prime // 2 - prime + 1
    F.dinvp = 1.0 / prime
    return F


def view_csr(A):
    """a struct spasm_csr whose arrays are A's numpy buffers (no copy; keep A alive)."""
    s = CCsr()
    s.nzmax = max(A.nnz, len(A.j))
    s.n = A.n
    s.m = A.m
    s.p = A.p.ctypes.data_as(C.POINTER(C.c_int64))
    s.j = A.j.ctypes.data_as(C.POINTER(C.c_int))
    s.x = A.x.ctypes.data_as(C.POINTER(C.c_int32))
    s.field = field_of(A.prime)
    return s


def copy_csr(ptr):
    """numpy copy of a library-owned struct spasm_csr*."""
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
    return Csr(n, s.m, p, j, x, s.field.p)
