"""Device-resident entry points (spasm_hip_d*): matrices stay in HBM as torch tensors.

torch is used only as the owner of device memory and streams; every kernel is
launched by libspasm_hip.so on the stream handed over as a raw hipStream_t.
"""
import ctypes as C

import numpy as np

from ._lib import lib, require_gpu
from .matrix import Csr, CDcsr, CSchurStats, view_csr


def _torch():
    import torch
    return torch


class DeviceCsr:
    """CSR matrix in HBM, layout of spasm_hip_dcsr (int64 p, int32 j, balanced int32 x)."""

    def __init__(self, n, m, p, j, x, prime):
        self.n, self.m, self.prime = int(n), int(m), int(prime)
        self.p, self.j, self.x = p, j, x

    @classmethod
    def from_host(cls, A, device="cuda:0"):
        torch = _torch()
        return cls(A.n, A.m,
                   torch.from_numpy(A.p).to(device),
                   torch.from_numpy(A.j if A.nnz else np.zeros(1, np.int32)).to(device),
                   torch.from_numpy(A.x if A.nnz else np.zeros(1, np.int32)).to(device), A.prime)

    @property
    def nnz(self):
        return int(self.p[self.n].item())

    def to_host(self):
        nnz = self.nnz
        return Csr(self.n, self.m, self.p.cpu().numpy(), self.j[:nnz].cpu().numpy(), self.x[:nnz].cpu().numpy(),
                   self.prime)

    def cstruct(self, nnz=None):
        s = CDcsr()
        s.n, s.m = self.n, self.m
        s.nnz = self.nnz if nnz is None else nnz
        s.p, s.j, s.x = self.p.data_ptr(), self.j.data_ptr(), self.x.data_ptr()
        return s


class DeviceFact:
    """device image of (U, qinv): level-sorted, relabelled, Montgomery-form rows (spasm_hip_dfact)."""

    def __init__(self, F, stream=0):
        require_gpu("DeviceFact")
        L = lib()
        u = view_csr(F.U)
        q = np.ascontiguousarray(F.qinv, np.int32)
        self._h = L.spasm_hip_dfact_create(C.byref(u), q.ctypes.data_as(C.POINTER(C.c_int)), stream)
        self.rank = L.spasm_hip_dfact_rank(self._h)
        self.levels = L.spasm_hip_dfact_levels(self._h)
        self.nnz = L.spasm_hip_dfact_nnz(self._h)
        self.m = F.U.m
        self.prime = F.U.prime

    def hint_density(self, density):
        """expected density of the Schur complements of this factor (dense ones go through the back-substituted image)"""
        lib().spasm_hip_dfact_hint_density(self._h, float(density))

    def hint_eliminations(self, per_row):
        """(row, pivot) eliminations a reduced row takes on the row-by-row path (the other input of the path choice)"""
        lib().spasm_hip_dfact_hint_eliminations(self._h, float(per_row))

    def forget(self):
        """drops derived state (the back-substituted rows): the next dschur pays for it again."""
        lib().spasm_hip_dfact_forget(self._h)

    def sparse_image_census(self, stream=0):
        """fill of R as the sparse image holds it, or None when the factor has no valid sparse image:
        {entries, tiles64 (occupied 64-column tiles), fragments, pairs ((row, segment) pairs)}"""
        out = (C.c_int64 * 4)()
        if not lib().spasm_hip_dfact_sparse_image_census(self._h, out, stream):
            return None
        return {"entries": out[0], "tiles64": out[1], "fragments": out[2], "pairs": out[3]}

    def close(self):
        if self._h:
            lib().spasm_hip_dfact_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SchurWorkspace:
    """scratch + row pool for dschur (spasm_hip_dwork)."""

    def __init__(self, max_rows, m, pool_entries):
        require_gpu("SchurWorkspace")
        self._h = lib().spasm_hip_dwork_create(int(max_rows), int(m), int(pool_entries))
        self.max_rows, self.m, self.pool_entries = int(max_rows), int(m), int(pool_entries)

    def close(self):
        if self._h:
            lib().spasm_hip_dwork_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dschur(A, rows, F, W, stream=0, fetch=True):
    """Schur complement of rows `rows` (int32 device tensor) of DeviceCsr A w.r.t. DeviceFact F.

    Returns (S or None, stats).  With fetch=True the result is copied out of the
    workspace into fresh tensors and returned as a DeviceCsr.
    """
    torch = _torch()
    L = lib()
    a = A.cstruct(nnz=-1)
    st = CSchurStats()
    nrows = int(rows.numel())
    rc = L.spasm_hip_dschur(C.byref(a), rows.data_ptr(), nrows, F._h, W._h, stream, C.byref(st))
    if rc != 0:
        return None, st
    if not fetch:
        return None, st
    dev = rows.device
    Sp = torch.empty(nrows + 1, dtype=torch.int64, device=dev)
    Sj = torch.empty(max(st.nnz, 1), dtype=torch.int32, device=dev)
    Sx = torch.empty(max(st.nnz, 1), dtype=torch.int32, device=dev)
    L.spasm_hip_dschur_fetch(W._h, Sp.data_ptr(), Sj.data_ptr(), Sx.data_ptr(), stream)
    return DeviceCsr(nrows, A.m, Sp, Sj, Sx, A.prime), st
