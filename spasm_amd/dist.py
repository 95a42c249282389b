"""Row-batch sharding of the Schur complement over the GPUs of one node.

Rows of a Schur complement are independent (spasm_schur.c:88-171 hands them to
OpenMP threads one by one), so every rank reduces a contiguous slice of the
row list against its own replica of (A, factor) and the slices are reassembled
with an all-gatherv: sizes first, then the padded payload through
all_gather_into_tensor (RCCL over xGMI on GPUs; gloo in the CPU tests).
"""
import ctypes as C

import numpy as np


class Comm:
    """RCCL communicator of the library itself (spasm_hip_comm, spasm_amd/csrc/dist_api.hip): one process per GPU.

    The id is drawn by rank 0 and handed to the others through `exchange(bytes or None) -> bytes` (bench.py broadcasts
    it with torch.distributed; any channel will do).  world == 1 needs no exchange."""

    def __init__(self, rank, world, exchange=None):
        from ._lib import lib, require_gpu
        require_gpu("Comm")
        L = lib()
        nbytes = L.spasm_hip_comm_id_bytes()
        buf = (C.c_ubyte * nbytes)()
        if rank == 0:
            L.spasm_hip_comm_new_id(buf)
        if world > 1:
            raw = exchange(bytes(buf) if rank == 0 else None)
            buf = (C.c_ubyte * nbytes).from_buffer_copy(raw)
        self.rank, self.world = rank, world
        self._h = L.spasm_hip_comm_create(buf, rank, world)

    def install(self):
        """the host-pointer entry points (schur, echelonize, ...) shard over this communicator from now on"""
        from ._lib import lib
        lib().spasm_hip_set_comm(self._h)

    def uninstall(self):
        from ._lib import lib
        lib().spasm_hip_set_comm(None)

    def allgatherv(self, W, m, prime, stream=0):
        """all-gatherv (on the devices) of what the last dschur left in every rank's workspace W -> DeviceCsr"""
        import torch
        from ._lib import lib
        from .device import DeviceCsr
        L = lib()
        rows, nnz = C.c_int(0), C.c_int64(0)
        L.spasm_hip_dschur_allgatherv(self._h, W._h, None, None, None, -1, C.byref(rows), C.byref(nnz), stream)
        dev = torch.device("cuda", torch.cuda.current_device())
        Sp = torch.empty(rows.value + 1, dtype=torch.int64, device=dev)
        Sj = torch.empty(max(nnz.value, 1), dtype=torch.int32, device=dev)
        Sx = torch.empty(max(nnz.value, 1), dtype=torch.int32, device=dev)
        rc = L.spasm_hip_dschur_allgatherv(self._h, W._h, Sp.data_ptr(), Sj.data_ptr(), Sx.data_ptr(), nnz.value, None, None, stream)
        if rc != 0:
            raise RuntimeError("spasm_hip_dschur_allgatherv failed")
        return DeviceCsr(rows.value, m, Sp, Sj, Sx, prime)

    def close(self):
        if self._h:
            from ._lib import lib
            lib().spasm_hip_comm_destroy(self._h)
            self._h = None


def echelonize_dist(A, comm, opts=None):
    """spasm_hip_echelonize_dist: echelonize with every round's Schur complement sharded over the ranks of comm."""
    import ctypes as C
    from ._lib import lib
    from .matrix import view_csr, copy_csr, Fact
    L = lib()
    a = view_csr(A)
    lu = L.spasm_hip_echelonize_dist(C.byref(a), C.byref(opts) if opts is not None else None, comm._h)
    s = lu.contents
    U = copy_csr(s.U)
    qinv = np.ctypeslib.as_array(s.qinv, shape=(max(A.m, 1),))[:A.m].copy()
    L.spasm_hip_lu_free(lu)
    return Fact(U, qinv)


def shard_bounds(n, rank, world):
    """[lo, hi) of the contiguous slice of n rows owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_rows(rows, rank, world):
    lo, hi = shard_bounds(len(rows), rank, world)
    return rows[lo:hi]


def allgatherv_csr(S, dist, group=None):
    """S: DeviceCsr-like (n, m, p, j, x tensors, prime) holding this rank's rows.
    Returns the concatenation over ranks, in rank order, as the same kind of object."""
    import torch
    from .device import DeviceCsr
    world = dist.get_world_size(group)
    dev = S.p.device
    nnz = int(S.p[S.n].item()) if S.n > 0 else 0
    mine = torch.tensor([S.n, nnz], dtype=torch.int64, device=dev)
    sizes = torch.empty(2 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, mine, group=group)
    sizes = sizes.view(world, 2).cpu()
    max_n, max_nz = int(sizes[:, 0].max()), int(sizes[:, 1].max())
    # row lengths (int64) and the two entry arrays, padded to the largest slice
    lens = torch.zeros(max(max_n, 1), dtype=torch.int64, device=dev)
    if S.n:
        lens[:S.n] = S.p[1:S.n + 1] - S.p[:S.n]
    jx = torch.zeros(2, max(max_nz, 1), dtype=torch.int32, device=dev)
    if nnz:
        jx[0, :nnz] = S.j[:nnz]
        jx[1, :nnz] = S.x[:nnz]
    all_lens = torch.empty(world * lens.numel(), dtype=torch.int64, device=dev)
    all_jx = torch.empty(world * jx.numel(), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(all_lens, lens, group=group)
    dist.all_gather_into_tensor(all_jx, jx.view(-1), group=group)
    all_lens = all_lens.view(world, -1)
    all_jx = all_jx.view(world, 2, -1)
    n_tot = int(sizes[:, 0].sum())
    keep_lens = torch.cat([all_lens[r, :int(sizes[r, 0])] for r in range(world)]) if n_tot else lens[:0]
    p = torch.zeros(n_tot + 1, dtype=torch.int64, device=dev)
    if n_tot:
        p[1:] = torch.cumsum(keep_lens, 0)
    j = torch.cat([all_jx[r, 0, :int(sizes[r, 1])] for r in range(world)])
    x = torch.cat([all_jx[r, 1, :int(sizes[r, 1])] for r in range(world)])
    if j.numel() == 0:
        j = torch.zeros(1, dtype=torch.int32, device=dev)
        x = torch.zeros(1, dtype=torch.int32, device=dev)
    return DeviceCsr(n_tot, S.m, p, j, x, S.prime)


def sharded_schur(A, rows, F, W, dist, reduce_rows, stream=0, group=None):
    """Schur complement of `rows` computed by all ranks of `group`.

    reduce_rows(my_rows) -> DeviceCsr of this rank's slice; the product passes
    the HIP path (spasm_amd.device.dschur), the CPU tests pass a stand-in so the
    sharding + all-gatherv logic can run under gloo without a GPU.
    """
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = shard_rows(rows, rank, world)
    S = reduce_rows(mine)
    return allgatherv_csr(S, dist, group=group)


class CXfer(C.Structure):          # spasm_hip_xfer
    _fields_ = [("kind", C.c_int), ("peer", C.c_int), ("array", C.c_int), ("src", C.c_int64), ("dst", C.c_int64), ("count", C.c_int64)]


XFER_SEND, XFER_RECV, XFER_COPY = 0, 1, 2


def allgatherv_plan(world, me, sizes):
    """spasm_hip_allgatherv_plan: the steps rank `me` issues in spasm_hip_dschur_allgatherv, in order, as tuples
    (kind, peer, array, src, dst, count); plus the row / entry offsets of the slices.  Pure host function (no GPU)."""
    from ._lib import lib
    L = lib()
    flat = (C.c_int64 * (2 * world))(*[int(v) for pair in sizes for v in pair])
    rb, zb = (C.c_int64 * (world + 1))(), (C.c_int64 * (world + 1))()
    n = L.spasm_hip_allgatherv_plan(world, me, flat, None, 0, rb, zb)
    buf = (CXfer * max(n, 1))()
    L.spasm_hip_allgatherv_plan(world, me, flat, C.cast(buf, C.c_void_p), n, None, None)
    return [(x.kind, x.peer, x.array, x.src, x.dst, x.count) for x in buf[:n]], list(rb), list(zb)


def column_slab(A, F, part, parts):
    """spasm_hip_column_slab: (A', F', cols) -- the problem of the rank that owns range `part` of `parts` of the non-pivotal
    columns; cols[c'] = original column of column c' of A'.  Pure host function (no GPU)."""
    from ._lib import lib
    from .matrix import CCsr, CLu, view_csr, copy_csr, Fact
    L = lib()
    a = view_csr(A)
    u = view_csr(F.U)
    qinv = np.ascontiguousarray(F.qinv, np.int32)
    lu = CLu()
    lu.r = F.U.n
    lu.complete = False
    lu.L = None
    lu.U = C.pointer(u)
    lu.qinv = qinv.ctypes.data_as(C.POINTER(C.c_int))
    lu.p = None
    lu.Ltmp = None
    cols = np.zeros(max(A.m, 1), np.int32)
    pa, pf = C.POINTER(CCsr)(), C.POINTER(CLu)()
    mm = L.spasm_hip_column_slab(C.byref(a), C.byref(lu), part, parts, C.byref(pa), C.byref(pf), cols.ctypes.data_as(C.POINTER(C.c_int)))
    As = copy_csr(pa)
    Us = copy_csr(pf.contents.U)
    q2 = np.ctypeslib.as_array(pf.contents.qinv, shape=(max(mm, 1),))[:mm].copy()
    L.spasm_hip_csr_free(pa)
    L.spasm_hip_lu_free(pf)
    return As, Fact(Us, q2), cols[:mm].copy()


def stitch_column_slabs(parts, cols_of, n, m, prime):
    """rows of the full Schur complement from those of its column slabs: parts[k] = Csr-like (p, j, x) of slab k on all n
    rows, cols_of[k] = its column map.  Slabs are ranges of increasing columns, so a row is the concatenation of its
    pieces in slab order.  Host arrays (numpy): this is what a rank does once it has been sent the other slabs."""
    from .matrix import Csr
    lens = np.zeros(n, np.int64)
    for S in parts:
        lens += np.diff(S.p)
    p = np.zeros(n + 1, np.int64)
    np.cumsum(lens, out=p[1:])
    j = np.zeros(int(p[n]), np.int32)
    x = np.zeros(int(p[n]), np.int32)
    at = p[:-1].copy()
    for S, cols in zip(parts, cols_of):
        ln = np.diff(S.p)
        # destination of every entry of the slab: start of its row + position inside the piece
        row_of = np.repeat(np.arange(n), ln)
        inside = np.arange(int(S.p[n])) - np.repeat(S.p[:-1], ln)
        dst = at[row_of] + inside
        j[dst] = cols[S.j[:int(S.p[n])]]
        x[dst] = S.x[:int(S.p[n])]
        at += ln
    return Csr(n, m, p, j, x, prime)
