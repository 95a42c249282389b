"""world_size-2 gloo test of the row sharding + all-gatherv reassembly (no GPU).

The per-shard compute is a stand-in (the oracle) injected through
sharded_schur's reduce_rows hook: what is under test is the N > 1 plumbing.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, matrix_path


def _worker(rank, world, port, name, prime, result_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from spasm_amd.device import DeviceCsr
    from spasm_amd.dist import sharded_schur, shard_bounds
    A = orc.load_sms(matrix_path(name), prime)
    npiv, perm, F = orc.pivots_extract_structural(A, orc.empty_fact(A.n, A.m, prime))
    rows = perm[npiv:]

    def reduce_rows(mine):
        S, _, _ = orc.schur(A, mine, F)
        return DeviceCsr(S.n, S.m, torch.from_numpy(S.p), torch.from_numpy(S.j if S.nnz else np.zeros(1, np.int32)),
                         torch.from_numpy(S.x if S.nnz else np.zeros(1, np.int32)), prime)

    full = sharded_schur(None, rows, None, None, dist, reduce_rows)
    want, _, _ = orc.schur(A, rows, F)
    nnz = int(full.p[full.n])
    ok = (full.n == want.n and np.array_equal(full.p.numpy(), want.p)
          and np.array_equal(full.j.numpy()[:nnz], want.j) and np.array_equal(full.x.numpy()[:nnz], want.x))
    lo, hi = shard_bounds(len(rows), rank, world)
    ok = ok and (hi - lo) in (len(rows) // world, len(rows) // world + 1)
    open(os.path.join(result_dir, "rank%d" % rank), "w").write("ok" if ok else "FAIL")
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("mat364.sms", 2), ("medium.sms", 2), ("small.sms", 3), ("void.sms", 2)])
def test_sharded_schur_gloo(tmp_path, name, world):
    port = 29500 + (os.getpid() + hash(name)) % 2000
    mp.spawn(_worker, args=(world, port, name, 42013, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d" % r)).read() == "ok"


def test_shard_bounds_cover():
    from spasm_amd.dist import shard_bounds
    for n in (0, 1, 7, 64, 1001):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))


# --------------------------------------------------------------------------
# the exchange plan of the RCCL all-gatherv (spasm_hip_allgatherv_plan: what dist_api.hip executes), for worlds 2 .. 8
# --------------------------------------------------------------------------
@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 6, 7, 8])
def test_allgatherv_plan_is_consistent_between_every_pair_of_ranks(world):
    """RCCL matches the k-th send of a to b with the k-th receive b posts for a (same group): for every ordered pair the
    sequence of (array, count) sent must equal the sequence received; the receives of a rank plus its local copy must tile
    the gathered arrays exactly; nothing of size zero is posted (by either side)."""
    from spasm_amd.dist import allgatherv_plan, XFER_SEND, XFER_RECV, XFER_COPY
    rng = np.random.default_rng(world)
    for trial in range(12):
        sizes = [(int(rng.integers(0, 50)), int(rng.integers(0, 5000))) for _ in range(world)]
        if trial == 0:
            sizes = [(0, 0)] * world                       # nothing anywhere
        if trial == 1:
            sizes = [(0, 0) if r % 2 else (7, 0) for r in range(world)]        # rows without entries, empty ranks
        sizes = [(n, z if n > 0 else 0) for n, z in sizes]
        plans = [allgatherv_plan(world, me, sizes) for me in range(world)]
        total_rows, total_nz = sum(s[0] for s in sizes), sum(s[1] for s in sizes)
        for me, (steps, rb, zb) in enumerate(plans):
            assert rb[-1] == total_rows and zb[-1] == total_nz
            assert all(rb[r + 1] - rb[r] == sizes[r][0] and zb[r + 1] - zb[r] == sizes[r][1] for r in range(world))
            assert all(count > 0 for (_, _, _, _, _, count) in steps)
            # the gathered arrays are written exactly once
            for array, total in ((0, total_rows), (1, total_nz), (2, total_nz)):
                covered = np.zeros(total, np.int32)
                for kind, peer, arr, src, dst, count in steps:
                    if arr == array and kind in (XFER_RECV, XFER_COPY):
                        covered[dst:dst + count] += 1
                        base = (rb if array == 0 else zb)[peer]
                        assert dst == base and count == sizes[peer][0 if array == 0 else 1]
                assert np.all(covered == 1)
            # sends read this rank's own slice from its start, whole
            for kind, peer, arr, src, dst, count in steps:
                if kind == XFER_SEND:
                    assert src == 0 and count == sizes[me][0 if arr == 0 else 1] and peer != me
        for a in range(world):
            for b in range(world):
                if a == b:
                    continue
                sent = [(arr, count) for kind, peer, arr, _, _, count in plans[a][0] if kind == XFER_SEND and peer == b]
                received = [(arr, count) for kind, peer, arr, _, _, count in plans[b][0] if kind == XFER_RECV and peer == a]
                assert sent == received, (world, a, b, sent, received)


# --------------------------------------------------------------------------
# the column split (spasm_hip_column_slab): the Schur complement of a slab problem is the slab of the Schur complement
# --------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["mat364.sms", "medium.sms", "rectangular_h.sms", "singular.sms", "trefethen_500.sms", "void.sms"])
@pytest.mark.parametrize("parts", [1, 2, 3, 8])
def test_column_slabs_stitch_to_the_full_schur_complement(oracle, name, parts):
    """every rank works on (A, U) with the non-pivotal columns of the other ranks deleted; mapped back and concatenated, the
    rows are those of the reference algorithm on the whole matrix (computed by the oracle on both sides: the function under
    test is the host-side split, spasm_hip_column_slab + stitch_column_slabs)."""
    import spasm_amd
    from spasm_amd.dist import column_slab, stitch_column_slabs
    p = 42013
    A = oracle.load_sms(matrix_path(name), p)
    npiv, perm, F = oracle.pivots_extract_structural(A, oracle.empty_fact(A.n, A.m, p))
    rows = perm[npiv:]
    want, _, _ = oracle.schur(A, rows, F)
    Ap = spasm_amd.Csr(A.n, A.m, A.p, A.j, A.x, p)
    Fp = spasm_amd.Fact(spasm_amd.Csr(F.U.n, F.U.m, F.U.p, F.U.j, F.U.x, p), F.qinv)
    pieces, maps, widths = [], [], []
    for k in range(parts):
        As, Fs, cols = column_slab(Ap, Fp, k, parts)
        assert As.n == A.n and As.m == len(cols) == Fs.U.m and Fs.U.n == F.U.n
        assert np.all(np.diff(cols) > 0)
        widths.append(len(cols) - F.U.n)
        Ao = oracle.CSR(As.n, As.m, As.p, As.j, As.x, p)
        Fo = oracle.Fact(oracle.CSR(Fs.U.n, Fs.U.m, Fs.U.p, Fs.U.j, Fs.U.x, p), Fs.qinv)
        S, _, _ = oracle.schur(Ao, rows, Fo)
        # (the library emits rows sorted by column; the oracle, like the reference, in the order of its elimination)
        for i in range(S.n):
            lo, hi = S.p[i], S.p[i + 1]
            o = np.argsort(S.j[lo:hi])
            S.j[lo:hi] = S.j[lo:hi][o]
            S.x[lo:hi] = S.x[lo:hi][o]
        pieces.append(S)
        maps.append(cols)
    assert sum(widths) == A.m - F.U.n and max(widths) - min(widths) <= 1
    full = stitch_column_slabs(pieces, maps, len(rows), A.m, p)
    assert full.nnz == want.nnz
    for k in range(len(rows)):
        wj, wx = want.row(k)
        o = np.argsort(wj)
        lo, hi = full.p[k], full.p[k + 1]
        assert np.array_equal(full.j[lo:hi], wj[o]) and np.array_equal(full.x[lo:hi], wx[o])


# --------------------------------------------------------------------------
# the column split under gloo: every rank its slab, only the row lengths are exchanged
# --------------------------------------------------------------------------
def _column_worker(rank, world, port, name, prime, result_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    import spasm_amd
    from spasm_amd.dist import column_slab
    A = orc.load_sms(matrix_path(name), prime)
    npiv, perm, F = orc.pivots_extract_structural(A, orc.empty_fact(A.n, A.m, prime))
    rows = perm[npiv:]
    Ap = spasm_amd.Csr(A.n, A.m, A.p, A.j, A.x, prime)
    Fp = spasm_amd.Fact(spasm_amd.Csr(F.U.n, F.U.m, F.U.p, F.U.j, F.U.x, prime), F.qinv)
    As, Fs, cols = column_slab(Ap, Fp, rank, world)
    S, _, _ = orc.schur(orc.CSR(As.n, As.m, As.p, As.j, As.x, prime), rows,
                        orc.Fact(orc.CSR(Fs.U.n, Fs.U.m, Fs.U.p, Fs.U.j, Fs.U.x, prime), Fs.qinv))
    # what bench.py --split columns exchanges: the lengths of this rank's pieces of every row
    mine = torch.from_numpy(np.diff(S.p).astype(np.int64)) if len(rows) else torch.zeros(0, dtype=torch.int64)
    everybody = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(everybody, mine)
    lengths = torch.stack(everybody).sum(0).numpy() if len(rows) else np.zeros(0, np.int64)
    want, _, _ = orc.schur(A, rows, F)
    ok = np.array_equal(lengths, np.diff(want.p))
    # and the pieces themselves are the entries of the whole rows on this rank's columns
    lo_col, hi_col = (int(cols[F.U.n:].min()), int(cols[F.U.n:].max())) if len(cols) > F.U.n else (0, -1)
    mycols = set(int(c) for c in cols)
    for k in range(len(rows)):
        wj, wx = want.row(k)
        keep = np.array([int(c) in mycols for c in wj], bool)
        gj, gx = S.row(k)
        o1, o2 = np.argsort(cols[gj]), np.argsort(wj[keep])
        ok = ok and np.array_equal(cols[gj][o1], wj[keep][o2]) and np.array_equal(gx[o1], wx[keep][o2])
    open(os.path.join(result_dir, "rank%d" % rank), "w").write("ok" if ok else "FAIL")
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("mat364.sms", 2), ("medium.sms", 3), ("rectangular_h.sms", 2)])
def test_column_split_gloo(tmp_path, name, world):
    port = 31500 + (os.getpid() + hash(name)) % 2000
    mp.spawn(_column_worker, args=(world, port, name, 42013, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d" % r)).read() == "ok"


# --------------------------------------------------------------------------
# round 6: the column slabs stay slabs between two rounds -- what the driver reads of them is column-separable
# --------------------------------------------------------------------------
def _slab_finish_worker(rank, world, port, name, prime, result_dir):
    """what schur_api.hip / dense_api.hip do with a Schur complement kept as column slabs, with the oracle as the compute and gloo as
    the collectives: (1) row lengths = sum over the ranks, (2) leftmost entry of every row = minimum over the ranks (the census of
    the next round's Faugere-Lachartre step), (3) random combinations of ALL rows, formed by every rank on its own columns and
    summed over the ranks (the slabs are disjoint: exact), equal the combinations of the whole rows."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    import spasm_amd
    from spasm_amd.dist import column_slab
    A = orc.load_sms(matrix_path(name), prime)
    npiv, perm, F = orc.pivots_extract_structural(A, orc.empty_fact(A.n, A.m, prime))
    rows = perm[npiv:]
    n = len(rows)
    Ap = spasm_amd.Csr(A.n, A.m, A.p, A.j, A.x, prime)
    Fp = spasm_amd.Fact(spasm_amd.Csr(F.U.n, F.U.m, F.U.p, F.U.j, F.U.x, prime), F.qinv)
    As, Fs, cols = column_slab(Ap, Fp, rank, world)
    S, _, _ = orc.schur(orc.CSR(As.n, As.m, As.p, As.j, As.x, prime), rows,
                        orc.Fact(orc.CSR(Fs.U.n, Fs.U.m, Fs.U.p, Fs.U.j, Fs.U.x, prime), Fs.qinv))
    want, _, _ = orc.schur(A, rows, F)
    ok = True
    # (1) lengths
    lens = torch.from_numpy(np.diff(S.p).astype(np.int32))
    if n:
        dist.all_reduce(lens, op=dist.ReduceOp.SUM)
    ok = ok and np.array_equal(lens.numpy(), np.diff(want.p))
    # (2) leftmost entries
    left = np.full(n, 0x7FFFFFFF, np.int32)
    for k in range(n):
        gj, _ = S.row(k)
        if len(gj):
            left[k] = int(cols[gj].min())
    left = torch.from_numpy(left)
    if n:
        dist.all_reduce(left, op=dist.ReduceOp.MIN)
    want_left = np.array([int(want.row(k)[0].min()) if len(want.row(k)[0]) else 0x7FFFFFFF for k in range(n)], np.int64)
    ok = ok and np.array_equal(left.numpy().astype(np.int64), want_left)
    # (3) combinations of all rows, on every rank's own columns, summed: the same draws on every rank (keyed by the problem)
    rng = np.random.default_rng(n * 1000003 + A.m)
    N = 5
    Cf = rng.integers(0, prime, size=(N, max(n, 1)), dtype=np.int64)[:, :n]
    mine = np.zeros((N, A.m), np.int64)
    for k in range(n):
        gj, gx = S.row(k)
        if len(gj):
            mine[:, cols[gj]] = (mine[:, cols[gj]] + Cf[:, [k]] * (np.asarray(gx, np.int64) % prime)[None, :]) % prime
    total = torch.from_numpy(mine.astype(np.int64))
    dist.all_reduce(total, op=dist.ReduceOp.SUM)          # (disjoint supports: at most one rank holds a non-zero word anywhere)
    whole = np.zeros((N, A.m), np.int64)
    for k in range(n):
        wj, wx = want.row(k)
        if len(wj):
            whole[:, wj] = (whole[:, wj] + Cf[:, [k]] * (np.asarray(wx, np.int64) % prime)[None, :]) % prime
    ok = ok and np.array_equal(total.numpy(), whole)
    open(os.path.join(result_dir, "rank%d" % rank), "w").write("ok" if ok else "FAIL")
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("mat364.sms", 2), ("medium.sms", 3), ("rectangular_h.sms", 2)])
def test_column_slabs_kept_through_the_census_and_the_finish_gloo(tmp_path, name, world):
    port = 33500 + (os.getpid() + hash(name)) % 2000
    mp.spawn(_slab_finish_worker, args=(world, port, name, 42013, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d" % r)).read() == "ok"
