"""GPU parity: the HIP Schur complement against the oracle, through the C ABI.

Bit-exact: same rows (row k of S = reduction of row p[k]), same (column,
value) sets, values compared as integers mod p.
"""
import os

import numpy as np
import pytest

from conftest import ALL_MODULI, ALL_TEST_MATRICES, matrix_path

import spasm_amd

pytestmark = pytest.mark.gpu


def _as_product(A):
    return spasm_amd.Csr(A.n, A.m, A.p, A.j, A.x, A.prime)


def _fact(F):
    return spasm_amd.Fact(_as_product(F.U), F.qinv)


def _round0(oracle, name, p):
    A = oracle.load_sms(matrix_path(name), p)
    npiv, perm, F = oracle.pivots_extract_structural(A, oracle.empty_fact(A.n, A.m, p))
    return A, npiv, perm, F


def _check(oracle, S_gpu, p_out_gpu, S_want, p_out_want, sorted_rows=True):
    assert np.array_equal(p_out_gpu, p_out_want)
    G = oracle.CSR(S_gpu.n, S_gpu.m, S_gpu.p, S_gpu.j, S_gpu.x, S_gpu.prime)
    assert oracle.same_matrix(G, S_want)
    P = S_gpu.prime
    assert np.all(S_gpu.x <= P // 2) and np.all(S_gpu.x >= -(P // 2))          # balanced representatives
    if sorted_rows:
        for i in range(S_gpu.n):
            jj, _ = S_gpu.row(i)
            assert np.all(np.diff(jj) > 0)


@pytest.fixture(params=["auto", "rows"])
def schur_path(request, monkeypatch):
    """auto: the library chooses (factors with few non-pivotal columns go through the back-substituted image,
    backsolve.hip); rows: the row-by-row elimination kernels only."""
    if request.param == "rows":
        monkeypatch.setenv("SPASM_HIP_BACKSOLVE", "0")
    return request.param


@pytest.mark.parametrize("name", ALL_TEST_MATRICES)
@pytest.mark.parametrize("p", ALL_MODULI)
def test_schur_reference_matrices(oracle, name, p, schur_path):
    A, npiv, perm, F = _round0(oracle, name, p)
    rows = perm[npiv:]
    want, p_out_want, _ = oracle.schur(A, rows, F)
    S, p_out = spasm_amd.schur(_as_product(A), rows, _fact(F))
    _check(oracle, S, p_out, want, p_out_want)
    assert np.all(F.qinv[S.j] < 0)          # tests/schur.c:62-72


@pytest.mark.parametrize("tier", [1, 2])
@pytest.mark.parametrize("name", ["mat364.sms", "medium.sms", "trefethen_500.sms", "singular.sms", "rectangular_l.sms",
                                  "BIOMD0000000424.int.mpl.sms", "void.sms", "empty.sms"])
@pytest.mark.parametrize("p", [3, 42013, 4294967291])
def test_schur_large_table_and_dense_tiers(oracle, name, p, tier, monkeypatch):
    """same answers when every row is pushed through the large LDS table / the dense accumulator tier."""
    monkeypatch.setenv("SPASM_HIP_FORCE_TIER", str(tier))
    A, npiv, perm, F = _round0(oracle, name, p)
    rows = perm[npiv:]
    want, p_out_want, _ = oracle.schur(A, rows, F)
    S, p_out = spasm_amd.schur(_as_product(A), rows, _fact(F))
    _check(oracle, S, p_out, want, p_out_want)


@pytest.mark.parametrize("name", ["mat364.sms", "medium.sms", "m1.sms", "trefethen_500.sms", "l1.sms", "G2.sms"])
@pytest.mark.parametrize("p", [257, 42013, 189812507])
def test_schur_second_round(oracle, name, p, schur_path):
    """factor with rows from two rounds: old rows of U meet columns that became pivotal later."""
    A, npiv, perm, F = _round0(oracle, name, p)
    S1, p_out1, _ = oracle.schur(A, perm[npiv:], F)
    if S1.nnz == 0:
        return
    npiv2, perm2, F2 = oracle.pivots_extract_structural(S1, F)
    rows = perm2[npiv2:]
    want, p_out_want, _ = oracle.schur(S1, rows, F2, p_in=p_out1)
    S, p_out = spasm_amd.schur(_as_product(S1), rows, _fact(F2), p_in=p_out1)
    _check(oracle, S, p_out, want, p_out_want)


def _random_sparse(rng, n, m, per_row, p):
    ti = np.repeat(np.arange(n, dtype=np.int32), per_row)
    tj = rng.integers(0, m, size=n * per_row).astype(np.int32)
    tx = rng.integers(1, p, size=n * per_row).astype(np.int64)
    return ti, tj, tx


@pytest.mark.parametrize("p", [42013, 4294967291])
@pytest.mark.parametrize("shape", [(3000, 2000, 3), (1500, 4000, 6), (6000, 1200, 2)])
def test_schur_random_fill_in(oracle, shape, p, schur_path):
    """random matrices: heavy fill-in drives rows through every tier; full rows-vs-oracle equality."""
    n, m, per_row = shape
    rng = np.random.default_rng(n + m)
    ti, tj, tx = _random_sparse(rng, n, m, per_row, p)
    A = oracle.compress(p, n, m, ti, tj, tx)
    npiv, perm, F = oracle.pivots_extract_structural(A, oracle.empty_fact(A.n, A.m, p))
    rows = perm[npiv:]
    want, p_out_want, _ = oracle.schur(A, rows, F)
    S, p_out = spasm_amd.schur(_as_product(A), rows, _fact(F))
    _check(oracle, S, p_out, want, p_out_want)


def test_device_resident_path_and_stats(oracle):
    """spasm_hip_dschur on tensors resident in HBM; statistics agree with the oracle's work count."""
    import torch
    p = 42013
    A, npiv, perm, F = _round0(oracle, "mat364.sms", p)
    rows = perm[npiv:]
    want, _, _ = oracle.schur(A, rows, F)
    dA = spasm_amd.DeviceCsr.from_host(_as_product(A))
    dF = spasm_amd.DeviceFact(_fact(F))
    W = spasm_amd.SchurWorkspace(len(rows), A.m, 4 * want.nnz + (1 << 22))
    drows = torch.from_numpy(np.ascontiguousarray(rows, np.int32)).cuda()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        S, st = spasm_amd.dschur(dA, drows, dF, W, stream=stream.cuda_stream)
    assert st.status == 0 and st.nnz == want.nnz and st.rows == len(rows)
    assert st.rows_lds + st.rows_lds_big + st.rows_dense == len(rows)
    H = S.to_host()
    assert oracle.same_matrix(oracle.CSR(H.n, H.m, H.p, H.j, H.x, p), want)
    # a pool that is too small is reported, not silently truncated
    W2 = spasm_amd.SchurWorkspace(len(rows), A.m, 16)
    S2, st2 = spasm_amd.dschur(dA, drows, dF, W2)
    assert S2 is None and st2.status == 1


def test_empty_row_list(oracle):
    A, npiv, perm, F = _round0(oracle, "small.sms", 257)
    S, p_out = spasm_amd.schur(_as_product(A), np.zeros(0, np.int32), _fact(F))
    assert S.n == 0 and S.nnz == 0


@pytest.mark.parametrize("variant", ["push", "pull"])
@pytest.mark.parametrize("name", ["mat364.sms", "medium.sms", "trefethen_500.sms", "singular.sms", "rectangular_l.sms",
                                  "BIOMD0000000424.int.mpl.sms", "void.sms", "empty.sms", "m1.sms", "small.sms"])
@pytest.mark.parametrize("p", [3, 42013, 65537, 4294967291])
def test_schur_row_group_kernel(oracle, name, p, variant, monkeypatch):
    """the 64-rows-per-workgroup kernels (label-major accumulators) give the same matrix: push = schur_group_kernel (one
    no-return atomic per update), pull = schur_pull_kernel (symbolic sweep, then a left-looking numeric sweep with plain
    loads and one store per touched line: no atomics; SPASM_HIP_PULL=1)."""
    monkeypatch.setenv("SPASM_HIP_GROUP", "1")
    if variant == "pull":
        monkeypatch.setenv("SPASM_HIP_PULL", "1")
    A, npiv, perm, F = _round0(oracle, name, p)
    rows = perm[npiv:]
    want, p_out_want, _ = oracle.schur(A, rows, F)
    S, p_out = spasm_amd.schur(_as_product(A), rows, _fact(F))
    _check(oracle, S, p_out, want, p_out_want)


@pytest.mark.parametrize("touched_hbm", [False, True])
@pytest.mark.parametrize("waves", ["1", "2", "4"])
@pytest.mark.parametrize("name", ["mat364.sms", "trefethen_500.sms", "BIOMD0000000424.int.mpl.sms"])
@pytest.mark.parametrize("p", [257, 42013, 4294967291])
def test_schur_row_group_kernel_waves_per_group(oracle, name, p, waves, touched_hbm, monkeypatch):
    """every waves-per-group variant of the row-group kernel (32- and 64-bit sums; bits of the touched non-pivotal
    labels in LDS or in HBM) gives the same matrix."""
    monkeypatch.setenv("SPASM_HIP_GROUP", "1")
    monkeypatch.setenv("SPASM_HIP_GROUP_WAVES", waves)
    if touched_hbm:
        monkeypatch.setenv("SPASM_HIP_GROUP_TOUCHED_HBM", "1")
    A, npiv, perm, F = _round0(oracle, name, p)
    rows = perm[npiv:]
    want, p_out_want, _ = oracle.schur(A, rows, F)
    S, p_out = spasm_amd.schur(_as_product(A), rows, _fact(F))
    _check(oracle, S, p_out, want, p_out_want)


@pytest.mark.parametrize("variant", ["push", "pull"])
@pytest.mark.parametrize("p", [42013, 4294967291])
def test_schur_row_group_kernel_random(oracle, p, variant, monkeypatch):
    monkeypatch.setenv("SPASM_HIP_GROUP", "1")
    if variant == "pull":
        monkeypatch.setenv("SPASM_HIP_PULL", "1")
    n, m, per_row = 3000, 2000, 3
    rng = np.random.default_rng(99)
    ti, tj, tx = _random_sparse(rng, n, m, per_row, p)
    A = oracle.compress(p, n, m, ti, tj, tx)
    npiv, perm, F = oracle.pivots_extract_structural(A, oracle.empty_fact(A.n, A.m, p))
    rows = perm[npiv:]
    want, p_out_want, _ = oracle.schur(A, rows, F)
    S, p_out = spasm_amd.schur(_as_product(A), rows, _fact(F))
    _check(oracle, S, p_out, want, p_out_want)


@pytest.mark.parametrize("variant", ["push", "pull"])
@pytest.mark.parametrize("p", [42013, 4294967291])
@pytest.mark.parametrize("row_len", [5, 9, 68, 69, 150, 300])
def test_schur_row_group_kernel_long_pivot_rows(oracle, p, row_len, variant, monkeypatch):
    """pivot rows beyond the four-entry head: entries 4.. are fetched 64 per instruction (lane = entry), rows
    over 68 entries in several chunks.  Upper-trapezoidal pivot block with long rows + rows to reduce."""
    monkeypatch.setenv("SPASM_HIP_GROUP", "1")
    if variant == "pull":
        monkeypatch.setenv("SPASM_HIP_PULL", "1")
    rng = np.random.default_rng(row_len)
    npiv_rows, m, nred = 400, 900, 200
    ti, tj, tx = [], [], []
    for k in range(npiv_rows):          # row k: pivot on column k, then row_len - 1 entries to its right
        cols = np.concatenate(([k], k + 1 + rng.choice(m - k - 1, size=min(row_len - 1, m - k - 1), replace=False)))
        ti += [k] * len(cols); tj += list(cols); tx += list(rng.integers(1, p, size=len(cols)))
    for k in range(nred):               # rows to reduce: a few entries anywhere
        cols = rng.choice(m, size=6, replace=False)
        ti += [npiv_rows + k] * 6; tj += list(cols); tx += list(rng.integers(1, p, size=6))
    A = oracle.compress(p, npiv_rows + nred, m, np.array(ti, np.int32), np.array(tj, np.int32), np.array(tx, np.int64))
    npiv, perm, F = oracle.pivots_extract_structural(A, oracle.empty_fact(A.n, A.m, p))
    assert npiv >= npiv_rows // 2
    rows = perm[npiv:]
    want, p_out_want, _ = oracle.schur(A, rows, F)
    S, p_out = spasm_amd.schur(_as_product(A), rows, _fact(F))
    _check(oracle, S, p_out, want, p_out_want)


@pytest.mark.parametrize("regroup", ["1", "0"])
def test_row_group_kernel_gives_up_on_unrelated_rows(oracle, regroup, monkeypatch):
    """rows that are neighbours in the list but live in different diagonal blocks share nothing.  regroup=1 (default):
    the pivot graph has one connected component per block, the rows are grouped by component first and the row-group
    kernel finishes the batch; regroup=0: it must notice (lane efficiency), stop, and the per-row kernels finish the
    batch -- same matrix either way."""
    import torch
    monkeypatch.setenv("SPASM_HIP_GROUP_REGROUP", regroup)
    monkeypatch.setenv("SPASM_HIP_BACKSOLVE", "0")
    monkeypatch.setenv("SPASM_HIP_GROUP_WATCH_ROWS", "0")
    monkeypatch.setenv("SPASM_HIP_GROUP_MIN_PIVOTS", "20000")      # the batch is small: judge early
    p = 42013
    rng = np.random.default_rng(11)
    B, W, extra = 520, 32, 8
    ti, tj, tx = [], [], []
    row = 0
    for b in range(B):                      # W chain rows (structural pivots) + `extra` rows per block
        c0 = b * W
        for i in range(W):
            ti += [row, row]
            tj += [c0 + i, c0 + min(i + 1, W - 1)] if i + 1 < W else [c0 + i, c0 + i]
            tx += [int(rng.integers(1, p)), int(rng.integers(1, p))]
            row += 1
    n_chain = row
    blocks = rng.permutation(np.repeat(np.arange(B), extra))      # neighbours in the list: different blocks
    for b in blocks:
        cols = rng.choice(W, size=3, replace=False) + b * W
        for c in cols:
            ti.append(row)
            tj.append(int(c))
            tx.append(int(rng.integers(1, p)))
        row += 1
    A = oracle.compress(p, row, B * W, np.array(ti, np.int32), np.array(tj, np.int32), np.array(tx, np.int64))
    npiv, perm, F = oracle.pivots_extract_structural(A, oracle.empty_fact(A.n, A.m, p))
    rows = perm[npiv:]
    assert len(rows) >= 2048
    want, _, _ = oracle.schur(A, rows, F)
    dA = spasm_amd.DeviceCsr.from_host(_as_product(A))
    dF = spasm_amd.DeviceFact(_fact(F))
    Wk = spasm_amd.SchurWorkspace(len(rows), A.m, 4 * want.nnz + (1 << 22))
    drows = torch.from_numpy(np.ascontiguousarray(rows, np.int32)).cuda()
    S, st = spasm_amd.dschur(dA, drows, dF, Wk)
    assert st.status == 0 and st.used_group_kernel == 1 and st.group_aborted == (0 if regroup == "1" else 1)
    assert st.rows_lds + st.rows_lds_big + st.rows_dense == len(rows)
    # (rows this small end in the LDS tier when the per-row kernels get them)
    assert (st.rows_lds == 0) if regroup == "1" else (st.rows_lds > 0)
    H = S.to_host()
    assert oracle.same_matrix(oracle.CSR(H.n, H.m, H.p, H.j, H.x, p), want)


def _triplet_set(trip, p):
    return sorted(zip(trip[0].tolist(), trip[1].tolist(), (np.asarray(trip[2], np.int64) % p).tolist()))


@pytest.mark.parametrize("group", ["0", "1"])
@pytest.mark.parametrize("name", ["mat364.sms", "medium.sms", "m1.sms", "singular.sms", "small.sms", "G2.sms", "void.sms"])
@pytest.mark.parametrize("p", [257, 42013, 4294967291])
def test_schur_records_L(oracle, name, p, group, monkeypatch):
    """the elimination coefficients (L parameter of spasm_schur, spasm_schur.c:160-167): same triplets."""
    monkeypatch.setenv("SPASM_HIP_GROUP", group)
    A, npiv, perm, F = _round0(oracle, name, p)
    rows = perm[npiv:]
    p_in = np.arange(A.n, dtype=np.int32)[::-1].copy()            # a non-trivial row relabelling
    want, p_out_want, L_want = oracle.schur(A, rows, F, p_in=p_in, want_L=True)
    S, p_out, L_got = spasm_amd.schur(_as_product(A), rows, _fact(F), p_in=p_in, want_L=True)
    _check(oracle, S, p_out, want, p_out_want)
    assert _triplet_set(L_got, p) == _triplet_set(L_want, p)
    P = p
    assert np.all(L_got[2] <= P // 2) and np.all(L_got[2] >= -(P // 2))


def test_wide_matrix_with_a_small_scratch_budget_takes_the_per_row_kernels(oracle, monkeypatch):
    """2 M columns: a row-group slice (one 256-byte line per label) is 512 MB, so a 1 GB scratch budget holds two -- far
    fewer than there are compute units.  The library must fall back to the per-row kernels (4 bytes per label and wave)
    instead of idling the chip or failing to allocate; same matrix as the oracle's."""
    import torch
    monkeypatch.setenv("SPASM_HIP_SCRATCH_GB", "1")
    p = 42013
    n, m, per_row = 9000, 2_000_000, 3
    rng = np.random.default_rng(77)
    ti = np.repeat(np.arange(n, dtype=np.int32), per_row)
    # every entry inside the first 4,000 columns (so at most 4,000 pivots and thousands of rows to reduce), one of them
    # moved far out for a few rows so that the 2 M columns are really addressed; the label space is 2 M wide either way
    tj = rng.integers(0, 4000, size=(n, per_row))
    far = rng.random(n) < 0.05
    tj[far, 2] = rng.integers(4000, m, size=int(far.sum()))
    tj = tj.reshape(-1).astype(np.int32)
    tx = rng.integers(1, p, size=n * per_row).astype(np.int64)
    A = oracle.compress(p, n, m, ti, tj, tx)
    npiv, perm, F = oracle.pivots_extract_structural(A, oracle.empty_fact(A.n, A.m, p))
    rows = perm[npiv:]
    assert len(rows) >= 2048
    want, _, _ = oracle.schur(A, rows, F)
    dA = spasm_amd.DeviceCsr.from_host(_as_product(A))
    dF = spasm_amd.DeviceFact(_fact(F))
    W = spasm_amd.SchurWorkspace(len(rows), A.m, 4 * want.nnz + (1 << 22))
    drows = torch.from_numpy(np.ascontiguousarray(rows, np.int32)).cuda()
    S, st = spasm_amd.dschur(dA, drows, dF, W)
    assert st.status == 0 and st.used_backsolve == 0 and st.used_group_kernel == 0
    assert st.rows_lds + st.rows_lds_big + st.rows_dense == len(rows)
    H = S.to_host()
    assert oracle.same_matrix(oracle.CSR(H.n, H.m, H.p, H.j, H.x, p), want)
