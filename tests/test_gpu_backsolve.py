"""GPU parity of the back-substituted factor image (spasm_amd/csrc/backsolve.hip): S = A_n - A_p R with
R = U_pp^-1 U_pn must be the matrix spasm_schur computes (spasm_schur.c:61-193), bit for bit.

The shapes below aim at the kernel's own seams: chains longer than a chunk (768 rows), levels wider than a
workgroup pass, rows with more than two dependencies outside their chunk, dependencies that straddle chunk
boundaries, column counts around the 16-column slabs and 64-column tiles, and every arithmetic of the kernels:
signed 16-bit entries with deferred reduction (p <= 44927, the largest prime whose four-term sums fit 32 bits),
unsigned 16-bit entries (p < 2^16, or SPASM_HIP_BS_SIGNED=0), 32-bit Montgomery entries."""
import numpy as np
import pytest

import spasm_amd

pytestmark = pytest.mark.gpu

# (prime, SPASM_HIP_BS_SIGNED)
SMALL = [(42013, "1"), (42013, "0"), (44927, "1"), (65521, "1")]
ARITH = [(3, "1"), (257, "1"), (257, "0")] + SMALL + [(4294967291, "1")]


def _as_product(A):
    return spasm_amd.Csr(A.n, A.m, A.p, A.j, A.x, A.prime)


def _fact(F):
    return spasm_amd.Fact(_as_product(F.U), F.qinv)


def _triangular_system(rng, p, npiv, nnon, nred, deps, reach, np_per_row, red_entries):
    """npiv pivot rows (row k: pivot on column k, then `deps(k)` pivotal entries within `reach` columns to the
    right and np_per_row entries on the nnon trailing non-pivotal columns), then nred rows to reduce."""
    m = npiv + nnon
    ti, tj, tx = [], [], []
    for k in range(npiv):
        cols = [k]
        room = min(reach, npiv - k - 1)
        d = min(deps(k), room)
        if d > 0:
            cols += list(k + 1 + rng.choice(room, size=d, replace=False))
        if nnon > 0 and np_per_row > 0:
            cols += list(npiv + rng.choice(nnon, size=min(np_per_row, nnon), replace=False))
        ti += [k] * len(cols)
        tj += [int(c) for c in cols]
        tx += [1] + [int(v) for v in rng.integers(1, p, size=len(cols) - 1)]
    for k in range(nred):
        cols = rng.choice(m, size=min(red_entries, m), replace=False)
        ti += [npiv + k] * len(cols)
        tj += [int(c) for c in cols]
        tx += [int(v) for v in rng.integers(1, p, size=len(cols))]
    return npiv + nred, m, np.array(ti, np.int32), np.array(tj, np.int32), np.array(tx, np.int64)


def _run(oracle, monkeypatch, p, n, m, ti, tj, tx, min_pivots, signed="1"):
    monkeypatch.setenv("SPASM_HIP_BACKSOLVE", "1")
    monkeypatch.setenv("SPASM_HIP_BS_SIGNED", signed)
    A = oracle.compress(p, n, m, ti, tj, tx)
    npiv, perm, F = oracle.pivots_extract_structural(A, oracle.empty_fact(A.n, A.m, p))
    assert npiv >= min_pivots
    rows = perm[npiv:]
    want, p_out_want, _ = oracle.schur(A, rows, F)
    S, p_out = spasm_amd.schur(_as_product(A), rows, _fact(F))
    assert np.array_equal(p_out, p_out_want)
    assert oracle.same_matrix(oracle.CSR(S.n, S.m, S.p, S.j, S.x, p), want)
    for i in range(S.n):
        jj, _ = S.row(i)
        assert np.all(np.diff(jj) > 0)
    return A, rows, F, want


@pytest.mark.parametrize("p,signed", ARITH)
@pytest.mark.parametrize("nnon", [1, 15, 16, 17, 63, 64, 65, 300, 513])
def test_backsolve_column_counts(oracle, monkeypatch, p, signed, nnon):
    rng = np.random.default_rng(nnon)
    sysm = _triangular_system(rng, p, npiv=500, nnon=nnon, nred=300, deps=lambda k: 2, reach=40, np_per_row=3, red_entries=5)
    _run(oracle, monkeypatch, p, *sysm, min_pivots=400, signed=signed)


@pytest.mark.parametrize("p,signed", SMALL + [(4294967291, "1")])
def test_backsolve_chain_longer_than_a_chunk(oracle, monkeypatch, p, signed):
    """one dependency on the next row: 2500 levels of one row each -- every chunk is a pure chain."""
    rng = np.random.default_rng(5)
    sysm = _triangular_system(rng, p, npiv=2500, nnon=40, nred=400, deps=lambda k: 1, reach=1, np_per_row=2, red_entries=4)
    _run(oracle, monkeypatch, p, *sysm, min_pivots=2000, signed=signed)


@pytest.mark.parametrize("p,signed", ARITH[1:])
@pytest.mark.parametrize("ndeps,reach", [(3, 3000), (8, 3000), (40, 3000), (6, 100), (150, 800)])
def test_backsolve_many_dependencies(oracle, monkeypatch, p, signed, ndeps, reach):
    """rows of U with many pivotal entries, near (inside the chunk) and far (beyond the two inline ones)."""
    rng = np.random.default_rng(ndeps + reach)
    sysm = _triangular_system(rng, p, npiv=3000, nnon=100, nred=500, deps=lambda k: ndeps, reach=reach, np_per_row=4, red_entries=6)
    _run(oracle, monkeypatch, p, *sysm, min_pivots=1500, signed=signed)


@pytest.mark.parametrize("signed", ["1", "0"])
def test_backsolve_wide_levels(oracle, monkeypatch, signed):
    """no dependencies between most pivot rows: a few levels of thousands of rows (levels wider than a chunk)."""
    p = 42013
    rng = np.random.default_rng(8)
    sysm = _triangular_system(rng, p, npiv=4000, nnon=200, nred=600, deps=lambda k: 1 if k % 7 == 0 else 0, reach=3000,
                              np_per_row=5, red_entries=8)
    _run(oracle, monkeypatch, p, *sysm, min_pivots=3000, signed=signed)


@pytest.mark.parametrize("p,signed", SMALL)
def test_backsolve_extreme_values(oracle, monkeypatch, p, signed):
    """every entry is (p - 1) / 2 or (p + 1) / 2 -- the balanced representatives of largest magnitude: sums of four
    products come as close to 2^31 as the data can push them."""
    rng = np.random.default_rng(77)
    n, m, ti, tj, tx = _triangular_system(rng, p, npiv=1500, nnon=70, nred=400, deps=lambda k: 4, reach=30, np_per_row=3, red_entries=8)
    tx = np.where(tx == 1, 1, np.where(rng.integers(0, 2, size=len(tx)) == 0, (p - 1) // 2, (p + 1) // 2)).astype(np.int64)
    _run(oracle, monkeypatch, p, n, m, ti, tj, tx, min_pivots=1000, signed=signed)


@pytest.mark.parametrize("env", [{"SPASM_HIP_BS_STAGED": "0"}, {"SPASM_HIP_STAGE_ROWS": "100"}, {"SPASM_HIP_STAGE_ROWS": "1"},
                                 {"SPASM_HIP_BS_DIRECT": "0"}, {"SPASM_HIP_BS_SPARSE_INIT": "0"}, {"SPASM_HIP_BS_SHAPE": "0"}, {"SPASM_HIP_BS_SHAPE": "1"}])
@pytest.mark.parametrize("p", [257, 42013, 65521, 4294967291])
def test_backsolve_output_modes(oracle, monkeypatch, p, env):
    """every arithmetic: look-back output instead of the staged one, staged output in slices of 100 rows and of
    one row, rows through the pool + gather pass; the build of R from a pre-filled R, in the other workgroup shapes."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(61)
    sysm = _triangular_system(rng, p, npiv=1500, nnon=333, nred=777, deps=lambda k: 3, reach=40, np_per_row=3, red_entries=7)
    _run(oracle, monkeypatch, p, *sysm, min_pivots=1000)


@pytest.mark.parametrize("signed", ["1", "0"])
def test_backsolve_long_input_rows(oracle, monkeypatch, signed):
    """rows to reduce with hundreds of pivotal entries (more than one pass of the apply kernel's list)."""
    p = 42013
    rng = np.random.default_rng(9)
    sysm = _triangular_system(rng, p, npiv=2000, nnon=130, nred=200, deps=lambda k: 2, reach=50, np_per_row=3, red_entries=900)
    _run(oracle, monkeypatch, p, *sysm, min_pivots=1500, signed=signed)


@pytest.mark.parametrize("p,signed", SMALL + [(4294967291, "1")])
def test_backsolve_dense_rows(oracle, monkeypatch, p, signed):
    """spasm_schur_dense (spasm_schur.c:257-343) through the same image."""
    monkeypatch.setenv("SPASM_HIP_BACKSOLVE", "1")
    monkeypatch.setenv("SPASM_HIP_BS_SIGNED", signed)
    rng = np.random.default_rng(21)
    n, m, ti, tj, tx = _triangular_system(rng, p, npiv=1200, nnon=90, nred=300, deps=lambda k: 2, reach=60, np_per_row=3, red_entries=6)
    A = oracle.compress(p, n, m, ti, tj, tx)
    npiv, perm, F = oracle.pivots_extract_structural(A, oracle.empty_fact(A.n, A.m, p))
    rows = perm[npiv:]
    want, q_want, p_out_want = oracle.schur_dense(A, rows, F)
    got, q, p_out = spasm_amd.schur_dense(_as_product(A), rows, _fact(F))
    assert np.array_equal(q, q_want) and np.array_equal(p_out, p_out_want)
    assert np.array_equal(np.asarray(got, np.int64) % p, np.asarray(want, np.int64) % p)


@pytest.mark.parametrize("p", [42013, 257])
@pytest.mark.parametrize("nnon", [24577, 32768, 33000, 41000])
def test_backsolve_rows_wider_than_the_lds(oracle, monkeypatch, p, nnon):
    """more than 24,576 non-pivotal columns (mk14.b4 has 42,356): the apply kernel produces a row in segments of 8,192
    columns -- whole segments, a last partial one, entries of the input row in every segment; sparse rows (staged output,
    also in slices) and dense rows.  With that many free columns every row of the matrix finds a structural pivot, so the
    factor is taken from the pivot rows alone and the other rows are reduced by it (spasm_schur takes any list of rows)."""
    import torch
    monkeypatch.setenv("SPASM_HIP_BACKSOLVE", "1")
    rng = np.random.default_rng(nnon)
    npiv, nred = 1200, 150
    n, m, ti, tj, tx = _triangular_system(rng, p, npiv=npiv, nnon=nnon, nred=nred, deps=lambda k: 3, reach=50, np_per_row=40, red_entries=60)
    A = oracle.compress(p, n, m, ti, tj, tx)
    top = ti < npiv
    P = oracle.compress(p, npiv, m, ti[top], tj[top], tx[top])
    found, perm, F = oracle.pivots_extract_structural(P, oracle.empty_fact(P.n, P.m, p))
    assert found == npiv
    rows = np.arange(npiv, n, dtype=np.int32)
    want, p_out_want, _ = oracle.schur(A, rows, F)
    assert want.nnz > nred * nnon // 4               # (the rows fill in: every segment holds entries)
    for stage_rows in (None, "37"):
        if stage_rows:
            monkeypatch.setenv("SPASM_HIP_STAGE_ROWS", stage_rows)
        S, p_out = spasm_amd.schur(_as_product(A), rows, _fact(F))
        assert np.array_equal(p_out, p_out_want)
        assert oracle.same_matrix(oracle.CSR(S.n, S.m, S.p, S.j, S.x, p), want)
        for i in range(S.n):
            jj, _ = S.row(i)
            assert np.all(np.diff(jj) > 0)
    monkeypatch.delenv("SPASM_HIP_STAGE_ROWS")
    dense_want, q_want, p_out_want = oracle.schur_dense(A, rows, F)
    got, q, p_out = spasm_amd.schur_dense(_as_product(A), rows, _fact(F))
    assert np.array_equal(q, q_want) and np.array_equal(p_out, p_out_want)
    assert np.array_equal(np.asarray(got, np.int64) % p, np.asarray(dense_want, np.int64) % p)
    # ... and it IS the image that produced them (device API: the statistics say which path ran)
    dA = spasm_amd.DeviceCsr.from_host(_as_product(A))
    dF = spasm_amd.DeviceFact(_fact(F))
    W = spasm_amd.SchurWorkspace(len(rows), A.m, 2 * want.nnz + (1 << 20))
    drows = torch.from_numpy(np.ascontiguousarray(rows, np.int32)).cuda()
    S, st = spasm_amd.dschur(dA, drows, dF, W)
    assert st.status == 0 and st.used_backsolve == 1 and st.nnz == want.nnz
    assert st.kernel_other.decode().split("<")[0] == "bs_apply_s16_kernel" or st.kernel.decode().split("<")[0] == "bs_apply_s16_kernel"
    H = S.to_host()
    assert oracle.same_matrix(oracle.CSR(H.n, H.m, H.p, H.j, H.x, p), want)


@pytest.mark.parametrize("signed", ["1", "0"])
def test_backsolve_is_rebuilt_after_forget(oracle, monkeypatch, signed):
    """device API: the image is built by the first call, reused by the second, rebuilt after forget()."""
    import torch
    monkeypatch.setenv("SPASM_HIP_BACKSOLVE", "1")
    monkeypatch.setenv("SPASM_HIP_BS_SIGNED", signed)
    p = 42013
    rng = np.random.default_rng(33)
    n, m, ti, tj, tx = _triangular_system(rng, p, npiv=1500, nnon=70, nred=400, deps=lambda k: 2, reach=30, np_per_row=3, red_entries=5)
    A = oracle.compress(p, n, m, ti, tj, tx)
    npiv, perm, F = oracle.pivots_extract_structural(A, oracle.empty_fact(A.n, A.m, p))
    rows = perm[npiv:]
    want, _, _ = oracle.schur(A, rows, F)
    dA = spasm_amd.DeviceCsr.from_host(_as_product(A))
    dF = spasm_amd.DeviceFact(_fact(F))
    W = spasm_amd.SchurWorkspace(len(rows), A.m, 4 * want.nnz + (1 << 20))
    drows = torch.from_numpy(np.ascontiguousarray(rows, np.int32)).cuda()
    built = []
    for step in range(3):
        if step == 2:
            dF.forget()
        S, st = spasm_amd.dschur(dA, drows, dF, W)
        assert st.status == 0 and st.used_backsolve == 1 and st.nnz == want.nnz
        assert {st.kernel.decode().split("<")[0], st.kernel_other.decode().split("<")[0]} == {"backsolve_kernel", "bs_apply_s16_kernel" if signed == "1" else "bs_apply_kernel"}
        built.append(st.backsolve_built)
        H = S.to_host()
        assert oracle.same_matrix(oracle.CSR(H.n, H.m, H.p, H.j, H.x, p), want)
    assert built == [1, 0, 1]
