"""GPU parity of the SPARSE back-substituted image (spasm_amd/csrc/sparse_image.hip): S = A_n - A_p R with R = U_pp^-1 U_pn
kept as sparse fragments must be the matrix spasm_schur computes (spasm_schur.c:61-193 -> spasm_triangular.c:109-146), bit
for bit -- checked against the oracle (oracle/spasm_oracle.c, pinned on the compiled reference by tests/test_oracle.py).

The shapes aim at the kernels' own seams: column counts around the 8,192-column segments, rows of U and of A with more
than 64 entries (the batches of a wave), chains of one row per level (one launch each), wide levels, values at the bound of
the signed 16-bit arithmetic (p = 44,927), a fragment pool that is extended in mid-build, an output pool that is too small."""
import os

import numpy as np
import pytest

from conftest import ALL_TEST_MATRICES, matrix_path

import spasm_amd

pytestmark = pytest.mark.gpu

# signed 16-bit entries up to p = 44,927; beyond (round 5) 8-byte entries (column, residue) and 32-bit accumulators: every odd
# prime below 2^32, as the reference takes them (spasm_ZZp.c:5-15; its own tests run 65537 ... 4294967291)
PRIMES = [3, 257, 42013, 44927, 44939, 65537, 189812507, 4294967291]


def _as_product(A):
    return spasm_amd.Csr(A.n, A.m, A.p, A.j, A.x, A.prime)


def _fact(F):
    return spasm_amd.Fact(_as_product(F.U), F.qinv)


def _triangular_system(rng, p, npiv, nnon, nred, deps, reach, np_per_row, red_entries):
    """npiv pivot rows (row k: pivot on column k, then `deps(k)` pivotal entries within `reach` columns to the right and
    np_per_row entries on the nnon trailing non-pivotal columns), then nred rows to reduce."""
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


def _check(oracle, A, rows, F, want, p_out_want, p):
    S, p_out = spasm_amd.schur(_as_product(A), rows, _fact(F))
    assert np.array_equal(p_out, p_out_want)
    assert oracle.same_matrix(oracle.CSR(S.n, S.m, S.p, S.j, S.x, p), want)
    for i in range(S.n):
        jj, _ = S.row(i)
        assert np.all(np.diff(jj) > 0)
    return S


def _run(oracle, monkeypatch, p, n, m, ti, tj, tx, min_pivots, env=None):
    monkeypatch.setenv("SPASM_HIP_SPARSE_IMAGE", "1")
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    A = oracle.compress(p, n, m, ti, tj, tx)
    npiv, perm, F = oracle.pivots_extract_structural(A, oracle.empty_fact(A.n, A.m, p))
    assert npiv >= min_pivots
    rows = perm[npiv:]
    want, p_out_want, _ = oracle.schur(A, rows, F)
    _check(oracle, A, rows, F, want, p_out_want, p)
    return A, rows, F, want


def _stats_of_device_call(A, rows, F, pool):
    import torch
    dA = spasm_amd.DeviceCsr.from_host(_as_product(A))
    dF = spasm_amd.DeviceFact(_fact(F))
    drows = torch.from_numpy(np.ascontiguousarray(rows, np.int32)).cuda()
    W = spasm_amd.SchurWorkspace(len(rows), A.m, pool)
    S, st = spasm_amd.dschur(dA, drows, dF, W, fetch=True)
    return S, st, W, dF, dA, drows


@pytest.mark.parametrize("p", PRIMES)
@pytest.mark.parametrize("name", ALL_TEST_MATRICES)
def test_sparse_image_on_the_reference_matrices(oracle, monkeypatch, name, p):
    """the reference's own 32 test matrices (tests/CMakeLists.txt:75-110): pivots by the oracle, every non-pivotal row reduced"""
    monkeypatch.setenv("SPASM_HIP_SPARSE_IMAGE", "1")
    A = oracle.load_sms(matrix_path(name), p)
    npiv, perm, F = oracle.pivots_extract_structural(A, oracle.empty_fact(A.n, A.m, p))
    rows = perm[npiv:]
    if len(rows) == 0 or npiv == 0:
        pytest.skip("nothing to reduce")
    want, p_out_want, _ = oracle.schur(A, rows, F)
    _check(oracle, A, rows, F, want, p_out_want, p)


@pytest.mark.parametrize("p", PRIMES)
@pytest.mark.parametrize("nnon", [1, 63, 8191, 8192, 8193, 16385, 20011])
def test_sparse_image_column_counts(oracle, monkeypatch, p, nnon):
    rng = np.random.default_rng(nnon)
    sysm = _triangular_system(rng, p, npiv=500, nnon=nnon, nred=300, deps=lambda k: 2, reach=40, np_per_row=3, red_entries=5)
    _run(oracle, monkeypatch, p, *sysm, min_pivots=400)


@pytest.mark.parametrize("p", [257, 42013])
def test_sparse_image_chain_of_single_row_levels(oracle, monkeypatch, p):
    """one dependency on the next row: 1500 levels of one row each, one launch per level"""
    rng = np.random.default_rng(5)
    sysm = _triangular_system(rng, p, npiv=1500, nnon=9000, nred=400, deps=lambda k: 1, reach=1, np_per_row=2, red_entries=4)
    _run(oracle, monkeypatch, p, *sysm, min_pivots=1200)


@pytest.mark.parametrize("p", PRIMES[1:])
@pytest.mark.parametrize("ndeps,reach", [(3, 3000), (40, 3000), (150, 800)])
def test_sparse_image_many_dependencies(oracle, monkeypatch, p, ndeps, reach):
    """rows of U with more pivotal entries than a wave has lanes (batches of 64), rows that fill up"""
    rng = np.random.default_rng(ndeps + reach)
    sysm = _triangular_system(rng, p, npiv=3000, nnon=300, nred=500, deps=lambda k: ndeps, reach=reach, np_per_row=4, red_entries=6)
    _run(oracle, monkeypatch, p, *sysm, min_pivots=1500)


@pytest.mark.parametrize("p", [257, 42013])
def test_sparse_image_long_input_rows(oracle, monkeypatch, p):
    """rows of A with 300 entries (what the second round of a flow sees): five batches of 64 per row and segment"""
    rng = np.random.default_rng(11)
    sysm = _triangular_system(rng, p, npiv=2000, nnon=9000, nred=300, deps=lambda k: 3, reach=500, np_per_row=6, red_entries=300)
    _run(oracle, monkeypatch, p, *sysm, min_pivots=1500)


def test_sparse_image_wide_levels(oracle, monkeypatch):
    p = 42013
    rng = np.random.default_rng(8)
    sysm = _triangular_system(rng, p, npiv=4000, nnon=12000, nred=600, deps=lambda k: 1 if k % 7 == 0 else 0, reach=3000,
                              np_per_row=5, red_entries=8)
    _run(oracle, monkeypatch, p, *sysm, min_pivots=3000)


@pytest.mark.parametrize("p", [42013, 44927])
def test_sparse_image_extreme_values(oracle, monkeypatch, p):
    """every entry is (p - 1) / 2 or (p + 1) / 2: accumulator + product as close to 2^31 as the data can push it"""
    rng = np.random.default_rng(77)
    n, m, ti, tj, tx = _triangular_system(rng, p, npiv=1500, nnon=70, nred=400, deps=lambda k: 4, reach=30, np_per_row=3, red_entries=8)
    tx = np.where(tx == 1, 1, np.where(rng.integers(0, 2, size=len(tx)) == 0, (p - 1) // 2, (p + 1) // 2)).astype(np.int64)
    _run(oracle, monkeypatch, p, n, m, ti, tj, tx, min_pivots=1000)


@pytest.mark.parametrize("chunk", [1024, 4096, 65536])
def test_sparse_image_pool_extended_in_mid_build(oracle, monkeypatch, chunk):
    """a first chunk far too small: the build runs out of room in some level, takes the next chunk (twice the size) and redoes
    the levels from there -- the fragments end up spread over several chunks, S is the same"""
    p = 42013
    rng = np.random.default_rng(chunk)
    sysm = _triangular_system(rng, p, npiv=3000, nnon=9000, nred=500, deps=lambda k: 3, reach=200, np_per_row=4, red_entries=6)
    _run(oracle, monkeypatch, p, *sysm, min_pivots=2000, env={"SPASM_HIP_SPARSE_IMAGE_CHUNK": str(chunk), "SPASM_HIP_SPARSE_IMAGE_GB": "1"})


def test_sparse_image_device_call_statistics_rebuild_and_small_pool(oracle, monkeypatch):
    """the device-level call: the statistics say which path ran; an output pool that is too small is reported (status 1) and a
    larger one succeeds; forgetting the image and building it again (what bench.py does every step) gives the same S; a second
    batch on the same factor reuses R"""
    import torch
    monkeypatch.setenv("SPASM_HIP_SPARSE_IMAGE", "1")
    p = 42013
    rng = np.random.default_rng(3)
    n, m, ti, tj, tx = _triangular_system(rng, p, npiv=3000, nnon=9000, nred=800, deps=lambda k: 3, reach=300, np_per_row=4, red_entries=6)
    A = oracle.compress(p, n, m, ti, tj, tx)
    npiv, perm, F = oracle.pivots_extract_structural(A, oracle.empty_fact(A.n, A.m, p))
    rows = perm[npiv:]
    want, _, _ = oracle.schur(A, rows, F)
    S, st, W, dF, dA, drows = _stats_of_device_call(A, rows, F, 64)
    assert st.status == 1 and st.used_sparse_image == 1
    W.close()
    W = spasm_amd.SchurWorkspace(len(rows), A.m, want.nnz + 100)
    S, st = spasm_amd.dschur(dA, drows, dF, W, fetch=True)
    assert st.status == 0 and st.used_sparse_image == 1 and st.sparse_image_built == 0 and st.nnz == want.nnz
    assert st.sparse_image_nnz > 0 and st.sparse_image_levels > 1
    H = S.to_host()
    assert oracle.same_matrix(oracle.CSR(H.n, H.m, H.p, H.j, H.x, p), want)
    dF.forget()
    S2, st2 = spasm_amd.dschur(dA, drows, dF, W, fetch=True)
    assert st2.status == 0 and st2.sparse_image_built == 1 and st2.sparse_image_nnz == st.sparse_image_nnz
    assert torch.equal(S2.p, S.p) and torch.equal(S2.j, S.j) and torch.equal(S2.x, S.x)
    # a sub-batch, R being there
    assert len(rows) >= 60
    sub = drows[10:50].contiguous()
    S3, st3 = spasm_amd.dschur(dA, sub, dF, W, fetch=True)
    assert st3.status == 0 and st3.used_sparse_image == 1 and st3.sparse_image_built == 0
    H3 = S3.to_host()
    want3, _, _ = oracle.schur(A, rows[10:50], F)
    assert oracle.same_matrix(oracle.CSR(H3.n, H3.m, H3.p, H3.j, H3.x, p), want3)
    W.close()
    dF.close()


@pytest.mark.parametrize("p", [65521, 67108859, 4294967291])
def test_primes_beyond_the_signed_arithmetic_take_the_wide_variant(oracle, monkeypatch, p):
    """beyond p = 44,927 the image keeps (column, 32-bit residue) entries: the device call says the sparse image ran, S is the
    oracle's; a second batch reuses R; extreme values (p - 1 everywhere) stay exact"""
    import torch
    rng = np.random.default_rng(21)
    n, m, ti, tj, tx = _triangular_system(rng, p, npiv=3000, nnon=9000, nred=500, deps=lambda k: 3, reach=200, np_per_row=4, red_entries=6)
    tx = np.where(tx == 1, 1, np.where(rng.integers(0, 3, size=len(tx)) == 0, p - 1, tx)).astype(np.int64)
    # (R of this system is a quarter full: 8-byte entries need more than the 256 MB a small factor may take by default)
    A, rows, F, want = _run(oracle, monkeypatch, p, n, m, ti, tj, tx, min_pivots=2000, env={"SPASM_HIP_SPARSE_IMAGE_GB": "2"})
    S, st, W, dF, dA, drows = _stats_of_device_call(A, rows, F, want.nnz + 4096)
    assert st.status == 0 and st.used_sparse_image == 1 and st.nnz == want.nnz
    H = S.to_host()
    assert oracle.same_matrix(oracle.CSR(H.n, H.m, H.p, H.j, H.x, p), want)
    assert len(rows) >= 30
    S2, st2 = spasm_amd.dschur(dA, drows[5:25].contiguous(), dF, W, fetch=True)
    assert st2.status == 0 and st2.used_sparse_image == 1 and st2.sparse_image_built == 0
    want2, _, _ = oracle.schur(A, rows[5:25], F)
    H2 = S2.to_host()
    assert oracle.same_matrix(oracle.CSR(H2.n, H2.m, H2.p, H2.j, H2.x, p), want2)
    W.close()
    dF.close()


@pytest.mark.parametrize("chunk", [0, 2048])
def test_the_wide_variant_on_a_small_prime_gives_the_same_schur_complement(oracle, monkeypatch, chunk):
    """SPASM_HIP_SPARSE_IMAGE_WIDE=1: the 32-bit variant for p = 42013 -- the Schur complement of the 16-bit variant and of the
    oracle, with levels of one row, wide levels, more than 64 dependencies per row, and a pool extended in mid-build"""
    p = 42013
    rng = np.random.default_rng(9 + chunk)
    sysm = _triangular_system(rng, p, npiv=2500, nnon=9500, nred=400, deps=lambda k: 70 if k % 97 == 0 else 3, reach=400, np_per_row=5, red_entries=70)
    env = {"SPASM_HIP_EXPERIMENT": "1", "SPASM_HIP_SPARSE_IMAGE_WIDE": "1", "SPASM_HIP_SPARSE_IMAGE_GB": "2"}
    if chunk:
        env.update({"SPASM_HIP_SPARSE_IMAGE_CHUNK": str(chunk)})
    _run(oracle, monkeypatch, p, *sysm, min_pivots=1500, env=env)


def test_a_build_that_finds_R_dense_gives_its_memory_back_and_the_other_paths_take_the_batch(oracle, monkeypatch):
    """R of this factor is dense (every pivot row holds 300 of the 2,000 non-pivotal columns and the rows chain): with first
    chunks of 1,024 entries the build runs out of chunks and gives up.  The same call must then come out right through
    another path, and what the attempt allocated must be back (the advisor's finding of round 4: the fallback sized its
    scratch by free memory with up to a third of the HBM still held by the failed attempt): the factor holds no more device
    memory afterwards than one that never tried."""
    import torch
    monkeypatch.setenv("SPASM_HIP_SPARSE_IMAGE_CHUNK", "1024")
    p = 42013
    rng = np.random.default_rng(5)
    n, m, ti, tj, tx = _triangular_system(rng, p, npiv=4000, nnon=2000, nred=300, deps=lambda k: 3, reach=60, np_per_row=300, red_entries=6)
    A = oracle.compress(p, n, m, ti, tj, tx)
    npiv, perm, F = oracle.pivots_extract_structural(A, oracle.empty_fact(A.n, A.m, p))
    rows = perm[npiv:]
    want, _, _ = oracle.schur(A, rows, F)

    def held_by_the_factor(mode):
        monkeypatch.setenv("SPASM_HIP_SPARSE_IMAGE", mode)
        torch.cuda.synchronize()
        spasm_amd.release_cached_memory()
        free_before = torch.cuda.mem_get_info()[0]
        S, st, W, dF, dA, drows = _stats_of_device_call(A, rows, F, want.nnz + 4096)
        assert st.status == 0 and st.used_sparse_image == 0 and st.nnz == want.nnz
        H = S.to_host()
        assert oracle.same_matrix(oracle.CSR(H.n, H.m, H.p, H.j, H.x, p), want)
        # a second batch on the same factor: the image is not tried again
        S2, st2 = spasm_amd.dschur(dA, drows, dF, W, fetch=True)
        assert st2.status == 0 and st2.used_sparse_image == 0 and st2.nnz == want.nnz
        del S, S2
        W.close()
        spasm_amd.release_cached_memory()
        torch.cuda.synchronize()
        held = free_before - torch.cuda.mem_get_info()[0]
        dF.close()
        del dA, drows
        return held

    never_tried = held_by_the_factor("0")
    tried_and_failed = held_by_the_factor("1")
    # (the pool chunks of the failed attempt: 1,024 * (2^12 - 1) entries = 16 MB, and 4,000 fragment words)
    assert tried_and_failed <= never_tried + (4 << 20), "the factor holds %.1f MB more after a failed sparse-image build" % ((tried_and_failed - never_tried) / 1048576.0)
