"""Pins the oracle (oracle/spasm_oracle.c) before anything is checked against it.

Two anchors (see oracle/spasm_oracle.h):
  * the real reference compiled into oracle/_ref/libspasm_ref.so -- exact
    equality, entry order included, on the reference's own test matrices and
    moduli (skipped only where that library could not be built);
  * the properties the reference's tests assert (tests/GFp.c, tests/schur.c,
    tests/schur_dense.c, tests/echelonize.c, tests/dense_rref_ffpack.c) and an
    independent numpy elimination for ranks.
"""
import numpy as np
import pytest

from conftest import ALL_MODULI, ALL_TEST_MATRICES, matrix_path

SMALL_SET = [m for m in ALL_TEST_MATRICES if m not in ("mat364.sms", "trefethen_500.sms", "medium.sms", "m1.sms")]


def need_ref(orc):
    if not orc.ref_available():
        pytest.skip("oracle/_ref/libspasm_ref.so not built (no /root/reference)")


def numpy_rank(D, p):
    """plain Gaussian elimination mod p (p < 2^31) -- independent of the oracle."""
    D = np.mod(D.astype(np.int64), p)
    n, m = D.shape
    r = 0
    for c in range(m):
        if r == n:
            break
        nz = np.nonzero(D[r:, c])[0]
        if len(nz) == 0:
            continue
        s = r + int(nz[0])
        if s != r:
            D[[r, s]] = D[[s, r]]
        inv = pow(int(D[r, c]), p - 2, p)
        D[r] = (D[r] * inv) % p
        rows = np.nonzero(D[:, c])[0]
        rows = rows[rows != r]
        if len(rows):
            D[rows] = (D[rows] - np.outer(D[rows, c], D[r])) % p
        r += 1
    return r


# ---------------------------------------------------------------- field
@pytest.mark.parametrize("p", [3, 257, 65537])
def test_field_inverse_all(oracle, p):
    L = oracle.lib()
    for a in range(1, min(p, 3000)):
        x = L.orc_zp_init(p, a)
        y = L.orc_zp_inverse(p, x)
        assert L.orc_zp_mul(p, x, y) == 1
        assert -(p // 2) <= y <= p // 2


@pytest.mark.parametrize("p", ALL_MODULI + [0x7fffffff, 3037000493])
def test_field_matches_reference(oracle, p):
    need_ref(oracle)
    L, R = oracle.lib(), oracle.ref()
    F = oracle.ref_field(p)
    import ctypes as C
    rng = np.random.default_rng(p % 1000)
    vals = rng.integers(-(p // 2), p // 2 + 1, size=(2000, 3))
    for a, b, c in vals.tolist():
        assert L.orc_zp_add(p, a, b) == R.spasm_ZZp_add(C.byref(F), a, b)
        assert L.orc_zp_sub(p, a, b) == R.spasm_ZZp_sub(C.byref(F), a, b)
        assert L.orc_zp_mul(p, a, b) == R.spasm_ZZp_mul(C.byref(F), a, b)
        assert L.orc_zp_axpy(p, a, b, c) == R.spasm_ZZp_axpy(C.byref(F), a, b, c)
        if a % p:
            assert L.orc_zp_inverse(p, a) == R.spasm_ZZp_inverse(C.byref(F), a)
    for big in rng.integers(-2**62, 2**62, size=200).tolist():
        assert L.orc_zp_init(p, big) == R.spasm_ZZp_init(C.byref(F), big)


def test_prng_golden_is_reference(oracle):
    """tests/Expected/prng is reproduced by the compiled reference (sanity of the _ref build)."""
    need_ref(oracle)
    import ctypes as C
    R = oracle.ref()
    lines = open(matrix_path("../Expected/prng")).read().strip().split("\n")
    cases = [(257, 0, 0), (257, 0, 1), (257, 1, 0), (257, 1, 1), (65537, 0xdead00000000beef, 0)]
    for line, (p, seed, seq) in zip(lines, cases):
        want = [int(t) for t in line.split("out=")[1].split(",")]
        ctx = C.create_string_buffer(512)
        R.spasm_prng_seed_simple(p, seed, seq, ctx)
        got = [R.spasm_prng_ZZp(ctx) for _ in range(10)]
        assert got == want


# ---------------------------------------------------------------- compress
@pytest.mark.parametrize("name", ALL_TEST_MATRICES)
def test_compress_matches_reference(oracle, name):
    need_ref(oracle)
    for p in (257, 4294967291):
        n, m, ti, tj, tx = oracle.read_sms(matrix_path(name))
        A = oracle.compress(p, n, m, ti, tj, tx)
        B = oracle.ref_compress(p, n, m, ti, tj, tx)
        assert (A.n, A.m) == (B.n, B.m)
        assert np.array_equal(A.p, B.p) and np.array_equal(A.j, B.j) and np.array_equal(A.x, B.x)


# ---------------------------------------------------------------- pivots + solve + schur
def _round0(oracle, name, p, use_ref=False):
    A = oracle.load_sms(matrix_path(name), p)
    F0 = oracle.empty_fact(A.n, A.m, p)
    if use_ref:
        return (A,) + oracle.ref_pivots_extract_structural(A, F0)
    return (A,) + oracle.pivots_extract_structural(A, F0)


@pytest.mark.parametrize("name", ALL_TEST_MATRICES)
@pytest.mark.parametrize("p", [257, 189812507])
def test_pivots_match_reference(oracle, name, p):
    need_ref(oracle)
    A, npiv, perm, F = _round0(oracle, name, p)
    _, npiv_r, perm_r, F_r = _round0(oracle, name, p, use_ref=True)
    assert npiv == npiv_r
    assert np.array_equal(perm, perm_r)
    assert np.array_equal(F.qinv, F_r.qinv)
    assert np.array_equal(F.U.p, F_r.U.p) and np.array_equal(F.U.j, F_r.U.j) and np.array_equal(F.U.x, F_r.U.x)


@pytest.mark.parametrize("name", SMALL_SET)
def test_triangular_solve_matches_reference(oracle, name):
    need_ref(oracle)
    p = 65537
    A, npiv, perm, F = _round0(oracle, name, p)
    for i in perm[npiv:npiv + 25]:
        pat, x = oracle.solve_row(F.U, F.qinv, A, int(i))
        pat_r, x_r = oracle.ref_solve_row(F.U, F.qinv, A, int(i))
        assert np.array_equal(pat, pat_r)
        assert np.array_equal(x[pat], x_r[pat_r])


@pytest.mark.parametrize("name", ALL_TEST_MATRICES)
@pytest.mark.parametrize("p", ALL_MODULI)
def test_schur_matches_reference(oracle, name, p):
    need_ref(oracle)
    A, npiv, perm, F = _round0(oracle, name, p)
    rows = perm[npiv:]
    S, p_out, _ = oracle.schur(A, rows, F)
    S_r, p_out_r = oracle.ref_schur(A, rows, F, threads=1)
    assert np.array_equal(p_out, p_out_r)
    assert np.array_equal(S.p, S_r.p) and np.array_equal(S.j, S_r.j) and np.array_equal(S.x, S_r.x)
    # property of tests/schur.c:62-72: nothing is left under a pivot
    assert np.all(F.qinv[S.j] < 0)


@pytest.mark.parametrize("name", SMALL_SET)
def test_schur_multithreaded_reference_same_matrix(oracle, name):
    """with several OpenMP threads the reference permutes rows; as a set of rows it is the same."""
    need_ref(oracle)
    p = 42013
    A, npiv, perm, F = _round0(oracle, name, p)
    rows = perm[npiv:]
    S, p_out, _ = oracle.schur(A, rows, F)
    S_r, p_out_r = oracle.ref_schur(A, rows, F, threads=4)
    mine = {int(r): c for r, c in zip(p_out, S.canonical())}
    for r, (jj, xx) in zip(p_out_r, S_r.canonical()):
        a, b = mine[int(r)]
        assert np.array_equal(a, jj) and np.array_equal(b, xx)


@pytest.mark.parametrize("name", SMALL_SET)
@pytest.mark.parametrize("p", [257, 65537, 4294967291])
def test_schur_dense_matches_reference(oracle, name, p):
    need_ref(oracle)
    A, npiv, perm, F = _round0(oracle, name, p)
    rows = perm[npiv:]
    S, q, p_out = oracle.schur_dense(A, rows, F)
    S_r, q_r, p_out_r = oracle.ref_schur_dense(A, rows, F)
    assert np.array_equal(q, q_r) and np.array_equal(p_out, p_out_r)
    assert np.array_equal(S, S_r)


# ---------------------------------------------------------------- dense rref
@pytest.mark.parametrize("name", SMALL_SET)
@pytest.mark.parametrize("p", ALL_MODULI)
def test_dense_rref_contract(oracle, name, p):
    A = oracle.load_sms(matrix_path(name), p)
    if A.n == 0 or A.m == 0:
        return
    D = A.to_dense()
    r, R, qinv = oracle.dense_rref(p, D)
    m = A.m
    assert sorted(qinv.tolist()) == list(range(m))
    if p < 2**31:
        assert r == numpy_rank(D, p)
    # rebuild the RREF in natural column order and check (a) it is reduced, (b) rowspan(A) inside
    E = np.zeros((r, m), dtype=object)
    for i in range(r):
        E[i, qinv[i]] = 1
        for k in range(r, m):
            E[i, qinv[k]] = int(R[i, k]) % p
    piv = [int(qinv[i]) for i in range(r)]
    assert piv == sorted(piv)                       # pivots = column rank profile, in row order
    for i in range(r):
        assert all(E[i, c] == 0 for c in range(piv[i]))
        for i2 in range(r):
            assert E[i2, piv[i]] == (1 if i2 == i else 0)
    for i in range(A.n):                            # tests/dense_rref_ffpack.c:85-106
        x = [int(v) for v in D[i]]
        for k in range(r):
            a = x[piv[k]]
            if a:
                for c in range(m):
                    x[c] = (x[c] - a * int(E[k, c])) % p
        assert not any(x)


# ---------------------------------------------------------------- driver
@pytest.mark.parametrize("name", ALL_TEST_MATRICES)
@pytest.mark.parametrize("p", [257, 65537, 189812507])
def test_echelonize_rank_and_shape(oracle, name, p):
    A = oracle.load_sms(matrix_path(name), p)
    F = oracle.echelonize(A)
    U, qinv = F.U, F.qinv
    if A.n and A.m and A.n * A.m <= 400 * 400:
        assert U.n == numpy_rank(A.to_dense(), p)
    # tests/echelonize.c:33-52 (echelon_form_check)
    seen = set()
    for i in range(U.n):
        jj, xx = U.row(i)
        assert len(jj) > 0 and xx[0] == 1 and int(jj[0]) not in seen
        seen.add(int(jj[0]))
        assert qinv[jj[0]] == i
    # tests/echelonize.c:80-117 (deterministic_inclusion_test)
    for i in range(A.n):
        pat, x = oracle.solve_row(U, qinv, A, i)
        assert not any(x[j] != 0 and qinv[j] < 0 for j in pat)


@pytest.mark.parametrize("name", SMALL_SET)
def test_echelonize_modes_agree(oracle, name):
    """GPLU-only, dense-only and multi-round give the same rank."""
    p = 42013
    A = oracle.load_sms(matrix_path(name), p)
    ranks = []
    for max_round, thr in [(0, 2.0), (3, 0.05), (3, -1.0), (0, -1.0)]:
        o = oracle.default_opts()
        o.max_round = max_round
        o.sparsity_threshold = thr
        ranks.append(oracle.echelonize(A, o).U.n)
    assert len(set(ranks)) == 1


@pytest.mark.parametrize("name", SMALL_SET)
def test_rref_matches_reference(oracle, name):
    need_ref(oracle)
    p = 65537
    A = oracle.load_sms(matrix_path(name), p)
    F = oracle.echelonize(A)
    R, Rq = oracle.rref(F)
    R_r, Rq_r = oracle.ref_rref(F)
    assert np.array_equal(Rq, Rq_r)
    assert oracle.same_matrix(R, R_r)
