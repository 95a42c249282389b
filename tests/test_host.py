"""CPU-side checks of the product library: ABI completeness, host logic, level schedule."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ALL_TEST_MATRICES, ROOT, matrix_path

import spasm_amd
from spasm_amd.matrix import view_csr


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "spasm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(spasm_hip_[a-zA-Z0-9_]+)\s*\(", hdr)))
    assert len(names) > 40
    L = C.CDLL(spasm_amd.LIB_PATH)
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_no_product_dependency_on_oracle():
    """the shipped library and package never reference oracle/ (checker only)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "spasm_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no CPU fallback", ""), (dirpath, f)


def test_compute_entry_points_fail_loudly_without_gpu(oracle):
    if spasm_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    A = spasm_amd.load(matrix_path("small.sms"), 257)
    with pytest.raises(RuntimeError):
        spasm_amd.schur(A, np.arange(A.n, dtype=np.int32), spasm_amd.empty_fact(A.m, 257))


@pytest.mark.parametrize("name", ALL_TEST_MATRICES)
def test_load_and_compress_match_oracle(oracle, name):
    for p in (257, 4294967291):
        A = spasm_amd.load(matrix_path(name), p)
        B = oracle.load_sms(matrix_path(name), p)
        assert (A.n, A.m) == (B.n, B.m)
        assert np.array_equal(A.p, B.p) and np.array_equal(A.j, B.j) and np.array_equal(A.x, B.x)


def test_compress_duplicates_and_cancellation(oracle):
    p = 97
    ti = np.array([0, 0, 0, 1, 1, 1, 2, 0], np.int32)
    tj = np.array([1, 1, 2, 0, 0, 3, 2, 1], np.int32)
    tx = np.array([5, 92, 7, 3, -3, 11, 97, 1], np.int64)   # (0,1): 5+92+1 = 98 = 1; (1,0) cancels; (2,2) = 0
    A = spasm_amd.compress(p, 3, 4, ti, tj, tx)
    B = oracle.compress(p, 3, 4, ti, tj, tx)
    assert np.array_equal(A.p, B.p) and np.array_equal(A.j, B.j) and np.array_equal(A.x, B.x)
    assert A.nnz == 3


def test_transpose_matches_oracle(oracle):
    A = spasm_amd.load(matrix_path("rectangular_h.sms"), 65537)
    T = spasm_amd.transpose(A)
    To = oracle.transpose(oracle.CSR(A.n, A.m, A.p, A.j, A.x, A.prime))
    assert np.array_equal(T.p, To.p) and np.array_equal(T.j, To.j) and np.array_equal(T.x, To.x)


@pytest.mark.parametrize("name", ALL_TEST_MATRICES)
@pytest.mark.parametrize("p", [257, 189812507])
def test_pivot_search_matches_oracle(oracle, name, p):
    A = spasm_amd.load(matrix_path(name), p)
    npiv, perm, F = spasm_amd.pivots_extract_structural(A, spasm_amd.empty_fact(A.m, p))
    Ao = oracle.load_sms(matrix_path(name), p)
    npiv_o, perm_o, F_o = oracle.pivots_extract_structural(Ao, oracle.empty_fact(Ao.n, Ao.m, p))
    assert npiv == npiv_o
    assert np.array_equal(perm, perm_o)
    assert np.array_equal(F.qinv, F_o.qinv)
    assert np.array_equal(F.U.p, F_o.U.p) and np.array_equal(F.U.j, F_o.U.j) and np.array_equal(F.U.x, F_o.U.x)


def test_threaded_steps_of_the_pivot_search_on_a_large_matrix(oracle, monkeypatch):
    """beyond 200,000 rows the Faugere-Lachartre steps run on threads (leftmost entries by atomic minima, the column step with
    its first pass and a filter for its second on threads) and so do the lengths and the rows of U (round 5): the pivots, the
    permutation and U must still be the oracle's, entry for entry.  210,000 rows of 1-3 entries whose leftmost columns fill a
    third of the 260,000 columns, so that the column step has pivots to find; the row-order greedy search behind them
    (SPASM_HIP_THREADS=1, no labels: the reference's pivot set) is cheap on this shape."""
    monkeypatch.setenv("SPASM_HIP_THREADS", "1")
    monkeypatch.setenv("SPASM_HIP_PIVOT_LABELS", "0")
    monkeypatch.setenv("SPASM_HIP_PIVOT_SEARCH", "host")
    rng = np.random.default_rng(7)
    p, n, m = 42013, 210000, 260000
    lens = rng.integers(1, 4, size=n)
    ti = np.repeat(np.arange(n, dtype=np.int32), lens)
    tj = rng.integers(m // 3, m, size=len(ti)).astype(np.int32)
    tj[np.cumsum(lens) - lens] = rng.integers(0, m // 3, size=n)
    tx = rng.integers(1, p, size=len(ti)).astype(np.int64)
    A = oracle.compress(p, n, m, ti, tj, tx)
    npiv_o, perm_o, F_o = oracle.pivots_extract_structural(A, oracle.empty_fact(A.n, A.m, p))
    Ap = spasm_amd.Csr(A.n, A.m, A.p, A.j, A.x, p)
    npiv, perm, F = spasm_amd.pivots_extract_structural(Ap, spasm_amd.empty_fact(A.m, p))
    assert npiv == npiv_o and npiv > len(np.unique(tj[np.cumsum(lens) - lens]))          # (more than the leftmost entries alone give)
    assert np.array_equal(perm, perm_o)
    assert np.array_equal(F.qinv, F_o.qinv)
    assert np.array_equal(F.U.p, F_o.U.p) and np.array_equal(F.U.j, F_o.U.j) and np.array_equal(F.U.x, F_o.U.x)
    # ... and the plan of this factor (rows gathered by threads beyond 20,000 pivots): a label per row, the pivot column of a
    # row carries the row's label, every other pivotal column of the row belongs to a strictly later level, the rest lies
    # behind the pivots
    nlev, label, lvl_end_row, lab, rpad = _plan(F)
    U, qinv = F.U, np.asarray(F.qinv)
    r = U.n
    assert nlev >= 1 and len(np.unique(label)) == r and label.min() >= 0 and label.max() < rpad
    Up, Uj = np.asarray(U.p), np.asarray(U.j)
    row_of = np.repeat(np.arange(r), np.diff(Up))
    first = np.zeros(len(Uj), bool)
    first[Up[:-1]] = True
    assert np.array_equal(lab[Uj[first]], label)
    rest_lab, rest_row = lab[Uj[~first]], row_of[~first]
    pivotal = qinv[Uj[~first]] >= 0
    assert np.all(rest_lab[pivotal] >= lvl_end_row[rest_row[pivotal]])
    assert np.all(rest_lab[~pivotal] >= rpad)

    # ... and with levels OFFERED to the plan, as the device pivot search offers its depth labels (round 5): valid ones are taken
    # (checked entry by entry on threads first), wrong ones are refused and the plan computes its own -- either way the schedule
    # that comes back must pass the same checks
    def check(plan):
        nlev2, label2, lvl_end2, lab2, rpad2 = plan
        assert nlev2 >= 1 and len(np.unique(label2)) == r and label2.min() >= 0 and label2.max() < rpad2
        assert np.array_equal(lab2[Uj[first]], label2)
        rl = lab2[Uj[~first]]
        assert np.all(rl[pivotal] >= lvl_end2[rest_row[pivotal]]) and np.all(rl[~pivotal] >= rpad2)
        return nlev2

    level_of_row = np.searchsorted(np.unique(lvl_end_row), lvl_end_row)          # 0 = the level eliminated first
    L = C.CDLL(spasm_amd.LIB_PATH)
    u = view_csr(F.U)

    def offer(heights):
        h = np.ascontiguousarray(heights, np.int32)
        L.spasm_hip_debug_level_hint(C.byref(u), h.ctypes.data_as(C.POINTER(C.c_int)))
        lab_out, lvl_out, lab_cols, info = np.zeros(r, np.int32), np.zeros(r, np.int32), np.zeros(F.U.m, np.int32), np.zeros(2, np.int32)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        q = np.ascontiguousarray(F.qinv, np.int32)
        n_levels = L.spasm_hip_debug_plan(C.byref(u), ip(q), ip(lab_out), ip(lvl_out), ip(lab_cols), ip(info))
        return n_levels, lab_out, lvl_out, lab_cols, int(info[0])

    top = int(level_of_row.max())
    assert check(offer(3 * (top - level_of_row))) == top + 1               # (valid, with gaps: taken, the gaps closed)
    assert check(offer(np.zeros(r, np.int32))) == nlev                      # (every dependency violated: refused)
    assert check(offer(level_of_row)) == nlev                               # (upside down: refused)


def _plan(F):
    L = C.CDLL(spasm_amd.LIB_PATH)
    u = view_csr(F.U)
    r, m = F.U.n, F.U.m
    label = np.zeros(max(r, 1), np.int32)
    lvl_end = np.zeros(max(r, 1), np.int32)
    lab = np.zeros(max(m, 1), np.int32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    q = np.ascontiguousarray(F.qinv, np.int32)
    info = np.zeros(2, np.int32)
    nlev = L.spasm_hip_debug_plan(C.byref(u), ip(q), ip(label), ip(lvl_end), ip(lab), ip(info))
    return nlev, label[:r], lvl_end[:r], lab[:m], int(info[0])


def _multi_round_fact(oracle, name, p):
    """a factor with rows from two rounds (old rows gain pivotal columns later)."""
    A = oracle.load_sms(matrix_path(name), p)
    F = oracle.empty_fact(A.n, A.m, p)
    npiv, perm, F = oracle.pivots_extract_structural(A, F)
    S, p_out, _ = oracle.schur(A, perm[npiv:], F)
    if S.nnz:
        npiv2, perm2, F = oracle.pivots_extract_structural(S, F)
        return A, S, perm2[npiv2:], F
    return A, S, np.zeros(0, np.int32), F


@pytest.mark.parametrize("name", ALL_TEST_MATRICES)
def test_level_schedule_is_valid_and_eliminates_correctly(oracle, name):
    """plan_factor's labels: every pivotal column met in a row of U belongs to a strictly later
    level; simulating the level-by-level elimination (what the kernels do) reproduces the oracle."""
    p = 65537
    A, S, rows2, F = _multi_round_fact(oracle, name, p)
    nlev, label, lvl_end_row, lab, rpad = _plan(F)
    U, qinv = F.U, F.qinv
    r = U.n
    assert len(set(label.tolist())) == r and (r == 0 or (label.min() >= 0 and label.max() < rpad))
    assert rpad % 32 == 0 and rpad <= r + 32 * max(nlev, 0)
    lvl_end = np.zeros(max(rpad, 1), np.int64)          # by label
    lvl_end[label] = lvl_end_row
    for k in range(r):
        jj, _ = U.row(k)
        assert lab[jj[0]] == label[k]
        for j in jj[1:]:
            if qinv[j] >= 0:
                assert lab[j] >= lvl_end[label[k]]
            else:
                assert lab[j] >= rpad
    r_real, r = r, rpad                                  # below, "r" is the size of the pivot label space
    # simulate on the second-round rows of S
    if len(rows2) == 0:
        return
    kof = np.zeros(max(r, 1), np.int64)
    kof[label] = np.arange(r_real)
    want, _, _ = oracle.schur(S, rows2[:40], F)
    for t, i in enumerate(rows2[:40]):
        x = {}
        jj, xx = S.row(int(i))
        for j, v in zip(jj.tolist(), xx.tolist()):
            x[int(lab[j])] = v % p
        while True:
            pend = sorted(c for c in x if c < r and x[c] != -1)
            if not pend:
                break
            lend = lvl_end[pend[0]]
            for c in [c for c in pend if c < lend]:
                v = x[c]
                x[c] = -1                       # processed marker
                if v == 0:
                    continue
                uj, ux = U.row(int(kof[c]))
                for j, u in zip(uj[1:].tolist(), ux[1:].tolist()):
                    cc = int(lab[j])
                    assert x.get(cc, 0) != -1
                    x[cc] = (x.get(cc, 0) - v * u) % p
        got = sorted((c, v) for c, v in x.items() if c >= r and v not in (0, -1))
        wj, wx = want.row(t)
        ref = sorted((int(lab[j]), int(v) % p) for j, v in zip(wj.tolist(), wx.tolist()))
        assert got == ref


def test_prng_matches_reference_golden_vector():
    """tests/Expected/prng of the reference (SHA-256 counter mode, rejection sampling)."""
    L = C.CDLL(spasm_amd.LIB_PATH)
    L.spasm_hip_debug_prng.argtypes = [C.c_int64, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_int32)]
    lines = open(matrix_path("../Expected/prng")).read().strip().split("\n")
    cases = [(257, 0, 0), (257, 0, 1), (257, 1, 0), (257, 1, 1), (65537, 0xdead00000000beef, 0)]
    for line, (p, seed, seq) in zip(lines, cases):
        want = [int(t) for t in line.split("out=")[1].split(",")]
        out = (C.c_int32 * 10)()
        L.spasm_hip_debug_prng(p, seed, seq, 10, out)
        assert list(out) == want


def test_row_group_kernels_keep_the_accumulation_registers_to_the_prefetch_ring(tmp_path):
    """schur_group_kernel parks prefetched lines in AGPRs that the compiler must not use for anything else
    (csrc/schur_kernels.hip, "software-managed vmcnt").  Disassemble the gfx950 code object and check: no
    v_accvgpr_write, no AGPR above the reserved range, no scratch (a spill could land in an AGPR), and the only
    instructions that touch AGPRs are the ring's loads and v_accvgpr_read."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import check_isa
    assert os.path.exists(check_isa.OBJDUMP), "llvm-objdump is part of the build (spasm_amd/csrc/Makefile runs the same check)"
    n, problems = check_isa.check(spasm_amd.LIB_PATH)
    assert n >= 12 and not problems, problems[:10]


def test_bitonic_network_of_the_row_regrouping_sorts():
    """the compare-exchange schedule of launch_regroup_rows (csrc/schur_kernels.hip: global steps for strides >= 2048,
    one in-LDS kernel per merge stage for the shorter ones), replayed here on random 64-bit keys."""
    LOCAL = 2048
    rng = np.random.default_rng(5)
    for npad in (1, 2, 64, 2048, 8192):
        keys = rng.integers(0, 1 << 62, size=npad, dtype=np.int64)

        def step(lo, hi, size, stride):          # bitonic_step_kernel / one pass of bitonic_local_kernel on [lo, hi)
            t = np.arange(lo, hi)
            t = t[(t & stride) == 0]
            partner = t | stride
            asc = (t & size) == 0
            x, y = keys[t].copy(), keys[partner].copy()
            swap = (x > y) == asc
            keys[t] = np.where(swap, y, x)
            keys[partner] = np.where(swap, x, y)

        size = 2
        while size <= npad:
            stride = size >> 1
            while stride >= LOCAL:
                step(0, npad, size, stride)
                stride >>= 1
            for base in range(0, npad, LOCAL):            # bitonic_local_kernel: one workgroup per 2048 keys
                s = stride
                while s > 0:
                    step(base, min(base + LOCAL, npad), size, s)
                    s >>= 1
            size <<= 1
        assert np.all(np.diff(keys) >= 0), npad


# --------------------------------------------------------------------------
# drop-in: the reference's own tool sources, unmodified, bound to the HIP library
# --------------------------------------------------------------------------
REF_TOOLS = "/root/reference/tools"
DROPIN = {name: os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", name)
          for name in ("ref_echelonize_shim", "ref_rank_facade", "ref_kernel_facade")}


def _undefined(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--undefined-only", path], check=True, capture_output=True, text=True).stdout
    return {line.split()[-1] for line in out.splitlines() if "spasm" in line}


def _defined(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    return {line.split()[-1] for line in out.splitlines() if "spasm" in line}


@pytest.mark.skipif(not os.path.isdir(REF_TOOLS), reason="the reference tree is not on this machine")
def test_reference_tools_compile_and_link_against_the_hip_library():
    """tools/echelonize.c through the -include shim (every spasm_* call becomes its spasm_hip_ twin: the prototypes of
    include/spasm_hip.h must agree with the reference's spasm.h or this does not compile), tools/rank.c and
    tools/kernel.c through the facade library (hot path from the GPU library, the rest from the reference's)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "oracle"), "dropin"], check=True, capture_output=True)
    for path in DROPIN.values():
        assert os.path.exists(path), path
    # shim build: nothing un-prefixed is left, and every symbol it wants is exported by libspasm_hip.so
    want = _undefined(DROPIN["ref_echelonize_shim"])
    assert want and all(s.startswith("spasm_hip_") for s in want), want
    assert want <= _defined(spasm_amd.LIB_PATH)
    # facade build: the hot-path symbols can only come from the facade (the reference library here is built without
    # spasm_echelonize.c / spasm_ffpack.cpp), the facade forwards to libspasm_hip.so
    facade = os.path.join(os.path.dirname(spasm_amd.LIB_PATH), "libspasm_hip_facade.so")
    provided = _defined(facade)
    for tool in ("ref_rank_facade", "ref_kernel_facade"):
        need = _undefined(DROPIN[tool])
        assert "spasm_echelonize" in need and "spasm_echelonize" in provided
        ref_lib = _defined(os.path.join(os.path.dirname(DROPIN[tool]), "libspasm_ref.so"))
        assert "spasm_echelonize" not in ref_lib
        assert need <= (provided | ref_lib), need - (provided | ref_lib)
    assert _undefined(facade) <= _defined(spasm_amd.LIB_PATH)


@pytest.mark.skipif(not os.path.exists(DROPIN["ref_echelonize_shim"]), reason="drop-in programs not built")
def test_reference_tool_bound_to_the_hip_library_dies_loudly_without_a_gpu():
    import subprocess
    if spasm_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with open(matrix_path("small.sms")) as f:
        out = subprocess.run([DROPIN["ref_echelonize_shim"]], stdin=f, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "no HIP device" in out.stderr


def test_fp32_reduction_of_the_signed_16_bit_path_stays_inside_its_slack():
    """backsolve.hip (SgnDev): an entry is reduced by q = rint(float(t) * float(1 / p)), r = t - q p; the kernels rely on
    |r| <= B = p/2 + p/64 + 1 for every |t| <= 4 B^2 + B < 2^31 (four products of such values on top of one).  Replayed in
    numpy float32 on the extreme sums, on values around the multiples of p/2 (where the quotient is least certain) and on
    random sums, for the primes at both ends of the eligible range."""
    rng = np.random.default_rng(5)
    for p in (3, 5, 257, 8191, 32003, 42013, 44927):
        B = p // 2 + p // 64 + 1
        T = 4 * B * B + B
        assert B <= 32767 and T <= 2 ** 31 - 1                       # sgn_eligible()
        invp = np.float32(1.0) / np.float32(p)
        k = np.arange(0, T // p + 2, max(1, (T // p) // 200000), dtype=np.int64)
        near_half = np.concatenate([k * p + p // 2 + d for d in (-1, 0, 1, 2)])
        t = np.concatenate([np.array([T, T - 1, -T, -T + 1, 0, 1, -1], np.int64), near_half, -near_half,
                            rng.integers(-T, T + 1, size=400000)])
        t = t[np.abs(t) <= T]
        q = np.rint(t.astype(np.float32) * invp).astype(np.int64)        # (numpy rounds half to even, as v_rndne_f32 does)
        r = t - q * p
        assert np.all(np.abs(r) <= B), (p, int(np.abs(r).max()), B)
        assert np.all((r - t) % p == 0)
    # the next prime is out: four-term sums would not fit 32 bits
    Bn = 44939 // 2 + 44939 // 64 + 1
    assert 4 * Bn * Bn + Bn > 2 ** 31 - 1


def test_vectorised_matching_complexes_equal_the_recursive_enumeration():
    """tools/workloads.py enumerates the matchings level by level with numpy (mk15.b4 -- 2.8 M rows, the at-scale stand-in --
    in seconds); the triplets must be those of the recursive enumeration, and the sizes the published ones"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import workloads
    for nv, K in ((7, 1), (9, 1), (10, 2), (11, 3), (12, 3)):
        a = workloads.mk_boundary(nv, K)
        b = workloads.mk_boundary_reference(nv, K)
        assert a[0] == b[0] and a[1] == b[1]
        for x, y in zip(a[2:], b[2:]):
            assert np.array_equal(x, y)
    n, m, ti, tj, tx = workloads.mk_boundary(9, 1)
    assert (n, m, len(ti)) == (378, 36, 756)
    n, m, ti, tj, tx = workloads.mk_boundary(12, 3)
    assert (n, m, len(ti)) == (51975, 13860, 207900)


def test_generated_workloads_are_built_like_spasm_compress():
    """tools/workloads.py builds the CSR of a generated matrix with numpy (five million add_entry calls through ctypes take
    longer than the elimination): same arrays as spasm_hip_compress on the same triplets, in both orientations; and the
    chessboard generator has the published shape on a small member of the family."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import workloads
    for name in ("mk10.b3", "ch6-6.b3"):
        n, m, ti, tj, tx = workloads._triplets_of(name)
        for a, b, N, M in ((ti, tj, n, m), (tj, ti, m, n)):
            want = spasm_amd.compress(42013, N, M, a, b, tx)
            got = workloads._csr_of_triplets(42013, N, M, a, b, tx)
            assert np.array_equal(want.p, got.p) and np.array_equal(want.j, got.j) and np.array_equal(want.x, got.x)
    n, m, ti, tj, tx = workloads._triplets_of("ch6-6.b3")          # 4-rook / 3-rook placements on a 6 x 6 board
    assert (n, m, len(ti)) == (15 * 15 * 24, 20 * 20 * 6, 15 * 15 * 24 * 4)


def test_usable_cpus_is_positive_and_within_the_hardware():
    assert 1 <= spasm_amd.usable_cpus() <= (os.cpu_count() or 1)


def _write_mtx(A, path, comments=True):
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate integer general\n")
        if comments:
            f.write("% written by tests/test_host.py\n%\n")
        f.write("%d %d %d\n" % (A.n, A.m, A.nnz))
        for i in range(A.n):
            for px in range(A.p[i], A.p[i + 1]):
                f.write("%d %d %d\n" % (i + 1, A.j[px] + 1, A.x[px]))


@pytest.mark.parametrize("name", ["small.sms", "rectangular_h.sms", "singular.sms", "medium.sms"])
@pytest.mark.parametrize("p", [257, 42013])
def test_matrix_market_files_load_like_the_reference(oracle, name, p, tmp_path):
    """spasm_triplet_load reads MatrixMarket coordinate/integer/general files as well as SMS (spasm_io.c:76-97, 127-139): the
    same matrix written as .mtx (with comment lines) loads to the same CSR as its .sms, and to what the compiled reference
    makes of the very same .mtx file."""
    A = spasm_amd.load(matrix_path(name), p)
    path = str(tmp_path / (name.replace(".sms", "") + ".mtx"))
    _write_mtx(A, path)
    B = spasm_amd.load(path, p)
    assert (A.n, A.m) == (B.n, B.m)
    assert np.array_equal(A.p, B.p) and np.array_equal(A.j, B.j) and np.array_equal(A.x, B.x)
    T = spasm_amd.load(path, p, transpose_if_wide=True)
    assert T.n >= T.m
    if oracle.ref_available():
        R = oracle.ref_load(path, p)
        assert (R.n, R.m) == (B.n, B.m)
        assert np.array_equal(R.p, B.p) and np.array_equal(R.j, B.j) and np.array_equal(R.x, B.x)


def test_matrix_market_header_is_validated(tmp_path):
    """the reference refuses anything but `matrix coordinate integer general` (spasm_io.c:31-53): so does the loader
    (it dies: checked in a child process)."""
    import subprocess
    import sys
    bad = tmp_path / "bad.mtx"
    bad.write_text("%%MatrixMarket matrix coordinate real general\n2 2 1\n1 1 1\n")
    code = "import spasm_amd; spasm_amd.load(%r, 257)" % str(bad)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode != 0 and "unsupported MatrixMarket data type" in out.stderr
    end = tmp_path / "end.mtx"
    end.write_text("%%MatrixMarket matrix coordinate integer general\n2 2 2\n1 1 1\n0 0 0\n")
    code = "import spasm_amd; spasm_amd.load(%r, 257)" % str(end)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode != 0 and "SMS end marker" in out.stderr


def test_bench_line_is_compact_and_carries_roofline_and_cpu_baseline():
    """the driver parses the LAST line of bench.py's stdout from a tail it keeps: the formatter must cut a recorded full
    result (round 5's 38 KB object, which the driver could not parse) down to a line under 12 KB that still holds the
    contract keys, roofline.frac and cpu_baseline.value"""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import bench_format
    full = json.loads(open(os.path.join(root, "profiles", "r05_bench_line.json")).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 30000
    # grow it the way a later round might: the line must still fit
    full["cpu_baseline"]["rank_time"] = {"seconds": 61.2, "projected": True, "measured_s": 14.8, "pivots_s": 0.9, "cores": 16,
                                         "what": "x" * 300}
    full["stand_ins"] = full["stand_ins"] * 3
    line = bench_format.line(full)
    assert "\n" not in line
    assert len(line) < bench_format.LINE_LIMIT == 12288
    got = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "data", "config", "roofline",
                "cpu_baseline", "factor_image_ms", "rows_per_s_cold", "higher_is_better", "scaling", "vs_baseline"):
        assert key in got, key
    assert got["value"] == full["value"] and got["ms_per_step"] == full["ms_per_step"]
    assert abs(got["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-5
    assert got["roofline"]["bound"] == "hbm" and got["roofline"]["kernels"]
    assert got["cpu_baseline"]["value"] > 0 and got["cpu_baseline"]["rank_time"]["seconds"] == 61.2
    assert "workload" in got["config"] and all(len(v) <= 100 for v in got["config"].values() if isinstance(v, str))
    # an error object (data file absent) is a valid line too
    err = json.loads(bench_format.line({"metric": "m", "value": None, "unit": "rows/s", "n_gpus": 1, "data": "absent",
                                        "config": {"workload": "GL7d19"}, "error": "no such file"}))
    assert err["value"] is None and err["error"] == "no such file"


def test_bench_prints_its_line_last_on_stdout_and_nothing_else(tmp_path):
    """bench.py sends everything else that reaches descriptor 1 to stderr: run its emit path in a child process whose C library
    also writes to stdout, and look at what the pipe holds"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import ctypes, json, os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
sys.argv = ["bench.py", "--workload", "no_such_config_file", "--gpus", "1"]
import types
torch = types.ModuleType("torch")
class _Cuda:
    def set_device(self, d): ctypes.CDLL(None).puts(b"a library prints to stdout")
torch.cuda = _Cuda(); torch.device = lambda *a: None
sys.modules["torch"] = torch
import workloads
def _absent(*a, **k): raise FileNotFoundError("GL7d19.sms is absent")
workloads.round0 = _absent
import bench
bench.ROOT = %r
bench.main()
''' % (root, root, str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    got = json.loads(lines[-1])
    assert got["data"] == "absent" and "absent" in got["error"]
    assert "a library prints to stdout" in r.stderr
    assert json.loads(open(os.path.join(str(tmp_path), "bench_full.json")).read())["error"]
