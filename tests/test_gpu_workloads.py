"""The BASELINE workloads at full size on the GPU, against the compiled reference (oracle/_ref) or the oracle.

mk13.b5 is regenerated from its definition (tools/workloads.py); the other BASELINE matrices run when their
.sms files are found under $SPASM_DATA (default tests/data/) and are reported as skipped otherwise.

What is compared (tools/rank.c:88-92 orientation, spasm_schur.c:61-193):
  * the round-0 Schur complement of EVERY non-pivotal row, computed by spasm_hip_dschur on the full batch, once per
    elimination path (back-substituted image; row-group kernel);
  * a deterministic sample of >= 2000 of its rows, spread over the whole batch (so over all row groups), entry for
    entry against the reference's spasm_schur on the same rows;
  * its total number of entries against the reference's on the whole batch (real reference only: the single-thread
    oracle would take minutes);
  * the rank through the drop-in tools/rank binary.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

import spasm_amd

sys.path.insert(0, os.path.join(ROOT, "tools"))
import workloads  # noqa: E402

pytestmark = pytest.mark.gpu

PRIME = 42013
# mk13.b4 is not a BASELINE config: it is the sibling round 1 benchmarked by mistake, kept because its Schur complement
# is SPARSE (4.4 %, 169 M entries) where mk13.b5's is dense -- the two matrices take the two elimination paths.
# ch7-8.b5 (chessboard complex, 141,120 x 141,120): a STAND-IN of the GL7d19 class, never a BASELINE config -- 49,000 non-
# pivotal columns (too wide for the back-substituted image: row-group kernel), Schur complement 18 % dense.
EXTRA = ["mk13.b4", "ch7-8.b5"]
NAMES = [c["name"] for c in workloads.CONFIGS] + EXTRA
# mk13.b4: the CPU oracle's single-thread orc_echelonize (243 s, round 1).  mk13.b5: every path combination of this
# library agrees on 134211; the CPU oracle had not finished it within the round.
# ch7-8.b5 / ch8-8.b5: 92959 is the published rank of ch7-8.b5 (hpac table); 276031 was recomputed on the CPU with the compiled
# reference and numpy (tools/cpu_rank_check_projected.py: 271,552 structural pivots + 4,479, profiles/r04_cpu_rank_check_ch8-8.b5.log).
# mk14.b4: 272,862 + 321 by tools/cpu_rank_check.py (profiles/r04_cpu_rank_check_mk14.b4.log).
# mk15.b4 (2,837,835 x 675,675, 14.2 M entries: the at-scale stand-in): 604,176 structural pivots + 415 = 604,591 by
# tools/cpu_rank_check_sparse.py -- pivots and all 2,233,659 rows of the Schur complement (3.8e9 entries) from the COMPILED
# REFERENCE, folded into 2,048 random combinations and eliminated exactly in numpy, 2 h 21 min on six cores, run end to end by
# the script (profiles/r05_cpu_rank_check_mk15.b4.log).  mk14.b5 (945,945 x 945,945): what every flow of this library
# returns, equal to the rank of the transpose -- no independent value (its Schur complement, 22 % dense on 290,000 columns
# with a rank of 17,000, is out of reach of the reference's spasm_schur on this container: see DESIGN.md section 5).
RANKS = {"mk13.b5": 134211, "mk13.b4": 111463, "ch7-8.b5": 92959, "ch8-8.b5": 276031, "mk14.b4": 273183, "mk15.b4": 604591,
         "mk14.b5": 672762}


def _available(name):
    c = workloads.config(name)
    if c is None:
        return True                      # a generated sibling
    return c["generator"] is not None or workloads.find_data(c["file"]) is not None


def _device_rows(S, ks):
    """rows ks of a DeviceCsr as host (j, x) pairs."""
    Sp = S.p.cpu().numpy()
    out = []
    for k in ks:
        lo, hi = int(Sp[k]), int(Sp[k + 1])
        out.append((S.j[lo:hi].cpu().numpy(), S.x[lo:hi].cpu().numpy()))
    return Sp, out


def _full_schur(A, rows, F, env, pool=None):
    import torch
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        dA = spasm_amd.DeviceCsr.from_host(A)
        dF = spasm_amd.DeviceFact(F)
        drows = torch.from_numpy(np.ascontiguousarray(rows, np.int32)).cuda()
        pool = pool or 4 * A.nnz + (1 << 24)
        while True:
            W = spasm_amd.SchurWorkspace(len(rows), A.m, pool)
            S, st = spasm_amd.dschur(dA, drows, dF, W, fetch=True)
            if st.status == 0:
                break
            W.close()
            pool *= 2
        return S, st, W, dF
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("path", ["backsolve", "sparse_image", "row_groups", "row_groups_pull"])
def test_round0_schur_of_baseline_workload(oracle, name, path):
    if not _available(name):
        pytest.skip("%s: data file absent (SPASM_DATA=%s)" % (name, workloads.data_dir()))
    A, rows, F, source = workloads.round0(name, PRIME)
    env = {"SPASM_HIP_BACKSOLVE": "1", "SPASM_HIP_SPARSE_IMAGE": "0"} if path == "backsolve" else \
          {"SPASM_HIP_SPARSE_IMAGE": "1"} if path == "sparse_image" else \
          {"SPASM_HIP_BACKSOLVE": "0", "SPASM_HIP_GROUP": "1", "SPASM_HIP_SPARSE_IMAGE": "0"}
    if path == "row_groups_pull":
        if name != "mk13.b4":
            pytest.skip("the pull variant is an experiment: checked at full size on the sparse sibling only")
        env["SPASM_HIP_PULL"] = "1"
    S, st, W, dF = _full_schur(A, rows, F, env)
    if path == "backsolve" and not st.used_backsolve:
        pytest.skip("%s: the factor is not eligible for the back-substituted image" % name)
    if path == "sparse_image" and not st.used_sparse_image:
        pytest.skip("%s: no sparse image (R is not sparse, or the prime is beyond the signed 16-bit arithmetic)" % name)
    assert st.rows == len(rows) and (st.used_backsolve == 1) == (path == "backsolve") and (st.used_sparse_image == 1) == (path == "sparse_image")
    Ao = oracle.CSR(A.n, A.m, A.p, A.j, A.x, PRIME)
    Fo = oracle.Fact(oracle.CSR(F.U.n, F.U.m, F.U.p, F.U.j, F.U.x, PRIME), F.qinv)
    # >= 2000 rows spread evenly over the batch, first and last included
    count = min(len(rows), 2048)
    ks = np.unique(np.linspace(0, len(rows) - 1, count).astype(np.int64))
    if oracle.ref_available():
        want, p_out = oracle.ref_schur(Ao, rows[ks], Fo, threads=spasm_amd.usable_cpus())
        order = {int(r): t for t, r in enumerate(p_out)}          # the reference emits rows in thread-arrival order
    else:
        want, p_out, _ = oracle.schur(Ao, rows[ks], Fo)
        order = {int(r): t for t, r in enumerate(p_out)}
    Sp, got = _device_rows(S, ks)
    for k, (gj, gx) in zip(ks, got):
        wj, wx = want.row(order[int(rows[k])])
        o = np.argsort(wj)
        assert np.array_equal(gj, wj[o]) and np.array_equal(np.asarray(gx, np.int64) % PRIME, np.asarray(wx[o], np.int64) % PRIME), \
            "row %d of the batch (row %d of A) differs" % (k, rows[k])
        assert np.all(np.diff(gj) > 0)
    assert int(Sp[-1]) == st.nnz
    if oracle.ref_available() and len(rows) <= 400000:
        full, _ = oracle.ref_schur(Ao, rows, Fo, threads=spasm_amd.usable_cpus())
        assert full.nnz == st.nnz
    W.close()
    dF.close()


def test_round0_schur_at_the_size_of_GL7d19_against_the_compiled_reference(oracle):
    """mk15.b4 (2,837,835 x 675,675: the at-scale stand-in): its round-0 Schur complement -- 2.2 M rows on 71,000 columns, 1.4-2.6e9
    entries, 18 segments, ~2,600 elimination levels: the largest thing the sparse image ever produces -- against the compiled
    reference's spasm_schur on 2,048 rows spread over the whole batch, entry for entry (the pivots come from the device search:
    the sequential host search of round0() would take minutes on this matrix; S is checked against the reference on the SAME
    factor, whatever pivots it holds)."""
    name = "mk15.b4"
    A, rows, F, source = workloads.round0(name, PRIME, threads=0)
    S, st, W, dF = _full_schur(A, rows, F, {}, pool=3 << 30)
    assert st.status == 0 and st.used_sparse_image == 1 and st.rows == len(rows)
    Ao = oracle.CSR(A.n, A.m, A.p, A.j, A.x, PRIME)
    Fo = oracle.Fact(oracle.CSR(F.U.n, F.U.m, F.U.p, F.U.j, F.U.x, PRIME), F.qinv)
    ks = np.unique(np.linspace(0, len(rows) - 1, 2048).astype(np.int64))
    if oracle.ref_available():
        want, p_out = oracle.ref_schur(Ao, rows[ks], Fo, threads=spasm_amd.usable_cpus())
    else:
        want, p_out, _ = oracle.schur(Ao, rows[ks], Fo)
    order = {int(r): t for t, r in enumerate(p_out)}
    Sp, got = _device_rows(S, ks)
    total = 0
    for k, (gj, gx) in zip(ks, got):
        wj, wx = want.row(order[int(rows[k])])
        o = np.argsort(wj)
        assert np.array_equal(gj, wj[o]) and np.array_equal(np.asarray(gx, np.int64) % PRIME, np.asarray(wx[o], np.int64) % PRIME), \
            "row %d of the batch (row %d of A) differs" % (k, rows[k])
        assert np.all(np.diff(gj) > 0)
        total += len(gj)
    assert total > 0 and int(Sp[-1]) == st.nnz
    W.close()
    dF.close()


def test_sparse_image_beyond_16_bit_primes_at_scale(oracle):
    """mk14.b4 mod 65537 (673,000 rows on 42,000 columns: a GL7d19-class factor with a prime beyond the signed 16-bit arithmetic,
    which until round 5 had no image at all -- minutes in the row-group kernel): the sparse image with 32-bit entries takes it by
    itself, and its Schur complement is the one the row-by-row kernels compute, entry for entry; 64 rows against the compiled
    reference."""
    import torch
    prime = 65537
    A, rows, F, source = workloads.round0("mk14.b4", prime, threads=0)
    S2, st2, W2, dF2 = _full_schur(A, rows, F, {}, pool=1 << 30)
    assert st2.used_sparse_image == 1 and st2.status == 0
    S0, st0, W0, dF0 = _full_schur(A, rows, F, {"SPASM_HIP_BACKSOLVE": "0", "SPASM_HIP_SPARSE_IMAGE": "0"}, pool=1 << 30)
    assert st0.used_sparse_image == 0 and st0.used_backsolve == 0 and st0.status == 0 and st0.nnz == st2.nnz
    assert torch.equal(S2.p, S0.p)
    assert torch.equal(S2.j[:st2.nnz], S0.j[:st0.nnz]) and torch.equal(S2.x[:st2.nnz], S0.x[:st0.nnz])
    Ao = oracle.CSR(A.n, A.m, A.p, A.j, A.x, prime)
    Fo = oracle.Fact(oracle.CSR(F.U.n, F.U.m, F.U.p, F.U.j, F.U.x, prime), F.qinv)
    ks = np.unique(np.linspace(0, len(rows) - 1, 64).astype(np.int64))
    if oracle.ref_available():
        want, p_out = oracle.ref_schur(Ao, rows[ks], Fo, threads=spasm_amd.usable_cpus())
    else:
        want, p_out, _ = oracle.schur(Ao, rows[ks], Fo)
    order = {int(r): t for t, r in enumerate(p_out)}
    Sp, got = _device_rows(S2, ks)
    for k, (gj, gx) in zip(ks, got):
        wj, wx = want.row(order[int(rows[k])])
        o = np.argsort(wj)
        assert np.array_equal(gj, wj[o]) and np.array_equal(np.asarray(gx, np.int64) % prime, np.asarray(wx[o], np.int64) % prime), \
            "row %d of the batch (row %d of A) differs" % (k, rows[k])
    for W, dF in ((W2, dF2), (W0, dF0)):
        W.close()
        dF.close()


def test_default_path_follows_the_cost_model(oracle, monkeypatch):
    """no path forced (fitted on tools/sweep_cost.py): a FULL batch of mk13.b5 (Sm = 4,952) and of mk13.b4
    (Sm = 23,958: round 2's rule sent it row by row and lost) builds the back-substituted image; a one-off batch of 4,096
    rows does not pay for the build and goes row by row -- unless R is already there; a factor whose rows take few
    eliminations (hint: 50 per row instead of the ~5,000 measured) stays row by row even on the full batch."""
    import torch
    # (the choice between the dense image and the row-by-row kernels: the sparse image is kept out of it here -- it has a
    #  test of its own below)
    monkeypatch.setenv("SPASM_HIP_SPARSE_IMAGE", "0")
    for name in ("mk13.b5", "mk13.b4"):
        A, rows, F, _ = workloads.round0(name, PRIME)
        dA = spasm_amd.DeviceCsr.from_host(A)
        drows = torch.from_numpy(np.ascontiguousarray(rows, np.int32)).cuda()
        sub = drows[:4096].contiguous()
        dF = spasm_amd.DeviceFact(F)
        W = spasm_amd.SchurWorkspace(len(rows), A.m, 1 << 30)
        S, st = spasm_amd.dschur(dA, sub, dF, W, fetch=False)
        assert st.status == 0 and st.used_backsolve == 0, (name, "one-off sub-batch")
        # (a batch under 1,024 rows -- the driver's density sample -- builds R at once only when its rows are short)
        tiny = drows[:100].contiguous()
        S, st = spasm_amd.dschur(dA, tiny, dF, W, fetch=False)
        assert st.status == 0 and st.used_backsolve == (1 if A.m - F.U.n <= 8192 else 0), (name, "density sample")
        dF.forget()
        S, st = spasm_amd.dschur(dA, drows, dF, W, fetch=False)
        assert st.status == 0 and st.used_backsolve == 1, (name, "full batch")
        full_nnz = st.nnz
        S, st = spasm_amd.dschur(dA, sub, dF, W, fetch=False)
        assert st.status == 0 and st.used_backsolve == 1, (name, "sub-batch once R is there")
        dF.forget()
        dF.hint_eliminations(50.0)
        S, st = spasm_amd.dschur(dA, drows, dF, W, fetch=False)
        assert st.status == 0 and st.used_backsolve == 0 and st.nnz == full_nnz, (name, "few eliminations per row")
        W.close()
        dF.close()


def test_default_path_takes_the_sparse_image_where_the_schur_complement_stays_sparse(oracle):
    """mk13.b4 (23,958 non-pivotal columns, S 4.4 % dense): a full batch with no path forced and no density hint builds the
    sparse image (R 3.4 % full) and S is the one the dense image gives; with the hint that S is dense (50 %) the dense image
    runs; a density sample of 100 rows never builds anything.  mk13.b5 (4,952 columns, S 72 % dense) has no plan for it."""
    import torch
    A, rows, F, _ = workloads.round0("mk13.b4", PRIME)
    dA = spasm_amd.DeviceCsr.from_host(A)
    drows = torch.from_numpy(np.ascontiguousarray(rows, np.int32)).cuda()
    dF = spasm_amd.DeviceFact(F)
    W = spasm_amd.SchurWorkspace(len(rows), A.m, 1 << 30)
    S, st = spasm_amd.dschur(dA, drows[:100].contiguous(), dF, W, fetch=False)
    assert st.status == 0 and st.used_sparse_image == 0 and st.used_backsolve == 0
    S, st = spasm_amd.dschur(dA, drows, dF, W, fetch=True)
    assert st.status == 0 and st.used_sparse_image == 1 and st.sparse_image_built == 1
    fill = st.sparse_image_nnz / (float(F.U.n) * (A.m - F.U.n))
    assert 0.005 < fill < 0.10
    dF.forget()
    dF.hint_density(0.5)
    S2, st2 = spasm_amd.dschur(dA, drows, dF, W, fetch=True)
    assert st2.status == 0 and st2.used_sparse_image == 0 and st2.used_backsolve == 1 and st2.nnz == st.nnz
    assert torch.equal(S.p, S2.p) and torch.equal(S.j, S2.j) and torch.equal(S.x, S2.x)
    W.close()
    dF.close()
    A, rows, F, _ = workloads.round0("mk13.b5", PRIME)
    dF = spasm_amd.DeviceFact(F)
    W = spasm_amd.SchurWorkspace(len(rows), A.m, 1 << 30)
    S, st = spasm_amd.dschur(spasm_amd.DeviceCsr.from_host(A), torch.from_numpy(np.ascontiguousarray(rows, np.int32)).cuda(), dF, W, fetch=False)
    assert st.status == 0 and st.used_sparse_image == 0 and st.used_backsolve == 1
    W.close()
    dF.close()


@pytest.mark.parametrize("name", NAMES)
def test_rank_tool_on_baseline_workload(name, tmp_path):
    """tools/rank with the options of the BASELINE config; rank against the CPU value where one is recorded."""
    if not _available(name):
        pytest.skip("%s: data file absent (SPASM_DATA=%s)" % (name, workloads.data_dir()))
    c = workloads.config(name) or {"file": name + ".sms", "rank_args": workloads.STAND_INS.get(name, {}).get("rank_args", [])}
    path = workloads.find_data(c["file"])
    if path is None:
        A, _ = workloads.load_matrix(name, PRIME, tall=False)
        path = str(tmp_path / (name + ".sms"))
        workloads.save_sms(A, path)
    elif path.endswith(".gz"):
        path, _ = workloads._open_plain(path)
    tool = os.path.join(ROOT, "tools", "rank")
    if not os.path.exists(tool):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tools")], check=True)
    env = dict(os.environ, SPASM_HIP_VERBOSE="0")
    out = subprocess.run([tool, "--matrix", path, "--modulus", str(PRIME)] + c["rank_args"], capture_output=True, text=True,
                         env=env, timeout=3000)
    assert out.returncode == 0, out.stderr[-2000:]
    rank = int(out.stdout.strip().split()[-1])
    if name in RANKS:
        assert rank == RANKS[name]
    else:
        assert rank > 0


def test_image_on_rows_wider_than_the_lds_at_full_size(oracle):
    """mk14.b4, the GL7d19-class stand-in: 673,000 rows to reduce on 42,000 non-pivotal columns -- rows of S too wide for the
    LDS of a wave, produced in segments by the apply kernel of the back-substituted image (a 23 GB image).  The whole
    Schur complement, 1.06e9 entries, must be the one the row-by-row kernels compute, entry for entry; sampled rows are
    checked against the compiled reference."""
    import torch
    A, rows, F, source = workloads.round0("mk14.b4", PRIME, threads=0)
    S1, st1, W1, dF1 = _full_schur(A, rows, F, {"SPASM_HIP_BACKSOLVE": "1", "SPASM_HIP_SPARSE_IMAGE": "0"})
    assert st1.used_backsolve == 1 and st1.status == 0
    S0, st0, W0, dF0 = _full_schur(A, rows, F, {"SPASM_HIP_BACKSOLVE": "0", "SPASM_HIP_SPARSE_IMAGE": "0"})
    assert st0.used_backsolve == 0 and st0.status == 0 and st0.nnz == st1.nnz
    assert torch.equal(S1.p, S0.p)
    assert torch.equal(S1.j[:st1.nnz], S0.j[:st0.nnz]) and torch.equal(S1.x[:st1.nnz], S0.x[:st0.nnz])
    # ... and the sparse image (what the library takes by itself on this factor: R is 1.6-1.9 % full), entry for entry
    del S0
    W0.close()
    S2, st2, W2, dF2 = _full_schur(A, rows, F, {})
    assert st2.used_sparse_image == 1 and st2.status == 0 and st2.nnz == st1.nnz
    assert torch.equal(S1.p, S2.p)
    assert torch.equal(S1.j[:st1.nnz], S2.j[:st2.nnz]) and torch.equal(S1.x[:st1.nnz], S2.x[:st2.nnz])
    W0, dF0 = W2, dF0
    dF2.close()
    Ao = oracle.CSR(A.n, A.m, A.p, A.j, A.x, PRIME)
    Fo = oracle.Fact(oracle.CSR(F.U.n, F.U.m, F.U.p, F.U.j, F.U.x, PRIME), F.qinv)
    ks = np.unique(np.linspace(0, len(rows) - 1, 64).astype(np.int64))
    if oracle.ref_available():
        want, p_out = oracle.ref_schur(Ao, rows[ks], Fo, threads=spasm_amd.usable_cpus())
    else:
        want, p_out, _ = oracle.schur(Ao, rows[ks], Fo)
    order = {int(r): t for t, r in enumerate(p_out)}
    Sp, got = _device_rows(S1, ks)
    for k, (gj, gx) in zip(ks, got):
        wj, wx = want.row(order[int(rows[k])])
        o = np.argsort(wj)
        assert np.array_equal(gj, wj[o]) and np.array_equal(np.asarray(gx, np.int64) % PRIME, np.asarray(wx[o], np.int64) % PRIME), \
            "row %d of the batch (row %d of A) differs" % (k, rows[k])
    for W, dF in ((W1, dF1), (W0, dF0)):
        W.close()
        dF.close()


def test_generated_stand_ins_have_the_published_shapes():
    """the chessboard complexes are closed-form: sizes against the published ones (tools/workloads.py)"""
    for name, info in workloads.STAND_INS.items():
        if name == "ch8-8.b5":
            continue                      # (3.4 M entries through the Python generator: covered by the rank test below)
        n, m, ti, tj, tx = workloads._triplets_of(name)
        assert (n, m) == info["shape"] and len(ti) == info["nnz"]


@pytest.mark.parametrize("name,threshold,min_sparse_rounds", [("ch7-8.b5", 0.01, 0), ("ch8-8.b5", 0.01, 0), ("mk14.b4", 0.05, 1), ("mk15.b4", 0.05, 1),
                                                              ("mk14.b5", 0.05, 0)])
def test_multi_round_stand_in(name, threshold, min_sparse_rounds, monkeypatch):
    """spasm_hip_echelonize end to end on the GL7d19-class stand-ins, with the options of the GL7d19 config for the chessboard
    complexes (--dense-threshold 0.01: their first Schur complement is 18 % dense, so the call is pivot search + the dense
    finish on 49,000 / 104,000 columns -- no back-substituted image) and the defaults for mk14.b4, whose first Schur
    complement (673,000 x 42,000, 3.7 % dense, 1.06e9 entries) IS computed sparse before the low-rank finish -- through the
    sparse image since round 4 --, for mk15.b4 (2,837,835 x 675,675, 14.2 M entries: the size of GL7d19; its Schur complement
    is 2.2 M x 71,000 with 1.4-1.9e9 entries, beyond any dense image) and for mk14.b5 (945,945 x 945,945: more columns than
    the pivot search has LDS bits for, a first Schur complement 22 % dense on 290,000 columns).  The rank must
    be the same on every call (the threaded pivot search picks different pivots each time) and equal to the recorded one;
    the factor must be a valid echelon form of the right shape."""
    A, _ = workloads.load_matrix(name, PRIME)
    o = spasm_amd.default_opts()
    o.sparsity_threshold = threshold
    ranks = []
    # (the combinations of all rows that end the low-rank finish are formed twice -- block sums in LDS, and one atomic per
    #  term -- and the library dies if the sums differ mod p: mk14.b4 is where the first kernel runs on 1e9 entries)
    monkeypatch.setenv("SPASM_HIP_EXPERIMENT", "1")          # (the two checks below are switches outside the supported list)
    monkeypatch.setenv("SPASM_HIP_COMBINE_CHECK", "1")
    # (... and the combinations of rows of a Schur complement are formed on its non-pivotal columns only: once more on all
    #  columns, and the library dies if the dense rows differ)
    monkeypatch.setenv("SPASM_HIP_COMPACT_CHECK", "1")
    for _ in range(2):
        F = spasm_amd.echelonize(A, o)
        prof = spasm_amd.echelonize_profile()
        ranks.append(F.U.n)
        assert prof["sparse_rounds"] >= min_sparse_rounds
        # echelon form: one pivot per row, on distinct columns, first in its row with value 1
        piv = F.U.j[F.U.p[:-1]]
        assert len(np.unique(piv)) == F.U.n and np.all(F.U.x[F.U.p[:-1]] == 1)
        assert np.array_equal(F.qinv[piv], np.arange(F.U.n))
    assert ranks[0] == ranks[1] == RANKS[name]
    # rank(A) = rank(A^T): the transposed matrix is wide, its pivots, Schur complements and finishing blocks are all different
    # (not for mk15.b4: its transpose leaves a remainder 15 % dense on 2.2 M columns, which no device finish holds -- the host
    #  loops return the same rank, 604,591, after 730 s: gpurun_out of round 4; tools/rank transposes such a matrix first)
    if name == "mk15.b4":
        return
    Ft = spasm_amd.echelonize(spasm_amd.transpose(A), o)
    assert Ft.U.n == RANKS[name]


@pytest.mark.parametrize("name,threshold,min_sparse_rounds", [("mk13.b4", 0.05, 2), ("ch7-8.b5", 0.01, 1), ("mk13.b5", 0.05, 2), ("mk14.b4", 0.05, 2)])
def test_flow_without_the_greedy_pivot_search(name, threshold, min_sparse_rounds, monkeypatch):
    """the option of BASELINE configs[4] (M0,6-D9, "greedy pivot search disabled": tools/echelonize.c:36 -> spasm_pivots.c:315;
    the data file cannot be fetched, the FLOW runs on every generated matrix): with Faugere-Lachartre pivots only the first
    Schur complement is larger and fills in, the driver runs several sparse rounds (spasm_echelonize.c:525-565) -- each with a
    factor image planned for the U of that round -- before the dense finish.  Same rank as every other flow, twice, and a
    valid echelon form."""
    A, _ = workloads.load_matrix(name, PRIME)
    o = spasm_amd.default_opts()
    o.enable_greedy_pivot_search = 0
    o.sparsity_threshold = threshold
    # (mk13.b5 ends on the round limit with a Schur complement of 152,000 rows on 17,000 of 135,000 columns: the combinations of
    #  the low-rank finish are formed on those columns only -- and, under this switch, once more on all columns: the library
    #  dies if the dense rows differ)
    monkeypatch.setenv("SPASM_HIP_EXPERIMENT", "1")
    monkeypatch.setenv("SPASM_HIP_COMPACT_CHECK", "1")
    for _ in range(2):
        F = spasm_amd.echelonize(A, o)
        prof = spasm_amd.echelonize_profile()
        events = spasm_amd.echelonize_counters()
        assert F.U.n == RANKS[name]
        assert prof["sparse_rounds"] >= min_sparse_rounds
        assert events["pivot_visits"] == 0 and events["pivot_rows_with_a_pivot"] == 0          # the greedy search never ran
        piv = F.U.j[F.U.p[:-1]]
        assert len(np.unique(piv)) == F.U.n and np.all(F.U.x[F.U.p[:-1]] == 1)
        assert np.array_equal(F.qinv[piv], np.arange(F.U.n))


@pytest.mark.gpu
def test_block_cache_serves_the_second_call_of_the_same_work(monkeypatch):
    """the cache of device blocks: a driver call that does exactly what the call before it did -- same matrix, the sequential
    host search (SPASM_HIP_THREADS=1: the same pivot set every time), hence the same buffers of the same sizes -- must not take
    a single large block fresh from the device (spasm_hip_echelonize_counters: block_cache_misses).  Round 5 found requests of
    17-32 MB rounded up to the very threshold from which a block counts as large: booked as large, parked as large, and never
    found again by the next small request -- five fresh blocks in every call."""
    monkeypatch.setenv("SPASM_HIP_THREADS", "1")
    A, _ = workloads.load_matrix("mk13.b5")
    o = spasm_amd.default_opts()
    ranks, misses = [], []
    for _ in range(3):
        F = spasm_amd.echelonize(A, o)
        ev = spasm_amd.echelonize_counters()
        ranks.append(F.U.n)
        misses.append((ev["block_cache_misses"], ev["block_cache_miss_bytes"]))
    assert ranks == [134211] * 3
    assert misses[1] == (0, 0) and misses[2] == (0, 0), misses          # (the first call of a process fills the cache, the next ones live off it)


@pytest.mark.gpu
@pytest.mark.parametrize("keep_gb", ["0", "4"])
def test_driver_calls_with_a_small_block_cache(keep_gb, monkeypatch):
    """SPASM_HIP_KEEP_GB at its edges: nothing (0) or next to nothing (4 GB) of the cache of device blocks survives a driver call --
    what a process that shares its device sets.  mk14.b4 parks tens of GB by default; with the cache emptied after every call the
    next call takes its blocks fresh (slower: the first touch of new device memory) and must give the same rank, call after call."""
    monkeypatch.setenv("SPASM_HIP_KEEP_GB", keep_gb)
    A, _ = workloads.load_matrix("mk14.b4")
    ranks, misses = [], []
    for _ in range(3):
        F = spasm_amd.echelonize(A)
        ranks.append(F.U.n)
        misses.append(spasm_amd.echelonize_counters()["block_cache_miss_bytes"])
    assert ranks == [RANKS["mk14.b4"]] * 3
    if keep_gb == "0":
        assert misses[1] > 0 and misses[2] > 0, misses          # (nothing was kept: the large blocks come from the device again)
    spasm_amd.release_cached_memory()


@pytest.mark.gpu
@pytest.mark.parametrize("mb", [0.5, 3, 17, 20, 31.9, 32, 33, 100, 300, 1500])
def test_block_cache_gives_a_block_back_to_the_next_request_of_its_size(mb):
    """a buffer taken, given back, and asked for again must come from the cache whatever its size -- in particular 17-32 MB,
    which the power-of-two classes of the small blocks round to 32 MB, the threshold of the large ones (the defect above)"""
    import ctypes as C
    L = C.CDLL(spasm_amd.LIB_PATH)
    L.spasm_hip_debug_block_cache_roundtrip.argtypes = [C.c_size_t]
    L.spasm_hip_debug_block_cache_roundtrip.restype = C.c_int
    assert L.spasm_hip_debug_block_cache_roundtrip(int(mb * (1 << 20))) == 1
