#!/usr/bin/env python3
"""Randomised parity stress on a GPU box (not part of the test suite: minutes, not seconds).
Sparse Schur complements through every waves-per-group variant of the row-group kernel and dense RREFs of random
rank-deficient blocks, each compared with the oracle.  python tools/stress_gpu.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")          # (this script names kernel variants and debugging aids)
os.environ.setdefault("SPASM_HIP_VERBOSE", "0")
import numpy as np
import spasm_amd
from oracle import oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
t_end = time.time() + budget
rng = np.random.default_rng(int(time.time()))
primes = [3, 257, 8191, 42013, 44927, 44939, 46349, 65521, 4294967291]          # (44927: the largest prime with signed 16-bit entries)
cases = fails = 0


def as_product(A):
    return spasm_amd.Csr(A.n, A.m, A.p, A.j, A.x, A.prime)


while time.time() < t_end:
    p = int(rng.choice(primes))
    # ---- sparse Schur complement
    n, m, per_row = int(rng.integers(200, 6000)), int(rng.integers(100, 4000)), int(rng.integers(2, 7))
    ti = np.repeat(np.arange(n, dtype=np.int32), per_row)
    tj = rng.integers(0, m, size=n * per_row).astype(np.int32)
    tx = rng.integers(1, p, size=n * per_row).astype(np.int64)
    A = orc.compress(p, n, m, ti, tj, tx)
    npiv, perm, F = orc.pivots_extract_structural(A, orc.empty_fact(A.n, A.m, p))
    rows = perm[npiv:]
    if len(rows):
        want, p_out_want, _ = orc.schur(A, rows, F)
        for waves in ("1", "2", "4"):
            os.environ["SPASM_HIP_GROUP"] = "1"
            os.environ["SPASM_HIP_GROUP_WAVES"] = waves
            S, p_out = spasm_amd.schur(as_product(A), rows, spasm_amd.Fact(as_product(F.U), F.qinv))
            ok = orc.same_matrix(orc.CSR(S.n, S.m, S.p, S.j, S.x, p), want) and np.array_equal(np.asarray(p_out), np.asarray(p_out_want))
            cases += 1
            if not ok:
                fails += 1
                print("MISMATCH schur n=%d m=%d per_row=%d p=%d waves=%s" % (n, m, per_row, p, waves), flush=True)
        os.environ.pop("SPASM_HIP_GROUP", None)
        os.environ.pop("SPASM_HIP_GROUP_WAVES", None)
        # the back-substituted image: every workgroup shape, 16- and 32-bit entries, both ways of starting the rows and of
        # writing the result
        bs_env = {"SPASM_HIP_BACKSOLVE": "1", "SPASM_HIP_BS_SHAPE": str(int(rng.integers(0, 3))),
                  "SPASM_HIP_BS_SPARSE_INIT": str(int(rng.integers(0, 2))),
                  "SPASM_HIP_BS_DIRECT": str(int(rng.integers(0, 2))), "SPASM_HIP_BS_SIGNED": str(int(rng.integers(0, 2))),
                  "SPASM_HIP_BS_STAGED": str(int(rng.integers(0, 2))), "SPASM_HIP_STAGE_ROWS": str(int(rng.choice([0, 1, 37, 1000])))}
        os.environ.update(bs_env)
        S, p_out = spasm_amd.schur(as_product(A), rows, spasm_amd.Fact(as_product(F.U), F.qinv))
        ok = orc.same_matrix(orc.CSR(S.n, S.m, S.p, S.j, S.x, p), want) and np.array_equal(np.asarray(p_out), np.asarray(p_out_want))
        cases += 1
        if not ok:
            fails += 1
            print("MISMATCH backsolve n=%d m=%d per_row=%d p=%d %s" % (n, m, per_row, p, bs_env), flush=True)
        for key in bs_env:
            os.environ.pop(key, None)
        # the sparse image (round 4): forced, with a random segment size and a pool small enough to be outgrown now and then
        sp_env = {"SPASM_HIP_SPARSE_IMAGE": "1", "SPASM_HIP_BACKSOLVE": "0", "SPASM_HIP_SPARSE_IMAGE_PERSISTENT": str(int(rng.integers(0, 2))),
                  "SPASM_HIP_SPARSE_IMAGE_CHUNK": str(int(rng.choice([0, 4096, 100000]))),
                  # (round 5) 8-byte entries with 32-bit accumulators whatever the prime; structural masks of the segments on or off
                  "SPASM_HIP_SPARSE_IMAGE_WIDE": str(int(rng.integers(0, 2)))}
        os.environ.update(sp_env)
        S, p_out = spasm_amd.schur(as_product(A), rows, spasm_amd.Fact(as_product(F.U), F.qinv))
        ok = orc.same_matrix(orc.CSR(S.n, S.m, S.p, S.j, S.x, p), want) and np.array_equal(np.asarray(p_out), np.asarray(p_out_want))
        cases += 1
        if not ok:
            fails += 1
            print("MISMATCH sparse image n=%d m=%d per_row=%d p=%d %s" % (n, m, per_row, p, sp_env), flush=True)
        for key in sp_env:
            os.environ.pop(key, None)
    # ---- dense RREF of a rank-deficient block
    dn, dm = int(rng.integers(1, 1500)), int(rng.integers(1, 700))
    k = int(rng.integers(0, min(dn, dm) + 1))
    L = rng.integers(0, p, size=(dn, max(k, 1)), dtype=np.int64).astype(object)
    R = rng.integers(0, p, size=(max(k, 1), dm), dtype=np.int64).astype(object)
    M = np.array((L.dot(R)) % p, dtype=np.int64) if k > 0 else np.zeros((dn, dm), np.int64)
    M[:, : int(rng.integers(0, dm + 1)) // 3] = 0
    r, Rm, q = spasm_amd.ffpack_rref(p, M)
    r0, R0, q0 = orc.dense_rref(p, M)
    cases += 1
    if r != r0 or not np.array_equal(np.asarray(q)[:r], np.asarray(q0)[:r]) or not np.array_equal(np.asarray(Rm)[:r], np.asarray(R0)[:r]):
        fails += 1
        print("MISMATCH rref %dx%d rank %d p=%d (got rank %d)" % (dn, dm, r0, p, r), flush=True)
    # ---- dense RREF of blocks on which the tries of the panel steps fail (round 4: retired rows, tries that take what comes,
    # proofs and rollbacks): dependent columns among independent ones, runs of dependent rows, zero rows
    if cases % 5 == 0:
        dn, dm = int(rng.integers(600, 1800)), int(rng.integers(1100, 2400))
        every = int(rng.integers(2, 5))
        kk = int(rng.integers(50, min(dn, dm // every)))
        gen = rng.integers(0, p, size=(kk, dm), dtype=np.int64)
        for j in range(dm):
            if j % every != 0 and j >= every:
                gen[:, j] = (gen[:, j - (j % every)] * int(rng.integers(1, p)) + gen[:, (j - (j % every)) - every] * int(rng.integers(0, p))) % p
        L = rng.integers(0, p, size=(dn, kk), dtype=np.int64)
        if rng.integers(0, 2):
            for g0 in range(0, dn, 64):          # runs of 64 rows that only involve 30 of the kk directions
                keep = rng.choice(kk, size=min(kk, 30), replace=False)
                mask = np.zeros(kk, bool)
                mask[keep] = True
                L[g0:g0 + 64, ~mask] = 0
        M = np.array((L.astype(object).dot(gen.astype(object))) % p, dtype=np.int64)
        M[rng.integers(0, dn, size=dn // 4)] = 0
        r, Rm, q = spasm_amd.ffpack_rref(p, M)
        r0, R0, q0 = orc.dense_rref(p, M)
        cases += 1
        if r != r0 or not np.array_equal(np.asarray(q)[:r], np.asarray(q0)[:r]) or not np.array_equal(np.asarray(Rm)[:r], np.asarray(R0)[:r]):
            fails += 1
            print("MISMATCH rref (dependent columns) %dx%d rank %d p=%d every=%d (got rank %d)" % (dn, dm, r0, p, every, r), flush=True)
    # ---- the whole driver on a random sparse matrix: rank against the oracle's driver
    if cases % 7 == 0:
        en, em, eper = int(rng.integers(50, 1500)), int(rng.integers(50, 1500)), int(rng.integers(1, 6))
        ti = np.repeat(np.arange(en, dtype=np.int32), eper)
        tj = rng.integers(0, em, size=en * eper).astype(np.int32)
        tx = rng.integers(1, p, size=en * eper).astype(np.int64)
        E = orc.compress(p, en, em, ti, tj, tx)
        want_rank = orc.echelonize(E).U.n
        for finish in ("1", "0"):            # device-resident dense finish / the host loop
            os.environ["SPASM_HIP_DEVICE_FINISH"] = finish
            os.environ["SPASM_HIP_RREF_LOOKAHEAD"] = str(int(rng.integers(0, 2)))          # dense RREF: tries one panel ahead / one after the other
            o = spasm_amd.default_opts()
            if rng.integers(0, 2):
                o.sparsity_threshold = -1.0          # dense finish straight away
                o.dense_block_size = int(rng.integers(16, 400))
                if rng.integers(0, 2):
                    o.tall_and_skinny_ratio = 0.0    # ... in low-rank mode
            got = spasm_amd.echelonize(as_product(E), o).U.n
            cases += 1
            if got != want_rank:
                fails += 1
                print("MISMATCH echelonize %dx%d per_row=%d p=%d finish=%s: rank %d, oracle %d" % (en, em, eper, p, finish, got, want_rank), flush=True)
        os.environ.pop("SPASM_HIP_DEVICE_FINISH", None)
        os.environ.pop("SPASM_HIP_RREF_LOOKAHEAD", None)
    # ---- matrices large enough for the pivot search on the device (>= 20,000 rows): the rank must not depend on where the
    # search runs (device with the reached-bits in LDS / in HBM, host threads) nor on transposition
    if cases % 23 == 0:
        en, em, eper = int(rng.integers(20000, 60000)), int(rng.integers(8000, 70000)), int(rng.integers(2, 6))
        ti = np.repeat(np.arange(en, dtype=np.int32), eper)
        tj = rng.integers(0, em, size=en * eper).astype(np.int32)
        tx = rng.integers(1, p, size=en * eper).astype(np.int64)
        E = as_product(orc.compress(p, en, em, ti, tj, tx))
        ranks = {}
        # (round 5) the labelled search under random limits (gap between labels, length of a cascade, work list), with the
        # triangular check of the host on the ordering the labels gave, and the checks of the compact combinations
        os.environ.update({"SPASM_HIP_PIVOT_CHECK": "1", "SPASM_HIP_COMBINE_CHECK": "1", "SPASM_HIP_COMPACT_CHECK": "1"})
        for where, bits, labels in (("device", "", "1"), ("device", "", "0"), ("device", "global", "1"), ("host", "", "1"), ("host", "", "0")):
            os.environ["SPASM_HIP_PIVOT_SEARCH"] = where
            os.environ["SPASM_HIP_PIVOT_BITS"] = bits
            os.environ["SPASM_HIP_PIVOT_LABELS"] = labels
            os.environ["SPASM_HIP_PIVOT_GAP"] = str(int(rng.choice([1, 8, 64, 4096])))
            os.environ["SPASM_HIP_PIVOT_CASCADE"] = str(int(rng.choice([16, 512, 8192])))
            os.environ["SPASM_HIP_PIVOT_LABEL_FIFO"] = str(int(rng.choice([64, 4096, 32768])))
            os.environ["SPASM_HIP_PIVOT_SECOND_PASS_ALWAYS"] = str(int(rng.integers(0, 2)))
            ranks[where + bits + labels] = spasm_amd.echelonize(E).U.n
        for key in ("SPASM_HIP_PIVOT_SEARCH", "SPASM_HIP_PIVOT_BITS", "SPASM_HIP_PIVOT_LABELS", "SPASM_HIP_PIVOT_GAP", "SPASM_HIP_PIVOT_CASCADE",
                    "SPASM_HIP_PIVOT_LABEL_FIFO", "SPASM_HIP_PIVOT_SECOND_PASS_ALWAYS"):
            os.environ.pop(key, None)
        ranks["transpose"] = spasm_amd.echelonize(spasm_amd.transpose(E)).U.n
        for key in ("SPASM_HIP_PIVOT_CHECK", "SPASM_HIP_COMBINE_CHECK", "SPASM_HIP_COMPACT_CHECK"):
            os.environ.pop(key, None)
        cases += 1
        if len(set(ranks.values())) != 1:
            fails += 1
            print("MISMATCH ranks of a %dx%d matrix, %d per row, p=%d: %s" % (en, em, eper, p, ranks), flush=True)
print("stress: %d cases, %d mismatches" % (cases, fails))
sys.exit(1 if fails else 0)
