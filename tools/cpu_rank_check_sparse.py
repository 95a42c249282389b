#!/usr/bin/env python3
"""Independent CPU check of a rank for the stand-ins whose Schur complement stays SPARSE (mk15.b4: 2.2 M rows x 71,000
columns, 1.4-2.6e9 entries, rank of a few hundred; mk14.b5).  None of this repository's elimination code: structural pivots
and the rows of the Schur complement come from the COMPILED REFERENCE (oracle/_ref: spasm_pivots_extract_structural,
spasm_schur -- the sparse one, block of rows by block of rows), every block is folded at once into Z = H S mod p with a random
sparse c x n matrix H (every row of S is added, with random coefficients, to FOLD random rows of Z; c above the rank), and Z
is eliminated exactly in numpy (blocked, float64 products of residues below 2^16 summed over <= 2,048 terms stay below 2^53).
rowspace(Z) is inside rowspace(S): pivots + rank(Z) is a PROVEN lower bound of the rank, and the rank itself unless H is
unlucky (probability ~ 1/p per missing dimension).  Run end to end by this script, nothing by hand.
python tools/cpu_rank_check_sparse.py mk15.b4 [rows of H] [threads] [rows per block]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import oracle as orc          # noqa: E402
import workloads                           # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mk15.b4"
c = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
threads = int(sys.argv[3]) if len(sys.argv) > 3 else (os.cpu_count() or 1)
BLOCK = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
FOLD = 8
p = 42013
t_all = time.time()
n, m, ti, tj, tx = workloads._triplets_of(name)
if n < m:
    ti, tj, n, m = tj, ti, m, n
A = orc.compress(p, n, m, ti, tj, tx)
orc.ref_set_threads(threads)
t0 = time.time()
npiv, perm, F = orc.ref_pivots_extract_structural(A, orc.empty_fact(A.n, A.m, p))
rows = perm[npiv:]
Sm = A.m - F.U.n
print("%s: %d x %d, %d structural pivots (compiled reference, %d threads), %d rows left, %d non-pivotal columns (%.0f s)" %
      (name, A.n, A.m, npiv, threads, len(rows), Sm, time.time() - t0), flush=True)
# the non-pivotal columns, numbered in increasing order
nonpiv = np.flatnonzero(np.asarray(F.qinv) < 0)
assert len(nonpiv) == Sm
colmap = np.full(A.m, -1, np.int64)
colmap[nonpiv] = np.arange(Sm)
rng = np.random.default_rng(12345)
Z = np.zeros((c, Sm), np.float64)
t0 = time.time()
total_nnz = 0
for lo in range(0, len(rows), BLOCK):
    sub = rows[lo:lo + BLOCK]
    S, p_out = orc.ref_schur(A, sub, F, threads=threads)
    k = S.n
    cols = colmap[np.asarray(S.j[:S.p[k]], np.int64)]
    assert cols.min(initial=0) >= 0                                  # a Schur complement has no entry on a pivotal column
    vals = np.mod(np.asarray(S.x[:S.p[k]], np.int64), p).astype(np.float64)
    total_nnz += int(S.p[k])
    Sb = sp.csr_matrix((vals, cols, np.asarray(S.p[:k + 1], np.int64)), shape=(k, Sm))
    # H restricted to these rows: FOLD entries per row of S.  Z += H_b S_b in slices of 2,048 rows of S (products of two
    # residues, at most FOLD * 2,048 of them per entry of Z before the reduction: below 2^53)
    for r0 in range(0, k, 2048):
        r1 = min(k, r0 + 2048)
        kk = r1 - r0
        dst = rng.integers(0, c, size=(kk, FOLD)).reshape(-1)
        coef = rng.integers(1, p, size=(kk, FOLD)).astype(np.float64).reshape(-1)
        src = np.repeat(np.arange(kk), FOLD)
        H = sp.csr_matrix((coef, (dst, src)), shape=(c, kk))
        Z += (H @ Sb[r0:r1]).toarray()
        Z %= p
    print("  rows %d / %d folded, %d entries of S so far (%.0f s)" % (lo + k, len(rows), total_nnz, time.time() - t0), flush=True)
print("folded %d rows (%d entries of S) into %d combinations, %d per row (%.0f s); eliminating" % (len(rows), total_nnz, c, FOLD, time.time() - t0), flush=True)

# exact elimination of Z (c x Sm) by columns, blocked: panels of 128 columns; entries of Z are residues
t0 = time.time()
rank = 0
row = 0
PANEL = 128
for j0 in range(0, Sm, PANEL):
    j1 = min(Sm, j0 + PANEL)
    if row >= c:
        break
    P = Z[row:, j0:j1]
    if not P.any():
        continue
    for j in range(j0, j1):
        if row >= c:
            break
        col = Z[row:, j]
        nz = np.flatnonzero(col)
        if nz.size == 0:
            continue
        r = row + int(nz[0])
        if r != row:
            Z[[row, r]] = Z[[r, row]]
        inv = pow(int(Z[row, j]), p - 2, p)
        Z[row, j0:] = np.mod(Z[row, j0:] * inv, p)
        f = Z[row + 1:, j].copy()
        nzf = np.flatnonzero(f)
        if nzf.size:
            # only the columns of this panel now; the columns to the right get the whole panel's operations below
            Z[row + 1 + nzf, j0:] = np.mod(Z[row + 1 + nzf, j0:] - np.outer(f[nzf], Z[row, j0:]), p)
        row += 1
        rank += 1
print("%s: rank >= %d pivots + %d = %d (equal unless the random combinations are unlucky; %d combinations of the rows%s) (%.0f s; %.0f s in all)" %
      (name, npiv, rank, npiv + rank, c, "" if rank < c else ": NOT above the rank, raise them", time.time() - t0, time.time() - t_all), flush=True)
