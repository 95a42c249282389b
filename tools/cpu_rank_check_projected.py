#!/usr/bin/env python3
"""Independent CPU check of a rank for matrices whose Schur complement is too wide and of too high a rank for
tools/cpu_rank_check.py (ch8-8.b5: 292,000 rows x 104,000 columns of rank ~4,350: a row-by-row Gauss-Jordan in numpy would
take days).  None of this repository's elimination code: structural pivots and dense rows of the Schur complement come from
the COMPILED REFERENCE (oracle/_ref: spasm_pivots_extract_structural, spasm_schur_dense); the rows are then folded into
Z = H S mod p with a random c x n matrix H (c = 8,192 > rank), block by block (float64 products of residues < 2^16 summed over
2,048 terms stay below 2^53), and Z is eliminated exactly (blocked Gauss-Jordan, same bound).  rowspace(Z) is inside
rowspace(S), so pivots + rank(Z) is a PROVEN lower bound of the rank; it is the rank itself unless H is unlucky
(probability ~ 1/p per missing dimension).   python tools/cpu_rank_check_projected.py ch8-8.b5 [c]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import oracle as orc          # noqa: E402
import workloads                           # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ch8-8.b5"
c = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
p = 42013
n, m, ti, tj, tx = workloads._triplets_of(name)
if n < m:
    ti, tj, n, m = tj, ti, m, n
A = orc.compress(p, n, m, ti, tj, tx)
t0 = time.time()
npiv, perm, F = orc.ref_pivots_extract_structural(A, orc.empty_fact(A.n, A.m, p))
rows = perm[npiv:]
Sm = A.m - F.U.n
print("%s: %d x %d, %d structural pivots (compiled reference), %d rows left, %d non-pivotal columns (%.0f s)" %
      (name, A.n, A.m, npiv, len(rows), Sm, time.time() - t0), flush=True)
rng = np.random.default_rng(12345)
Z = np.zeros((c, Sm), np.float64)
BLOCK = 2048
FOLD = 8                                  # every row of S is added, with random coefficients, to FOLD random rows of Z
t0 = time.time()
for lo in range(0, len(rows), BLOCK):
    sub = rows[lo:lo + BLOCK]
    S, q, p_out = orc.ref_schur_dense(A, sub, F)
    Y = np.mod(np.asarray(S, np.int64), p).astype(np.float64)
    for f in range(FOLD):
        dst = rng.integers(0, c, size=Y.shape[0])
        coef = rng.integers(1, p, size=Y.shape[0]).astype(np.float64)
        order = np.argsort(dst, kind="stable")
        dsts, starts = np.unique(dst[order], return_index=True)
        # (rows of Y that go to the same row of Z are added up first: products < 2^32, a few terms)
        contrib = np.add.reduceat(np.mod(Y[order] * coef[order, None], p), starts, axis=0)
        Z[dsts] = np.mod(Z[dsts] + contrib, p)
    if (lo // BLOCK) % 8 == 0:
        print("  rows %d / %d folded (%.0f s)" % (lo + len(sub), len(rows), time.time() - t0), flush=True)
print("folded %d rows into %d combinations, %d per row (%.0f s); eliminating" % (len(rows), c, FOLD, time.time() - t0), flush=True)

# exact blocked elimination of Z: panels of 256 columns, the rows below and the columns to the right updated by ONE product
# of rank <= 256 per panel (entries below p, 256 terms: far below 2^53)
t0 = time.time()
rank = 0
row = 0                                   # rows [0, row) are finished pivot rows (not kept reduced: only the rank is asked for)
PANEL = 256
for j0 in range(0, Sm, PANEL):
    j1 = min(Sm, j0 + PANEL)
    P = Z[row:, j0:j1].copy()
    if not P.any():
        continue
    k = P.shape[0]
    # Gaussian elimination of the panel by rows; the pivot rows are swapped to the top of the remaining rows
    perm = np.arange(k)
    piv_cols = []
    npiv_here = 0
    for j in range(j1 - j0):
        cand = np.flatnonzero(P[npiv_here:, j])
        if cand.size == 0:
            continue
        r = int(cand[0]) + npiv_here
        t = npiv_here
        if r != t:
            P[[t, r]] = P[[r, t]]
            perm[[t, r]] = perm[[r, t]]
        inv = pow(int(P[t, j]), p - 2, p)
        P[t] = np.mod(P[t] * inv, p)
        f = P[t + 1:, j].copy()
        nz = np.flatnonzero(f)
        if nz.size:
            P[t + 1 + nz] = np.mod(P[t + 1 + nz] - np.outer(f[nz], P[t]), p)
        piv_cols.append(j)
        npiv_here += 1
    if npiv_here == 0:
        continue
    # the same row operations on the columns to the right.  With Q = the rows in their new order, the pivot rows are
    # W = L^-1 Q[:np] (unit-lower-triangular system solved through the panel itself) and every other row loses its multiples of
    # them; both from the ORIGINAL panel: M = multipliers such that P_below_original = M @ (pivot rows of the panel)
    Zr = Z[row:][perm]                        # (a copy, rows in the new order)
    orig = Zr[:, j0:j1]
    # pivot rows of the panel in echelon form (unit pivots): E (np x 256); solve X E[:, piv_cols] = orig[:, piv_cols] for the
    # multipliers X of every row (triangular: E[:, piv_cols] is unit upper triangular)
    E = P[:npiv_here]
    U = E[:, piv_cols]                        # np x np, unit upper triangular
    B = orig[:, piv_cols].copy()              # k x np
    X = np.zeros_like(B)
    for t in range(npiv_here):                # forward substitution column by column (np <= 256 steps of a k-vector update)
        X[:, t] = B[:, t]
        if t + 1 < npiv_here:
            B[:, t + 1:] = np.mod(B[:, t + 1:] - np.outer(X[:, t], U[t, t + 1:]), p)
    # rows: new pivot rows = X[:np] relates original pivot rows to echelon rows: orig[:np] = X[:np] @ E  ->  E = X[:np]^-1 orig[:np]
    # apply to the right part: ER = X[:np]^-1 @ Zr[:np, right]; then rows below: Zr[np:, right] -= X[np:] @ ER
    Xp = X[:npiv_here]                        # np x np, lower triangular with non-zero diagonal
    for c0 in range(j1, Sm, 8192):
        R = Zr[:, c0:c0 + 8192]
        ER = R[:npiv_here].copy()
        for t in range(npiv_here):            # forward substitution with the lower-triangular Xp
            inv = pow(int(Xp[t, t]), p - 2, p)
            ER[t] = np.mod(ER[t] * inv, p)
            if t + 1 < npiv_here:
                ER[t + 1:] = np.mod(ER[t + 1:] - np.outer(Xp[t + 1:, t], ER[t]), p)
        R[:npiv_here] = ER
        R[npiv_here:] = np.mod(R[npiv_here:] - np.mod(X[npiv_here:] @ ER, p), p)
        Zr[:, c0:c0 + 8192] = R
    Zr[:, j0:j1] = P
    Z[row:] = Zr
    row += npiv_here
    rank += npiv_here
    if (j0 // PANEL) % 16 == 0:
        print("  columns %d / %d: rank so far %d (%.0f s)" % (j1, Sm, rank, time.time() - t0), flush=True)
    if row >= Z.shape[0]:
        break
print("%s: rank >= %d pivots + %d = %d (equal unless the random combinations are unlucky; c = %d%s) (%.0f s)" %
      (name, npiv, rank, npiv + rank, c, "" if rank < c else ": c is NOT above the rank, raise it", time.time() - t0), flush=True)
