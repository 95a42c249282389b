#!/usr/bin/env python3
"""Independent CPU check of a rank for matrices whose Schur complement is too wide and of too high a rank for
tools/cpu_rank_check.py (ch8-8.b5: 292,000 rows x 104,000 columns of rank ~4,350: a row-by-row Gauss-Jordan in numpy would
take days).  None of this repository's elimination code: structural pivots and dense rows of the Schur complement come from
the COMPILED REFERENCE (oracle/_ref: spasm_pivots_extract_structural, spasm_schur_dense); the rows are then folded into
Z = H S mod p with a random c x n matrix H (c = 8,192 > rank), block by block (float64 products of residues < 2^16 summed over
2,048 terms stay below 2^53), Z is projected once more on the column side (W = Z G, 8,192 x 5,120) and W is eliminated
exactly (blocked, same bound).  rowspace(Z) is inside rowspace(S) and rank(Z G) <= rank(Z), so pivots + rank(W) is a PROVEN
lower bound of the rank; it is the rank itself unless H or G is unlucky (probability ~ 1/p per missing dimension).
Six hours on eight cores for ch8-8.b5, nearly all of it in the reference's spasm_schur_dense.
python tools/cpu_rank_check_projected.py ch8-8.b5 [rows of H] [columns of G]"""
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

np.save(os.environ.get("RANK_CHECK_Z", "/tmp/rank_check_Z_%s.npy" % name), Z)          # (six hours of folding on ch8-8.b5: kept)

# Second projection, on the column side: W = Z G with G a random Sm x w matrix (w = 5,120 > rank): rank(W) <= rank(Z) always,
# equal unless G is unlucky.  An exact elimination of the 8,192 x 104,768 matrix Z itself in numpy takes 160 s per panel of 256
# columns (409 panels: the first version of this script); W is 8,192 x 5,120.
w = int(sys.argv[3]) if len(sys.argv) > 3 else 5120
t0 = time.time()
W = np.zeros((c, w), np.float64)
CH = 2048                                   # 2,048 products of residues: below 2^53
for k0 in range(0, Sm, CH):
    k1 = min(Sm, k0 + CH)
    G = rng.integers(0, p, size=(k1 - k0, w)).astype(np.float64)
    W = np.mod(W + np.mod(Z[:, k0:k1] @ G, p), p)
print("projected onto %d random combinations of the columns (%.0f s)" % (w, time.time() - t0), flush=True)

# exact blocked elimination of W: panels of 128 columns, the rows below and the columns to the right updated by ONE product
# of rank <= 128 per panel (entries below p, 128 terms: far below 2^53)
Z = W
Sm = w
t0 = time.time()
rank = 0
row = 0                                   # rows [0, row) are finished pivot rows (not kept reduced: only the rank is asked for)
PANEL = 128
for j0 in range(0, Sm, PANEL):
    j1 = min(Sm, j0 + PANEL)
    if row >= Z.shape[0]:
        break
    P = Z[row:, j0:j1].copy()
    if not P.any():
        continue
    k = P.shape[0]
    perm = np.arange(k)
    piv_cols = []
    npv = 0
    for j in range(j1 - j0):
        cand = np.flatnonzero(P[npv:, j])
        if cand.size == 0:
            continue
        r = int(cand[0]) + npv
        t = npv
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
        npv += 1
    if npv == 0:
        continue
    # the same row operations on the columns to the right: with the rows in their new order, orig[:, piv_cols] = X U (U unit upper
    # triangular: the pivot rows of the panel in echelon form) gives the multipliers X of every row; the pivot rows are
    # Xp^-1 (old pivot rows), every other row loses X[row] times them
    Zr = Z[row:][perm]
    U = P[:npv][:, piv_cols]
    B = Zr[:, j0:j1][:, piv_cols].copy()
    X = np.zeros_like(B)
    for t in range(npv):
        X[:, t] = B[:, t]
        if t + 1 < npv:
            B[:, t + 1:] = np.mod(B[:, t + 1:] - np.outer(X[:, t], U[t, t + 1:]), p)
    Xp = X[:npv]
    R = Zr[:, j1:]
    ER = R[:npv].copy()
    for t in range(npv):
        inv = pow(int(Xp[t, t]), p - 2, p)
        ER[t] = np.mod(ER[t] * inv, p)
        if t + 1 < npv:
            ER[t + 1:] = np.mod(ER[t + 1:] - np.outer(Xp[t + 1:, t], ER[t]), p)
    R[:npv] = ER
    R[npv:] = np.mod(R[npv:] - np.mod(X[npv:] @ ER, p), p)
    Zr[:, j1:] = R
    Zr[:, j0:j1] = P
    Z[row:] = Zr
    row += npv
    rank += npv
print("%s: rank >= %d pivots + %d = %d (equal unless the random combinations are unlucky; %d combinations of the rows, %d of the columns%s) (%.0f s)" %
      (name, npiv, rank, npiv + rank, c, w, "" if rank < min(c, w) else ": NOT above the rank, raise them", time.time() - t0), flush=True)
