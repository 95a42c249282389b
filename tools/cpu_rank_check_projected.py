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
t0 = time.time()
for lo in range(0, len(rows), BLOCK):
    sub = rows[lo:lo + BLOCK]
    S, q, p_out = orc.ref_schur_dense(A, sub, F)
    Y = np.mod(np.asarray(S, np.int64), p).astype(np.float64)
    H = rng.integers(0, p, size=(c, Y.shape[0])).astype(np.float64)
    for j0 in range(0, Sm, 8192):          # (tiles keep the temporaries small; 2,048 terms of < 2^32 each: exact)
        Z[:, j0:j0 + 8192] = np.mod(Z[:, j0:j0 + 8192] + H @ Y[:, j0:j0 + 8192], p)
    if (lo // BLOCK) % 8 == 0:
        print("  rows %d / %d folded (%.0f s)" % (lo + len(sub), len(rows), time.time() - t0), flush=True)
print("folded %d rows into %d combinations (%.0f s); eliminating" % (len(rows), c, time.time() - t0), flush=True)

# exact blocked elimination of Z: panels of 256 columns
t0 = time.time()
rank = 0
row = 0                                   # rows [0, row) are finished pivot rows
PANEL = 256
for j0 in range(0, Sm, PANEL):
    j1 = min(Sm, j0 + PANEL)
    P = Z[row:, j0:j1]
    if not P.any():
        continue
    # Gauss-Jordan of the panel, recording the row operations as a matrix T (applied to the rest of Z by one product per panel)
    k = Z.shape[0] - row
    T = np.eye(k)
    P = P.copy()
    piv_rows = []
    for j in range(j1 - j0):
        cand = np.flatnonzero(P[len(piv_rows):, j]) + len(piv_rows)
        if cand.size == 0:
            continue
        r = int(cand[0])
        t = len(piv_rows)
        if r != t:
            P[[t, r]] = P[[r, t]]
            T[[t, r]] = T[[r, t]]
        inv = pow(int(P[t, j]), p - 2, p)
        P[t] = np.mod(P[t] * inv, p)
        T[t] = np.mod(T[t] * inv, p)
        f = P[:, j].copy()
        f[t] = 0
        nz = np.flatnonzero(f)
        if nz.size:
            P[nz] = np.mod(P[nz] - np.outer(f[nz], P[t]), p)
            T[nz] = np.mod(T[nz] - np.outer(f[nz], T[t]), p)
        piv_rows.append(t)
    if not piv_rows:
        continue
    # apply T to the columns to the right (entries of T and Z below p, k <= 8192 terms: below 2^53)
    for c0 in range(j1, Sm, 8192):
        Z[row:, c0:c0 + 8192] = np.mod(T @ Z[row:, c0:c0 + 8192], p)
    Z[row:, j0:j1] = P
    row += len(piv_rows)
    rank += len(piv_rows)
    if (j0 // PANEL) % 32 == 0:
        print("  columns %d / %d: rank so far %d (%.0f s)" % (j1, Sm, rank, time.time() - t0), flush=True)
    if row >= Z.shape[0]:
        break
print("%s: rank >= %d pivots + %d = %d (equal unless the random combinations are unlucky; c = %d%s) (%.0f s)" %
      (name, npiv, rank, npiv + rank, c, "" if rank < c else ": c is NOT above the rank, raise it", time.time() - t0), flush=True)
