#!/usr/bin/env python3
"""Independent CPU check of a rank (no GPU, none of this repository's elimination code): structural pivots and dense rows
of the Schur complement from the COMPILED REFERENCE (oracle/_ref: spasm_pivots_extract_structural, spasm_schur_dense),
then an exact modular elimination of those rows in numpy (float64 products of residues < 2^16 summed over < 2^20 terms
stay below 2^53).  rank = pivots + rank(S).  Needs p < 2^16.   python tools/cpu_rank_check.py mk13.b5"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import oracle as orc          # noqa: E402
import workloads                           # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mk13.b5"
p = 42013
n, m, ti, tj, tx = workloads._triplets_of(name)
if n < m:
    ti, tj, n, m = tj, ti, m, n
A = orc.compress(p, n, m, ti, tj, tx)
t0 = time.time()
npiv, perm, F = orc.ref_pivots_extract_structural(A, orc.empty_fact(A.n, A.m, p)) if hasattr(orc, "ref_pivots_extract_structural") \
    else orc.pivots_extract_structural(A, orc.empty_fact(A.n, A.m, p))
rows = perm[npiv:]
Sm = A.m - F.U.n
print("%s: %d x %d, %d structural pivots, %d rows left, %d non-pivotal columns (%.0f s)" % (name, A.n, A.m, npiv, len(rows), Sm, time.time() - t0), flush=True)

basis = np.zeros((0, Sm), np.float64)      # reduced row echelon rows found so far
pivcols = []


def reduce_block(Y):
    """Y (float64 residues) minus its components on the basis; returns the rows that are left non-zero"""
    global basis
    if len(pivcols):
        C = Y[:, pivcols]                               # coefficients on the basis rows
        for lo in range(0, Sm, 1024):                   # (tiles keep the temporaries small)
            Y[:, lo:lo + 1024] = np.mod(Y[:, lo:lo + 1024] - np.mod(C @ basis[:, lo:lo + 1024], p), p)
    return Y[np.any(Y != 0, axis=1)]


def absorb(Y):
    """Gauss-Jordan of the rows of Y (already reduced by the basis) into the basis"""
    global basis
    for r in range(Y.shape[0]):
        row = Y[r]
        nz = np.flatnonzero(row)
        if nz.size == 0:
            continue
        j = int(nz[0])
        inv = pow(int(row[j]), p - 2, p)
        row = np.mod(row * inv, p)
        # clear column j in the basis and in the rows still to come
        if basis.shape[0]:
            f = basis[:, j].copy()
            basis -= np.outer(f, row)
            np.mod(basis, p, out=basis)
        if r + 1 < Y.shape[0]:
            f = Y[r + 1:, j].copy()
            sel = f != 0
            if sel.any():
                Y[r + 1:][sel] = np.mod(Y[r + 1:][sel] - np.outer(f[sel], row), p)
        basis = np.vstack([basis, row[None, :]])
        pivcols.append(j)


BLOCK = 2048
t0 = time.time()
for lo in range(0, len(rows), BLOCK):
    sub = rows[lo:lo + BLOCK]
    S, q, p_out = orc.ref_schur_dense(A, sub, F)
    Y = np.mod(np.asarray(S, np.int64), p).astype(np.float64)
    left = reduce_block(Y)
    if left.shape[0]:
        absorb(left)
    if (lo // BLOCK) % 8 == 0:
        print("  rows %d / %d: rank(S) so far %d (%.0f s)" % (lo + len(sub), len(rows), len(pivcols), time.time() - t0), flush=True)
print("%s: rank = %d pivots + %d = %d (%.0f s)" % (name, npiv, len(pivcols), npiv + len(pivcols), time.time() - t0), flush=True)
