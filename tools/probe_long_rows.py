#!/usr/bin/env python3
"""GPU box: the pivot search on a random matrix whose rows are too long for the 16-byte records of the device search
(GL7d19 has ~19 entries per row): device against host threads.  python tools/probe_long_rows.py [n m per_row]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")          # (this script names kernel variants and debugging aids)
os.environ.setdefault("SPASM_HIP_VERBOSE", "1")
os.environ.setdefault("SPASM_HIP_PIVOT_STATS", "1")
import spasm_amd
from spasm_amd.matrix import Csr

n, m, per_row = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (300000, 310000, 19)
rng = np.random.default_rng(5)
lens = np.full(n, per_row)
base = rng.integers(0, m, size=n)
stride = rng.integers(1, m // (per_row + 1), size=n)
p = np.zeros(n + 1, np.int64)
np.cumsum(lens, out=p[1:])
row = np.repeat(np.arange(n), lens)
k = np.arange(int(p[n])) - np.repeat(p[:-1], lens)
j = ((base[row] + k * stride[row]) % m).astype(np.int32)
x = rng.integers(1, 42013, size=int(p[n])).astype(np.int32)
A = Csr(n, m, p, j, x, 42013)
for where in (sys.argv[4].split(",") if len(sys.argv) > 4 else ("device", "host")):
    os.environ["SPASM_HIP_PIVOT_SEARCH"] = where
    t0 = time.time()
    npiv, perm, F = spasm_amd.pivots_extract_structural(A, spasm_amd.empty_fact(A.m, A.prime))
    print("== %s: %d pivots, %.2f s" % (where, npiv, time.time() - t0), flush=True)
