#!/usr/bin/env python3
"""Writes a regenerated hpac matrix in SMS format: `gen_matrix.py mk13.b5 out.sms`.
mkN.bK = boundary map of the matching complex of K_N from K-edge to (K-1)-edge matchings."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import matching_complex_boundary   # noqa: E402

name, out = sys.argv[1], sys.argv[2]
nv = int(name.split(".")[0][2:])
k = int(name.split(".b")[1])
n, m, ti, tj, tx = matching_complex_boundary(nv, k)
with open(out, "w") as f:
    f.write("%d %d M\n" % (n, m))
    for a, b, c in zip(ti.tolist(), tj.tolist(), tx.tolist()):
        f.write("%d %d %d\n" % (a + 1, b + 1, c))
    f.write("0 0 0\n")
print("wrote %s: %d x %d, %d entries" % (out, n, m, len(ti)))
