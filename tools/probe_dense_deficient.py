#!/usr/bin/env python3
"""GPU box: spasm_hip_drref on a random block whose independent columns are interleaved with dependent ones (one in three
independent: every panel of 64 columns holds ~21 pivots), generic rows: what the try that takes what its candidates give
(BlockGjArgs mode 2) is for.  SPASM_HIP_RREF_TIMING=1 prints how the panels went."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")
os.environ.setdefault("SPASM_HIP_VERBOSE", "0")
os.environ.setdefault("SPASM_HIP_RREF_TIMING", "1")
import torch
import spasm_amd
p = 42013
n, m, every = 4096, 4992, 3
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(7)
ngen = (m + every - 1) // every
k = min(n, ngen) - 80                                    # rank: a little below the number of independent columns
def mm(a, b):
    out = torch.zeros((a.shape[0], b.shape[1]), dtype=torch.int64, device=dev)
    af, bf = a.to(torch.float64), b.to(torch.float64)
    for c in range(0, a.shape[1], 64):
        out = (out + (af[:, c:c + 64] @ bf[c:c + 64]).to(torch.int64)) % p
    return out
L = torch.randint(0, p, (n, k), dtype=torch.int64, device=dev, generator=g)
Rg = torch.randint(0, p, (k, ngen), dtype=torch.int64, device=dev, generator=g)          # the generic columns
R = torch.zeros((k, m), dtype=torch.int64, device=dev)
R[:, ::every] = Rg
for off in range(1, every):
    cols = torch.arange(off, m, every, device=dev)
    # column j = a combination of the 8 generic columns before it
    for t in range(8):
        src = torch.clamp(cols // every - t, min=0)
        coef = torch.randint(0, p, (len(cols),), dtype=torch.int64, device=dev, generator=g)
        R[:, cols] = (R[:, cols] + Rg[:, src] * coef) % p
M = mm(L, R).to(torch.int32)
lib = spasm_amd.lib()
piv = torch.zeros(m, dtype=torch.int32, device=dev)
for _ in range(3):
    A = M.clone()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = lib.spasm_hip_drref(p, n, m, A.data_ptr(), m, piv.data_ptr(), 0)
    e1.record()
    torch.cuda.synchronize()
    print("rank %d (expected %d), %.2f ms" % (r, k, e0.elapsed_time(e1)), flush=True)
