#!/usr/bin/env python3
"""GPU box: the RREF of a dense block the flow really produces (the first 4,096 non-pivotal rows of a workload as dense Schur
rows) against a random block of the same shape, with the time split of the library (SPASM_HIP_RREF_TIMING=1)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")
os.environ.setdefault("SPASM_HIP_VERBOSE", "0")
os.environ.setdefault("SPASM_HIP_RREF_TIMING", "1")
import numpy as np
import torch
import spasm_amd
import workloads
import bench

name = sys.argv[1] if len(sys.argv) > 1 else "mk13.b5"
dev = torch.device("cuda:0")
A, rows, F, _ = workloads.round0(name, 42013)
dA = spasm_amd.DeviceCsr.from_host(A, dev)
drows = torch.from_numpy(np.ascontiguousarray(rows)).to(dev)
dF = spasm_amd.DeviceFact(F)
Sm = A.m - F.U.n
print("== real block", flush=True)
print(bench.dense_tail_real_probe(torch, spasm_amd, dev, dA, drows, dF, Sm), flush=True)
print("== random block of the same shape", flush=True)
print(bench.dense_tail_probe(torch, spasm_amd, dev, n=4096, m=Sm // 64 * 64), flush=True)
print("== random 4096 x 32768", flush=True)
print(bench.dense_tail_probe(torch, spasm_amd, dev), flush=True)
print("== the real block through the row panels (spasm_hip_dechelon_extend on an empty echelon form)", flush=True)
import ctypes as C
L = spasm_amd.lib()
n = min(4096, int(drows.numel()))
ld = (Sm + 63) // 64 * 64
S0 = torch.zeros((n, ld), dtype=torch.int32, device=dev)
W = spasm_amd.SchurWorkspace(n, dA.m, 1 << 20)
a = dA.cstruct(nnz=-1)
sub = drows[:n].contiguous()
rc = L.spasm_hip_dschur_dense(C.byref(a), sub.data_ptr(), n, dF._h, W._h, S0.data_ptr(), ld, 0)
torch.cuda.synchronize()
piv = torch.zeros(ld, dtype=torch.int32, device=dev)
for _ in range(4):
    M = S0.clone()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    r = L.spasm_hip_dechelon_extend(42013, Sm, M.data_ptr(), ld, 0, n, piv.data_ptr(), 0)
    ev1.record()
    torch.cuda.synchronize()
    print("  rank %d, %.2f ms" % (r, ev0.elapsed_time(ev1)), flush=True)
