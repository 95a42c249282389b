#!/usr/bin/env python3
"""GPU box: spasm_hip_drref of the first dense block of a workload's own finish, four calls (for a kernel trace)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")
os.environ.setdefault("SPASM_HIP_VERBOSE", "0")
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
print(bench.dense_tail_real_probe(torch, spasm_amd, dev, dA, drows, dF, A.m - F.U.n), flush=True)
