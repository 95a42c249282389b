#!/usr/bin/env python3
"""GPU box: the device pivot search with and without depth labels (pivots_device.hip, round 5) on generated stand-ins:
pivots found, time of spasm_hip_pivots_extract_structural, and the kernel's own statistics (SPASM_HIP_PIVOT_STATS).
python tools/probe_pivot_labels.py [name ...]  (environment: SPASM_HIP_PIVOT_GAP, SPASM_HIP_PIVOT_CASCADE)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")
os.environ.setdefault("SPASM_HIP_VERBOSE", "1")
os.environ["SPASM_HIP_PIVOT_STATS"] = "1"
os.environ["SPASM_HIP_PIVOT_SEARCH"] = "device"
import spasm_amd          # noqa: E402
import workloads          # noqa: E402

names = sys.argv[1:] or ["mk13.b5", "ch8-8.b5", "mk14.b4", "mk15.b4"]
for name in names:
    A, _ = workloads.load_matrix(name)
    print("== %s: %d x %d, %d nnz" % (name, A.n, A.m, A.nnz), flush=True)
    for labels in ("1", "0", "1"):
        os.environ["SPASM_HIP_PIVOT_LABELS"] = labels
        t = time.time()
        npiv, perm, F = spasm_amd.pivots_extract_structural(A, spasm_amd.empty_fact(A.m, 42013))
        print("== %s labels=%s: %d pivots, %.3f s" % (name, labels, npiv, time.time() - t), flush=True)
