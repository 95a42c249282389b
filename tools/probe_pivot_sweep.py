#!/usr/bin/env python3
"""GPU box: parameter sweep of the labelled pivot search (waves per CU, cascade cap, label gap).
python tools/probe_pivot_sweep.py name [name ...]"""
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["SPASM_HIP_EXPERIMENT"] = "1"
os.environ["SPASM_HIP_VERBOSE"] = "0"
os.environ["SPASM_HIP_PIVOT_SEARCH"] = "device"
import spasm_amd          # noqa: E402
import workloads          # noqa: E402

names = sys.argv[1:] or ["mk15.b4"]
for name in names:
    A, _ = workloads.load_matrix(name)
    spasm_amd.pivots_extract_structural(A, spasm_amd.empty_fact(A.m, 42013))          # warm-up (allocations)
    for waves in (8, 12, 16):
        for cap in (1024, 2048, 8192):
            for gap in (16, 64):
                os.environ["SPASM_HIP_PIVOT_WAVES_PER_CU"] = str(waves)
                os.environ["SPASM_HIP_PIVOT_CASCADE"] = str(cap)
                os.environ["SPASM_HIP_PIVOT_GAP"] = str(gap)
                best = None
                for rep in range(2):
                    t = time.time()
                    npiv, perm, F = spasm_amd.pivots_extract_structural(A, spasm_amd.empty_fact(A.m, 42013))
                    dt = time.time() - t
                    best = dt if best is None else min(best, dt)
                print("%s waves %2d cap %5d gap %3d: %d pivots, %.3f s" % (name, waves, cap, gap, npiv, best), flush=True)
