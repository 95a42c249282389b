#!/usr/bin/env python3
"""GPU box: spasm_hip_echelonize with the greedy pivot search DISABLED (the flow of BASELINE configs[4], M0,6-D9:
tools/echelonize.c:36, spasm_pivots.c:315) on generated stand-ins.  python tools/probe_no_greedy.py name[:threshold] ..."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")
os.environ.setdefault("SPASM_HIP_VERBOSE", "1")
import spasm_amd          # noqa: E402
import workloads          # noqa: E402

for arg in sys.argv[1:] or ["mk13.b4:0.05", "ch7-8.b5:0.01", "mk13.b5:0.05", "mk14.b4:0.05"]:
    name, _, thr = arg.partition(":")
    A, _ = workloads.load_matrix(name)
    o = spasm_amd.default_opts()
    o.enable_greedy_pivot_search = 0
    if thr:
        o.sparsity_threshold = float(thr)
    for k in range(2):
        t = time.time()
        F = spasm_amd.echelonize(A, o)
        print("== %s no-greedy call %d: rank %d, %.2f s, %s" % (name, k, F.U.n, time.time() - t, spasm_amd.echelonize_profile()), flush=True)
