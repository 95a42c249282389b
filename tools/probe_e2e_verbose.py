#!/usr/bin/env python3
"""spasm_hip_echelonize on a stand-in with the library's own log (SPASM_HIP_VERBOSE, 3 by default: the laps of the factor plan
too); the second call is the one to read (the first loads code objects and fills the block cache).
python tools/probe_e2e_verbose.py [name] [verbosity]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "mk13.b5"
os.environ["SPASM_HIP_VERBOSE"] = sys.argv[2] if len(sys.argv) > 2 else "3"
import workloads, spasm_amd
A, _ = workloads.load_matrix(name)
spasm_amd.echelonize(A)
print("=========== second call", flush=True)
t = time.time(); F = spasm_amd.echelonize(A); print("rank", F.U.n, time.time() - t, spasm_amd.echelonize_profile(), spasm_amd.echelonize_counters())
