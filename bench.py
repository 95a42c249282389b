#!/usr/bin/env python3
"""bench.py -- rows eliminated per second on the sparse Schur-complement hot path.

Workload (BASELINE.json configs[1]): mk13.b5, the 135135 x 270270 boundary
matrix of the matching complex of K13 (hpac "Homology" collection), mod 42013,
processed the way tools/rank does: transposed to 270270 x 135135, structural
pivots found on the host, then ONE STEP = the Schur complement of every
non-pivotal row w.r.t. those pivots (spasm_schur, spasm_schur.c:64) -- here
spasm_hip_dschur with A, the factor and the row list already resident in HBM.
There is no network, so the matrix is regenerated from its definition (rows =
5-edge matchings of K13, columns = 4-edge matchings, entries +-1); `data` says
so.  Its dimensions and nnz (1,351,350) are those of the published file.

Prints ONE JSON line (rank 0).  N > 1 (launched by torch.distributed.run): the
row batch is sharded over the ranks, each rank reduces its slice, the slices
are reassembled on every rank with an all-gatherv over RCCL (strong scaling).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("SPASM_HIP_VERBOSE", "0")

PRIME = 42013


# --------------------------------------------------------------------------
# workload
# --------------------------------------------------------------------------
def build_workload(name):
    """(A, rows, F, source) -- tools/workloads.py: the matrix as tools/rank prepares it, its structural pivots
    (single-threaded search: the same pivots on every rank and in every run), the non-pivotal rows."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import workloads
    return workloads.round0(name, PRIME)


# --------------------------------------------------------------------------
# CPU baseline: the real reference (oracle/_ref, OpenMP) when it travelled
# with the repo, else the single-thread oracle port.  Rank 0, N = 1 only.
# --------------------------------------------------------------------------
def cpu_baseline(A, rows, F, budget_s=20.0):
    from oracle import oracle as orc
    Ao = orc.CSR(A.n, A.m, A.p, A.j, A.x, A.prime)
    Fo = orc.Fact(orc.CSR(F.U.n, F.U.m, F.U.p, F.U.j, F.U.x, A.prime), F.qinv)
    cores = os.cpu_count() or 1
    if orc.ref_available():
        kind, threads = "reference", cores
        run = lambda sub: orc.ref_schur(Ao, sub, Fo, threads=threads)
    else:
        kind, threads = "port", 1
        run = lambda sub: orc.schur(Ao, sub, Fo)
    # bounded sample: grow the slice until it costs a couple of seconds, then time one bigger slice
    count = min(len(rows), 2000)
    t = 0.0
    while True:
        t0 = time.perf_counter()
        run(rows[:count])
        t = time.perf_counter() - t0
        if t > 2.0 or count == len(rows):
            break
        count = min(len(rows), count * 4)
    final = min(len(rows), max(count, int(count * min(budget_s / max(t, 1e-3), 8.0))))
    if final > count:
        t0 = time.perf_counter()
        run(rows[:final])
        t = time.perf_counter() - t0
        count = final
    return {"value": count / t, "unit": "rows/s", "cores": threads, "kind": kind,
            "sample": "first %d of %d non-pivotal rows, spasm_schur, %.1f s" % (count, len(rows), t)}


# --------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="mk13.b5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import spasm_amd
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    use_dist = world > 1 or os.environ.get("SPASM_BENCH_FORCE_DIST") == "1"   # the flag lets a 1-GPU box run the RCCL path
    if use_dist:
        import torch.distributed as dist
        # RCCL prints a version banner on stdout when the communicator is created: stdout carries the one JSON line
        # only, so the banner goes to stderr (file descriptor level: it is printed by the C library)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    A, rows, F, source = build_workload(args.workload)
    from spasm_amd.dist import shard_rows, allgatherv_csr
    my_rows = shard_rows(rows, rank, world)
    dA = spasm_amd.DeviceCsr.from_host(A, dev)
    dF = spasm_amd.DeviceFact(F)
    drows = torch.from_numpy(np.ascontiguousarray(my_rows)).to(dev)
    stream = torch.cuda.Stream(device=dev)

    # size the pool with one probing run (not timed)
    pool = 4 * A.nnz + (1 << 24)
    while True:
        W = spasm_amd.SchurWorkspace(max(len(my_rows), 1), A.m, pool)
        with torch.cuda.stream(stream):
            S, st = spasm_amd.dschur(dA, drows, dF, W, stream=stream.cuda_stream, fetch=False)
        if st.status == 0:
            break
        W.close()
        pool *= 2

    def step():
        dF.forget()          # a step is the whole of spasm_schur: derived factor state (the back-substituted rows) is rebuilt
        with torch.cuda.stream(stream):
            S, st = spasm_amd.dschur(dA, drows, dF, W, stream=stream.cuda_stream, fetch=use_dist)
            if use_dist:
                full = allgatherv_csr(S, dist)
                assert full.n == len(rows)
        return st

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k_ms, k_elim, k_stream, k_in, k_out = 0.0, 0, 0, 0, 0
    t_tiers = [0.0, 0.0, 0.0, 0.0, 0.0]
    rows_by_tier = (0, 0, 0)
    for _ in range(args.steps):
        st = step()
        k_ms += st.ms_group if st.used_group_kernel else st.ms_tier2
        for q, v in enumerate((st.ms_tier0, st.ms_tier1, st.ms_tier2, st.ms_finalize, st.ms_group)):
            t_tiers[q] += v
        rows_by_tier = (st.rows_lds, st.rows_lds_big, st.rows_dense)
        group = bool(st.used_group_kernel)
        k_elim, k_stream, k_in, k_out = st.eliminations, st.entries_streamed, st.input_entries, st.nnz
        k_gp = st.group_pivots
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_rows = len(rows)
        ms_per_step = 1e3 * elapsed / args.steps
        # algorithmic bytes per launch (DESIGN.md "Byte accounting"), the traffic of the reference's own
        # algorithm on a dense x: per streamed entry of U' 8 B read + 8 B read-modify-write of x;
        # per elimination 16 B of row extent + 4 B coefficient; per input/output entry 8 B;
        # 20 B per reduced row (row id, two row pointers)
        algo = 16 * k_stream + 8 * (k_in + k_out) + 20 * k_elim + 20 * len(my_rows)
        kernel_ms = k_ms / args.steps
        achieved = algo / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic = None
        atomic_info = None
        tfile = os.path.join(ROOT, "profiles", "traffic_r01.json")
        if os.path.exists(tfile) and world == 1:          # (measured on the whole batch: says nothing about a rank's slice)
            tj = json.load(open(tfile))
            traffic = tj.get("schur_group_kernel_bytes_per_launch" if group else "schur_wave_dense_kernel_bytes_per_launch")
            if group and tj.get("schur_group_kernel_atomic_requests_per_launch"):
                # what actually bounds the kernel: 64-byte atomic requests to the memory side
                atomic_info = {"requests_per_launch": tj["schur_group_kernel_atomic_requests_per_launch"],
                               "achieved_per_s": tj["schur_group_kernel_atomic_requests_per_launch"] / (kernel_ms * 1e-3),
                               "measured_ceiling_per_s": tj.get("atomic_request_ceiling_per_s"),
                               "source": "rocprofv3 TCC_ATOMIC_sum (profiles/r01_g_atomic_counters.txt); ceiling: tools/microbench_atomics.hip"}
        # waves per row group: the library's own rule (schur_api.hip), for the kernel name rocprofv3 shows
        cus = torch.cuda.get_device_properties(0).multi_processor_count
        ngroups_rank = (len(my_rows) + 63) // 64
        group_waves = int(os.environ.get("SPASM_HIP_GROUP_WAVES", 4 if ngroups_rank <= 3 * cus else 2 if ngroups_rank <= 12 * cus else 1))
        out = {
            "metric": "rows eliminated/sec (sparse Schur complement, mod 42013)",
            "value": total_rows / (elapsed / args.steps),
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic (mk13.b5 regenerated from the definition of the matching complex of K13; no network)",
            "config": {"workload": "%s (%dx%d, %d nnz) mod %d, round-0 Schur complement of %d non-pivotal rows "
                                   "w.r.t. %d structural pivots" % (args.workload, A.n, A.m, A.nnz, PRIME,
                                                                   total_rows, F.U.n),
                       "rows_per_step": total_rows, "pivots": int(F.U.n), "levels": dF.levels,
                       "eliminations_per_step": int(k_elim), "schur_nnz": int(k_out), "group_pivots": int(k_gp),
                       "sharding": "rows over %d rank(s)%s" % (world, ", all-gatherv of S" if use_dist else "")},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "kernel": ("schur_group_kernel<false,true,%d>" % group_waves) if group else "schur_wave_dense_kernel<false>",
                         "kernel_ms": kernel_ms,
                         "algorithmic_bytes": int(algo),
                         "atomic_requests": atomic_info,
                         "ms_by_kernel": {"schur_lds_kernel<1024>": t_tiers[0] / args.steps,
                                          "schur_lds_kernel<8192>": t_tiers[1] / args.steps,
                                          "schur_wave_dense_kernel": t_tiers[2] / args.steps,
                                          "schur_group_kernel": t_tiers[4] / args.steps,
                                          "scan+gather_rows": t_tiers[3] / args.steps},
                         "rows_by_kernel": {"lds_small": rows_by_tier[0], "lds_big": rows_by_tier[1],
                                            ("row_group" if group else "wave_dense"): rows_by_tier[2]},
                         "lane_efficiency": (k_elim / (64.0 * k_gp)) if (group and k_gp) else None},
        }
        if world == 1:
            # the other half of the headline metric: wall-clock time of the whole rank computation
            # (host I/O excluded: the matrix is already in memory), default options of tools/rank
            os.environ.pop("SPASM_HIP_THREADS", None)
            t0 = time.perf_counter()
            fact = spasm_amd.echelonize(A)
            cold = time.perf_counter() - t0          # first call of the process: one-time allocations included
            t0 = time.perf_counter()
            fact = spasm_amd.echelonize(A)
            out["end_to_end"] = {"what": "spasm_hip_echelonize on the same matrix, default options", "rank": int(fact.U.n),
                                 "seconds": time.perf_counter() - t0, "seconds_first_call": cold}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(A, rows, F)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
