#!/usr/bin/env python3
"""bench.py -- rows eliminated per second on the sparse Schur-complement hot path.

Workload (BASELINE.json configs[1]): mk13.b5, the 135135 x 270270 boundary matrix of the matching complex
of K13 (hpac "Homology" collection; rows = 6-edge matchings, columns = 5-edge matchings, 810,810 entries
+-1), mod 42013, processed the way tools/rank does (tools/rank.c:76-104): transposed to 270270 x 135135,
structural pivots found on the host, then ONE STEP = the Schur complement of every non-pivotal row w.r.t.
those pivots (spasm_schur, spasm_schur.c:61) -- here spasm_hip_dschur with A, the factor image and the row
list already resident in HBM.  Derived factor state (the back-substituted rows R = U_pp^-1 U_pn) is
forgotten before every step, so a step pays for all of spasm_schur.  There is no network: the matrix is
regenerated from its definition (tools/workloads.py) unless $SPASM_DATA/mk13.b5.sms exists; `data` says which.
The other BASELINE configs are data files; `configs` in the output says which are present.

Prints ONE JSON line (rank 0).  N > 1 (launched by torch.distributed.run), strong scaling (the matrix is fixed), two
splits (--split): `columns` -- the default when the factor takes the back-substituted path, whose columns never meet --
gives every rank a slab of the non-pivotal columns of ALL rows (spasm_hip_column_slab: nothing replicated but the few
pivotal entries of U, no gather; the ranks exchange the row lengths); `rows` shards the row list and reassembles S on
every rank with an all-gatherv over RCCL (the row-by-row kernels, whose rows never meet).

Everything in the `roofline` object is measured in this run (HIP events of the library on the launch
stream, the kernels' own work counters) except `traffic`, which rocprofv3 has to collect in separate
passes: it is quoted from profiles/r06_traffic.json (written by tools/profile.sh; older rounds' files are looked at
next) only when that file was recorded for the same kernel on the same workload, with its path in
`traffic_source`; otherwise null.

Next to the step, on one GPU: the same batch through the row-by-row kernels, the dense tail, five end-to-end calls
(first call, median and minimum are all reported), `sparse_path` -- the flow of the GL7d19 class on the generated
stand-in mk14.b4: its Schur complement through the sparse image (the default there), the dense image and the
row-by-row kernels --, `at_scale` -- mk15.b4, 2,837,835 x 675,675 with 14.2 M entries, the size of GL7d19 -- and the
chessboard stand-ins.
"""
import argparse
import ctypes as C
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("SPASM_HIP_VERBOSE", "0")

PRIME = 42013
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_I8_PEAK_TOPS = 5000.0     # dense i8 (the guide's ~5 P op/s class; AMD's sparsity figures are not used)
# rocprofv3 PMC passes (tools/profile.sh): newest first; a file only counts for the workload, row count and kernel it names
TRAFFIC_FILES = ["profiles/r06_traffic.json", "profiles/r05_traffic.json", "profiles/r04_traffic.json", "profiles/r04_sparse_traffic.json", "profiles/r03_traffic.json", "profiles/r03_sparse_traffic.json",
                 "profiles/r02_traffic.json"]


# --------------------------------------------------------------------------
# CPU baseline: the real reference (oracle/_ref, OpenMP) when it travelled
# with the repo, else the single-thread oracle port.  Rank 0, N = 1 only.
# --------------------------------------------------------------------------
def cpu_baseline(A, rows, F, budget_s=20.0):
    from oracle import oracle as orc
    Ao = orc.CSR(A.n, A.m, A.p, A.j, A.x, A.prime)
    Fo = orc.Fact(orc.CSR(F.U.n, F.U.m, F.U.p, F.U.j, F.U.x, A.prime), F.qinv)
    # the CPUs this process may really use (a box of this pool reports 256 hardware threads and grants 16: cgroup cpu.max)
    import spasm_amd
    cores = spasm_amd.usable_cpus()
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
            "hardware_threads": os.cpu_count(),
            "sample": "first %d of %d non-pivotal rows, spasm_schur, %.1f s, %d OpenMP threads = the CPU quota of the box"
                      % (count, len(rows), t, threads)}


class _ModEchelon:
    """reduced row echelon basis mod p of the rows handed to add(), exact, in float64 (residues < 2^16: products summed over
    <= 2^20 terms at a time stay below 2^53) -- the stand-in for FFPACK in cpu_rank_time: panels of 64 rows are reduced row
    by row, everything else is matrix products (numpy's BLAS, the threads of the box)."""

    def __init__(self, m, p):
        self.m, self.p = m, float(p)
        self.ip = int(p)
        self.basis = np.zeros((0, m), np.float64)
        self.pivcols = []

    def _mod(self, Y):
        Y -= np.floor(Y / self.p) * self.p
        return Y

    def _minus_product(self, Y, Cf, B):
        """Y -= Cf @ B mod p, 2^20 columns of Cf at a time (sums of 2^20 products of residues < 2^16: < 2^52)"""
        for k0 in range(0, Cf.shape[1], 1 << 20):
            Y -= Cf[:, k0:k0 + (1 << 20)] @ B[k0:k0 + (1 << 20)]
            self._mod(Y)

    def add(self, S):
        p = self.ip
        Y = self._mod(S.astype(np.float64))
        if self.pivcols:
            self._minus_product(Y, Y[:, self.pivcols].copy(), self.basis)
        Y = Y[np.any(Y != 0, axis=1)]
        while Y.shape[0]:
            P = Y[:64].copy()
            rest = Y[64:]
            cols, keep = [], []
            for r in range(P.shape[0]):              # Gauss-Jordan inside the panel
                nz = np.flatnonzero(P[r])
                if nz.size == 0:
                    continue
                j = int(nz[0])
                P[r] = self._mod(P[r] * float(pow(int(P[r, j]), p - 2, p)))
                f = P[:, j].copy()
                f[r] = 0.0
                sel = f != 0
                if sel.any():
                    P[sel] = self._mod(P[sel] - np.outer(f[sel], P[r]))
                cols.append(j)
                keep.append(r)
            if cols:
                P = P[keep]
                if rest.shape[0]:
                    self._minus_product(rest, rest[:, cols].copy(), P)
                if self.basis.shape[0]:
                    self._minus_product(self.basis, self.basis[:, cols].copy(), P)
                self.basis = np.vstack([self.basis, P])
                self.pivcols.extend(cols)
            Y = rest[np.any(rest != 0, axis=1)] if rest.shape[0] else rest


def cpu_rank_time(A, budget_s=20.0, block=2048):
    """The second half of the metric on the host cores: wall-clock time of a rank computation the way the reference does it
    (tools/rank.c:76-104 -> spasm_echelonize, spasm_echelonize.c:473).  spasm_echelonize.c itself cannot be compiled here (FFPACK),
    so the time is composed from the compiled reference's own pieces on the CPUs the box grants: spasm_pivots_extract_structural
    (in full) + the dense finish the reference takes on a matrix whose Schur complement is dense (echelonize_dense,
    spasm_echelonize.c:315-393: spasm_schur_dense on blocks of rows, each block reduced to echelon rows -- an exact numpy
    elimination stands in for FFPACK).  Blocks run until the budget is spent; the rest is PROJECTED from the blocks done after
    the rank of S stopped growing (their cost no longer changes) and reported as such."""
    from oracle import oracle as orc
    import spasm_amd
    if not orc.ref_available():
        return None
    cores = spasm_amd.usable_cpus()
    orc.ref_set_threads(cores)
    p = A.prime
    Ao = orc.CSR(A.n, A.m, A.p, A.j, A.x, p)
    t_begin = time.perf_counter()
    npiv, perm, F = orc.ref_pivots_extract_structural(Ao, orc.empty_fact(A.n, A.m, p))
    t_piv = time.perf_counter() - t_begin
    rows = perm[npiv:]
    Sm = A.m - F.U.n
    ech = _ModEchelon(Sm, p)
    per_block, ranks = [], []
    for lo in range(0, len(rows), block):
        t0 = time.perf_counter()
        S, _, _ = orc.ref_schur_dense(Ao, rows[lo:lo + block], F)
        ech.add(np.asarray(S, np.int64))
        per_block.append(time.perf_counter() - t0)
        ranks.append(len(ech.pivcols))
        if time.perf_counter() - t_begin > budget_s:
            break
    pivcols = ech.pivcols
    done = len(per_block)
    total = (len(rows) + block - 1) // block
    measured = time.perf_counter() - t_begin
    projected = done < total
    if projected:
        # blocks after the last growth of the rank all cost the same (schur_dense + one product with the basis)
        steady = [t for t, r in zip(per_block, ranks) if r == ranks[-1]][1:]
        settled = len(steady) >= 2
        seconds = measured + (total - done) * statistics.median(steady or per_block[-1:])
    else:
        seconds, settled = measured, True
    return {"seconds": seconds, "projected": projected, "upper_bound": not settled, "measured_s": measured, "pivots_s": t_piv, "pivots": int(npiv),
            "blocks_done": done, "blocks_total": total, "rank_of_S_so_far": len(pivcols), "rank": (int(npiv) + len(pivcols)) if not projected else None,
            "cores": cores, "kind": "reference + numpy",
            "what": "spasm_pivots_extract_structural + spasm_schur_dense per %d rows (compiled reference, %d threads) + exact numpy echelon form" % (block, cores)}


def cpu_baseline_sampled(A, rows, F, count=2000):
    """the reference's spasm_schur on `count` rows spread evenly over the batch (the rows of a stand-in's Schur complement differ
    by orders of magnitude from one end of the batch to the other: the first rows alone would not say much)"""
    from oracle import oracle as orc
    import spasm_amd
    Ao = orc.CSR(A.n, A.m, A.p, A.j, A.x, A.prime)
    Fo = orc.Fact(orc.CSR(F.U.n, F.U.m, F.U.p, F.U.j, F.U.x, A.prime), F.qinv)
    cores = spasm_amd.usable_cpus()
    ks = np.unique(np.linspace(0, len(rows) - 1, min(count, len(rows))).astype(np.int64))
    sub = np.ascontiguousarray(rows[ks])
    if orc.ref_available():
        kind, threads = "reference", cores
        run = lambda: orc.ref_schur(Ao, sub, Fo, threads=threads)
    else:
        kind, threads = "port", 1
        run = lambda: orc.schur(Ao, sub, Fo)
    t0 = time.perf_counter()
    run()
    t = time.perf_counter() - t0
    return {"value": len(sub) / t, "unit": "rows/s", "cores": threads, "kind": kind, "hardware_threads": os.cpu_count(),
            "sample": "%d of %d non-pivotal rows spread evenly over the batch, spasm_schur, %.1f s, %d OpenMP threads = the CPU quota of the box"
                      % (len(sub), len(rows), t, threads)}


def quoted_traffic(kernel, workload, rows):
    """HBM bytes per launch of `kernel` from the rocprofv3 PMC passes of this round, if they were recorded on
    this workload.  (value, fetch_doubled_upper_bound, source) or (None, None, None)."""
    for rel in TRAFFIC_FILES:
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        try:
            t = json.load(open(path))
        except ValueError:
            continue
        if t.get("workload") != workload or not t.get("rows"):
            continue
        # (the stand-in's pivots come from the threaded search: the batch differs by a few rows in 670,000 from run to run)
        if t.get("rows") != rows and not (workload.startswith("mk14") and abs(t["rows"] - rows) <= 0.002 * rows):
            continue
        k = t.get("kernels", {}).get(kernel)
        if not k:
            continue
        return k.get("bytes_per_launch"), k.get("bytes_per_launch_fetch_doubled"), rel
    return None, None, None


def dense_tail_probe(torch, spasm_amd, dev, n=4096, m=32768):
    """the dense tail on its own: RREF mod 42013 of a random full-rank n x m block resident in HBM
    (spasm_hip_drref); useful multiply-adds of the trailing updates per second against the i8 MFMA peak."""
    L = spasm_amd.lib()
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    A0 = torch.randint(0, PRIME, (n, m), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    piv = torch.zeros(m, dtype=torch.int32, device=dev)
    best = None
    r = 0
    for _ in range(3):
        A = A0.clone()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        r = L.spasm_hip_drref(PRIME, n, m, A.data_ptr(), m, piv.data_ptr(), 0)
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1)
        best = ms if best is None else min(best, ms)
    A = A0.clone()
    ms_upd = C.c_float(0)
    r = L.spasm_hip_drref_timed(PRIME, n, m, A.data_ptr(), m, piv.data_ptr(), 0, 1, C.byref(ms_upd))
    macs, c, left = 0, 0, r
    while left > 0 and c < m:
        k = min(64, left)
        macs += n * k * max(m - c - 64, 0)
        c += 64
        left -= k
    tmacs_total = macs / (best * 1e-3) / 1e12
    tmacs_upd = macs / (ms_upd.value * 1e-3) / 1e12 if ms_upd.value > 0 else None
    out = {"what": "spasm_hip_drref, random full-rank block mod %d" % PRIME, "shape": [n, m], "rank": int(r),
           "ms": best, "useful_Tmacs_per_s": tmacs_total,
           "update_kernels_ms_serialised": ms_upd.value, "update_kernels_Tmacs_per_s": tmacs_upd,
           # 4 int8 digit products per useful multiply-add (two base-256 digits each side), 2 ops per product
           "mfma_i8_frac_of_peak": (8 * tmacs_upd / MFMA_I8_PEAK_TOPS) if tmacs_upd else None}
    for rel in ("profiles/r06_dense_tail.json", "profiles/r05_dense_tail.json", "profiles/r04_dense_tail.json", "profiles/r03_dense_tail.json", "profiles/r02_dense_tail.json"):
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        try:
            q = json.load(open(path))
            if q.get("shape") == [n, m]:
                out["mfma_busy_pct"] = q.get("mfma_busy_pct")
                out["mfma_busy_source"] = rel
                break
        except ValueError:
            pass
    return out


def dense_tail_real_probe(torch, spasm_amd, dev, dA, drows, dF, Sm, nrows=4096):
    """the dense tail on a block the flow really produces: the first `nrows` non-pivotal rows of the workload reduced to dense
    rows by the factor (spasm_schur_dense, spasm_schur.c:257-343) -- for mk13.b5 a 4,096 x 4,952 block of rank ~1,600 (the
    pivots of the run decide): its first 25 panels of 64 columns are full, then a few hundred live rows face panels with no
    pivot at all, 64 consecutive rows have rank ~40 on a panel, and rows that have become zero stay among the free ones --
    then its RREF (spasm_hip_drref), timed like the random block."""
    L = spasm_amd.lib()
    n = min(nrows, int(drows.numel()))
    ld = (Sm + 63) // 64 * 64
    S0 = torch.zeros((n, ld), dtype=torch.int32, device=dev)
    W = spasm_amd.SchurWorkspace(n, dA.m, 1 << 20)
    a = dA.cstruct(nnz=-1)
    sub = drows[:n].contiguous()
    rc = L.spasm_hip_dschur_dense(C.byref(a), sub.data_ptr(), n, dF._h, W._h, S0.data_ptr(), ld, 0)
    torch.cuda.synchronize()
    W.close()
    if rc != 0:
        return {"status": "spasm_hip_dschur_dense returned %d" % rc}
    piv = torch.zeros(ld, dtype=torch.int32, device=dev)
    times, r = [], 0
    for _ in range(4):
        A = S0[:, :Sm].contiguous()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        r = L.spasm_hip_drref(PRIME, n, Sm, A.data_ptr(), Sm, piv.data_ptr(), 0)
        ev1.record()
        torch.cuda.synchronize()
        times.append(ev0.elapsed_time(ev1))
    return {"what": "spasm_hip_drref of the dense rows of the first %d non-pivotal rows of the workload (spasm_hip_dschur_dense), mod %d" % (n, PRIME),
            "shape": [n, Sm], "rank": int(r), "ms_first": times[0], "ms_median_of_the_rest": statistics.median(times[1:]), "ms_all": times}


def _calls(fn, count):
    """`count` timed calls of fn() -> (profile, rank): first call, median, minimum, all of them"""
    runs = []
    import spasm_amd
    for _ in range(count):
        t0 = time.perf_counter()
        prof, rank = fn()
        prof = dict(prof, events=spasm_amd.echelonize_counters())
        runs.append((time.perf_counter() - t0, prof, rank))
    secs = [r[0] for r in runs]
    med = sorted(runs, key=lambda r: r[0])[len(runs) // 2]
    stages = ("pivot_search", "density_estimates", "sparse_schur", "dense_finish")
    return {"rank": med[2], "ranks_agree": len({r[2] for r in runs}) == 1, "seconds_first_call": secs[0],
            "seconds_median": statistics.median(secs), "seconds_min": min(secs), "seconds_all": secs, "split_of_median_call": med[1],
            "stages_of_every_call": [dict({k: round(r[1].get(k, 0.0), 3) for k in stages}, events=r[1]["events"]) for r in runs]}


def sparse_path_probe(torch, spasm_amd, workloads, dev, name="mk14.b4", steps=3, paths=("default", "dense image", "row by row"), calls=5, cpu_rows=0):
    """The flow GL7d19 takes (a sparse round on a wide Schur complement, then the dense tail), on a matrix of the same
    collection that can be generated offline -- a STAND-IN, not a BASELINE config: mk14.b4 (945,945 x 315,315; its first
    Schur complement is 673,000 x 42,000, ~2 % dense, 5-7e8 entries depending on the pivots of the run) or, at the size of
    GL7d19, mk15.b4 (2,837,835 x 675,675; 2.2 M x 71,000, 1.4-1.9e9 entries).  Reports that Schur complement on the
    device-level API (the way the headline step is measured) through the path the library takes by itself -- since round 4
    the SPARSE image: R = U_pp^-1 U_pn as sparse fragments, sparse_image.hip -- and, where they exist for the factor, through
    the dense image and the row-by-row kernels; then whole spasm_hip_echelonize calls with their time split."""
    t0 = time.perf_counter()
    A, rows, F, source = workloads.round0(name, PRIME, threads=0)
    t_prep = time.perf_counter() - t0
    dA = spasm_amd.DeviceCsr.from_host(A, dev)
    drows = torch.from_numpy(np.ascontiguousarray(rows)).to(dev)
    ENV = {"default": {}, "dense image": {"SPASM_HIP_SPARSE_IMAGE": "0", "SPASM_HIP_BACKSOLVE": "1"},
           "row by row": {"SPASM_HIP_SPARSE_IMAGE": "0", "SPASM_HIP_BACKSOLVE": "0", "SPASM_HIP_SCRATCH_GB": "24"}}
    pool_hint = [1 << 30]

    def measure(env):
        os.environ.update(env)
        try:
            t0 = time.perf_counter()
            dF = spasm_amd.DeviceFact(F)
            torch.cuda.synchronize()
            image_ms = 1e3 * (time.perf_counter() - t0)
            while True:
                W = spasm_amd.SchurWorkspace(len(rows), A.m, pool_hint[0])
                _, st = spasm_amd.dschur(dA, drows, dF, W, fetch=False)
                if st.status == 0:
                    break
                W.close()
                pool_hint[0] *= 2
            all_ms = []
            best = None
            for _ in range(steps):
                dF.forget()                      # (a step builds what it needs: the image R is part of it)
                _, st = spasm_amd.dschur(dA, drows, dF, W, fetch=False)
                all_ms.append(st.ms_total)
                if best is None or st.ms_total < best.ms_total:
                    best = type(st).from_buffer_copy(st)
            census = dF.sparse_image_census() if best.used_sparse_image else None
            levels = dF.levels
            W.close()
            dF.close()
            return best, all_ms, image_ms, levels, census
        finally:
            for k in env:
                os.environ.pop(k, None)

    r, Sm = int(F.U.n), int(A.m - F.U.n)
    out = {"what": "%s (%dx%d, %d nnz), STAND-IN for the GL7d19 class: round-0 Schur complement of %d rows w.r.t. %d pivots, "
                   "%d non-pivotal columns, device-level API" % (name, A.n, A.m, A.nnz, len(rows), r, Sm),
           "rows": len(rows), "prepare_s": t_prep, "paths": {}}

    def kernel_entry(ms, by):
        gbs = by / (ms * 1e-3) / 1e9 if ms and ms > 0 else None
        return {"ms": ms if ms and ms > 0 else None, "algorithmic_bytes": int(by), "GB_per_s": gbs, "frac": gbs / HBM_PEAK_GBS if gbs else None}

    for path in paths:
        st, all_ms, image_ms, levels, census = measure(ENV[path])
        which = "sparse image" if st.used_sparse_image else "dense image" if st.used_backsolve else "row by row"
        if path == "dense image" and not st.used_backsolve:
            out["paths"][path] = {"status": "the factor is not eligible for the dense image (%d x %d entries)" % (r, Sm)}
            continue
        e = {"took": which, "ms_per_step": st.ms_total, "ms_all": all_ms, "rows_per_s": len(rows) / (st.ms_total * 1e-3), "schur_nnz": int(st.nnz),
             "schur_density": st.nnz / (len(rows) * float(Sm)), "factor_image_ms": image_ms, "levels": levels}
        if st.used_sparse_image:
            e["kernels"] = {"sp_build_kernel": kernel_entry(st.ms_sparse_build, st.bytes_sparse_build),
                            "sp_apply_kernel": kernel_entry(st.ms_sparse_apply, st.bytes_sparse_apply),
                            "scan + sp_gather_kernel": kernel_entry(st.ms_sparse_gather, st.bytes_sparse_gather)}
            e["multiply_adds"] = {"build": int(st.sparse_image_ops_build), "rows_of_S": int(st.sparse_image_ops_apply)}
            e["build_us_per_level"] = 1e3 * st.ms_sparse_build / max(1, st.sparse_image_levels)
            if census:
                tiles = r * ((Sm + 63) // 64)
                e["fill_of_R"] = {"entries": census["entries"], "fraction": census["entries"] / (float(r) * Sm),
                                  "entries_per_row": census["entries"] / float(r), "occupied_64_column_tiles": census["tiles64"],
                                  "tile_fraction": census["tiles64"] / float(tiles),
                                  "non_empty_fragments": census["fragments"], "fragment_fraction": census["fragments"] / float(max(1, census["pairs"]))}
        elif st.used_backsolve:
            e["kernels"] = {st.kernel.decode(): kernel_entry(st.ms_backsolve, st.bytes_backsolve),
                            st.kernel_other.decode(): kernel_entry(st.ms_apply, st.bytes_apply)}
            if st.ms_expand > 0:
                e["kernels"][st.kernel_expand.decode()] = kernel_entry(st.ms_expand, st.bytes_expand)
        else:
            kernel = st.kernel.decode()
            k_ms = st.ms_group if st.used_group_kernel and not st.group_aborted else st.ms_tier2
            algo = 16 * st.entries_streamed + 8 * (st.input_entries + st.nnz) + 20 * st.eliminations + 20 * st.rows
            e.update({"kernel": kernel, "kernel_ms": k_ms, "gather_ms": st.ms_finalize, "gave_up": bool(st.group_aborted),
                      "eliminations": int(st.eliminations), "entries_streamed": int(st.entries_streamed), "group_pivots": int(st.group_pivots),
                      "lane_efficiency": (st.eliminations / (64.0 * st.group_pivots)) if st.group_pivots else None,
                      "slices_in_flight": st.group_slots, "slices_wanted": st.group_slots_wanted, "waves_per_group": st.group_waves,
                      "slice_bytes": int(st.group_slot_bytes), "effective_bytes": int(algo),
                      "effective_frac": algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms > 0 else None,
                      "effective_note": "bytes the reference's algorithm moves on its dense x (DESIGN.md section 4); the row-group kernel "
                                        "shares a 256-B line among the 64 rows of a group, so this is not HBM traffic"})
            tr = quoted_traffic(kernel, name, len(rows))
            if tr[0] and k_ms > 0:
                e.update({"traffic": tr[0], "hbm_frac": tr[0] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic_source": tr[2]})
        out["paths"][path] = e
    nnzs = {e["schur_nnz"] for e in out["paths"].values() if "schur_nnz" in e}
    out["same_nnz_on_every_path"] = len(nnzs) == 1
    dflt = out["paths"].get("default", {})
    out.update({k: dflt.get(k) for k in ("ms_per_step", "rows_per_s", "schur_nnz", "schur_density", "took")})
    if cpu_rows > 0:
        out["cpu_baseline"] = cpu_baseline_sampled(A, rows, F, cpu_rows)
    # the same step on a pivot set that does not depend on timing (the sequential search with depth labels, one thread): the
    # device search above finds a different set in every run, and with it a Schur complement of another size
    t0 = time.perf_counter()
    A1, rows1, F1, _ = workloads.round0(name, PRIME, threads=1, labelled=True)
    t_fixed = time.perf_counter() - t0
    rows_keep, F_keep = rows, F
    try:
        rows, F = rows1, F1
        drows_keep = drows
        drows = torch.from_numpy(np.ascontiguousarray(rows)).to(dev)
        st1, all_ms1, image_ms1, levels1, _ = measure(ENV["default"])
        out["fixed_pivot_set"] = {"what": "the default path on the pivots of the sequential labelled search (SPASM_HIP_THREADS=1): the same workload in every run",
                                  "pivots": int(F1.U.n), "rows": len(rows1), "non_pivotal_columns": int(A.m - F1.U.n), "prepare_s": t_fixed,
                                  "took": "sparse image" if st1.used_sparse_image else "dense image" if st1.used_backsolve else "row by row",
                                  "ms_per_step": st1.ms_total, "ms_all": all_ms1, "rows_per_s": len(rows1) / (st1.ms_total * 1e-3), "schur_nnz": int(st1.nnz),
                                  "levels": levels1, "factor_image_ms": image_ms1}
        if st1.used_sparse_image:
            out["fixed_pivot_set"]["kernels_ms"] = {"sp_build_kernel": st1.ms_sparse_build, "sp_apply_kernel": st1.ms_sparse_apply, "scan + sp_gather_kernel": st1.ms_sparse_gather}
            out["fixed_pivot_set"]["kernels_frac"] = {k: kernel_entry(m_, b_)["frac"] for k, m_, b_ in (
                ("sp_build_kernel", st1.ms_sparse_build, st1.bytes_sparse_build), ("sp_apply_kernel", st1.ms_sparse_apply, st1.bytes_sparse_apply),
                ("scan + sp_gather_kernel", st1.ms_sparse_gather, st1.bytes_sparse_gather))}
            out["fixed_pivot_set"]["step_frac"] = (st1.bytes_sparse_build + st1.bytes_sparse_apply + st1.bytes_sparse_gather) / (st1.ms_total * 1e-3) / 1e9 / HBM_PEAK_GBS
            out["fixed_pivot_set"]["build_us_per_level"] = 1e3 * st1.ms_sparse_build / max(1, st1.sparse_image_levels)
    finally:
        rows, F, drows = rows_keep, F_keep, drows_keep
        del A1
    del dA, drows
    torch.cuda.empty_cache()
    os.environ.pop("SPASM_HIP_THREADS", None)

    def call():
        fact = spasm_amd.echelonize(A)
        return spasm_amd.echelonize_profile(), int(fact.U.n)
    out["end_to_end"] = dict(_calls(call, calls), what="spasm_hip_echelonize, default options, %d calls" % calls)
    return out


def stand_in_runs(spasm_amd, workloads):
    """the chessboard complexes with the options of the GL7d19 config (--dense-threshold 0.01): their first Schur complement
    is 18 % dense, so a call is pivot search + the device-resident low-rank finish on 49,000 / 104,000 columns."""
    out = []
    for name, info in workloads.STAND_INS.items():
        A, _ = workloads.load_matrix(name, PRIME)
        opts = spasm_amd.default_opts()
        for t, a in enumerate(info["rank_args"]):
            if a == "--dense-threshold":
                opts.sparsity_threshold = float(info["rank_args"][t + 1])
        def call():
            fact = spasm_amd.echelonize(A, opts)
            return spasm_amd.echelonize_profile(), int(fact.U.n)
        out.append(dict(_calls(call, 3), name=name, stand_in_for=info["for"], shape=[A.n, A.m], nnz=int(A.nnz), options=" ".join(info["rank_args"])))
    # the flow of BASELINE configs[4] (M0,6-D9: --no-greedy-pivot-search, tools/echelonize.c:36, spasm_pivots.c:315) on the generated
    # matrices: Faugere-Lachartre pivots only, so the first Schur complement is larger and the call runs two or three sparse
    # rounds, each with a factor image of its own, before the dense finish
    for name, thr in (("mk13.b4", 0.05), ("mk13.b5", 0.05), ("mk14.b4", 0.05)):
        A, _ = workloads.load_matrix(name, PRIME)
        opts = spasm_amd.default_opts()
        opts.enable_greedy_pivot_search = 0
        opts.sparsity_threshold = thr
        def call():
            fact = spasm_amd.echelonize(A, opts)
            return spasm_amd.echelonize_profile(), int(fact.U.n)
        runs = _calls(call, 3)
        out.append(dict(runs, name=name, stand_in_for="M0,6-D9 (the no-greedy flow)", shape=[A.n, A.m], nnz=int(A.nnz), options="--no-greedy-pivot-search",
                        sparse_rounds=runs["split_of_median_call"].get("sparse_rounds")))
    return out


# --------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None,
                    help="BASELINE config to run (default: GL7d19 -- the matrix BASELINE.json quotes its metric on -- when "
                         "$SPASM_DATA/GL7d19.sms[.gz] exists, else mk13.b5, the config that can be regenerated offline)")
    ap.add_argument("--split", choices=["auto", "rows", "columns"], default="auto",
                    help="how N > 1 ranks share a step: `rows` = contiguous slices of the row list, S reassembled on every rank with "
                         "an all-gatherv over RCCL; `columns` = every rank owns a slab of the non-pivotal columns of ALL rows "
                         "(spasm_hip_column_slab: no replicated factor image, only the row lengths are exchanged); auto = columns "
                         "(the columns of R and S never meet on either image path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--beyond-scale", action="store_true",
                    help="also run spasm_hip_echelonize twice on mk15.b5 (4,729,725 x 2,837,835, 28.4 M entries: beyond the size of GL7d19; "
                         "about a minute more, 40 GB of device memory) and report it as `beyond_scale`")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the row-by-row comparison, the dense-tail probe and the end-to-end runs")
    args = ap.parse_args()

    import torch
    import spasm_amd
    import workloads
    why = "asked for"
    if args.workload is None:
        if workloads.find_data(workloads.config("GL7d19")["file"]) is not None:
            args.workload, why = "GL7d19", "BASELINE.json quotes its metric on GL7d19 and its data file is present"
        else:
            args.workload, why = "mk13.b5", ("GL7d19.sms (the matrix BASELINE.json quotes its metric on) is absent under %s: "
                                            "configs[1], the config that can be regenerated offline" % workloads.data_dir())
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE line, the last thing this process writes: whatever else lands on descriptor 1 while the bench runs
    # (RCCL's banner, libdrm's "amdgpu.ids" warning on some boxes, a library's printf) goes to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(full):
        """bench_full.json + stderr get everything; stdout gets the compact line (tools/bench_format.py), last"""
        import bench_format
        full["full_path"] = "bench_full.json"
        text = json.dumps(full)
        for where in (os.path.join(ROOT, "bench_full.json"), os.path.join(ROOT, "gpurun_out", "bench_full.json")):
            try:
                if os.path.isdir(os.path.dirname(where)):
                    with open(where, "w") as f:
                        f.write(text + "\n")
            except OSError as e:
                sys.stderr.write("bench.py: cannot write %s: %s\n" % (where, e))
        sys.stderr.write(text + "\n")
        sys.stderr.flush()
        sys.stdout.flush()
        C.CDLL(None).fflush(None)
        os.write(real_stdout, (bench_format.line(full) + "\n").encode())

    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    use_dist = world > 1 or os.environ.get("SPASM_BENCH_FORCE_DIST") == "1"   # the flag lets a 1-GPU box run the RCCL path
    comm = None
    if use_dist:
        import torch.distributed as dist
        from spasm_amd.dist import Comm
        dist.init_process_group("nccl", device_id=dev)
        warm = torch.zeros(1, device=dev)
        dist.all_reduce(warm)
        torch.cuda.synchronize()

        # the library's own RCCL communicator (C ABI, section (M)); its id travels through torch.distributed
        def exchange(raw):
            box = [raw]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        comm = Comm(rank, world, exchange)

    try:
        A, rows, F, source = workloads.round0(args.workload, PRIME)
    except FileNotFoundError as e:
        if rank == 0:
            emit({"metric": "rows eliminated/sec (sparse Schur complement, mod 42013)", "value": None,
                  "unit": "rows/s", "n_gpus": world, "data": "absent", "config": {"workload": args.workload},
                  "error": str(e)})
        return
    from spasm_amd.dist import shard_rows, column_slab
    split = args.split
    if split == "auto":
        # both image paths -- the dense one and, since round 4, the sparse one (no limit on the width) -- compute the columns of
        # S independently of each other: a slab of the non-pivotal columns per rank replicates nothing.  One rank: nothing to split
        split = "columns" if (world > 1 or os.environ.get("SPASM_BENCH_FORCE_DIST") == "1") else "rows"
    if world == 1 and os.environ.get("SPASM_BENCH_FORCE_DIST") != "1":
        split = "rows"
    A_full, F_full = A, F
    if split == "columns":
        # this rank's problem: (A, U) without the non-pivotal columns of the other ranks -- all rows, a slab of the columns
        A, F, slab_cols = column_slab(A_full, F_full, rank, world)
        my_rows = rows
    else:
        my_rows = shard_rows(rows, rank, world)
    dA = spasm_amd.DeviceCsr.from_host(A, dev)
    # the factor image (host: level schedule, pass tables, chunk plans; then the upload) is built once per factor, outside
    # the timed steps: its cost is measured here and reported next to the step (`factor_image_ms`, `rows_per_s_cold`)
    image_ms = []
    for k in range(4):                           # (the first build also loads the code object and fills the buffer cache: not counted)
        t0 = time.perf_counter()
        dF = spasm_amd.DeviceFact(F)
        torch.cuda.synchronize()
        if k > 0:
            image_ms.append(1e3 * (time.perf_counter() - t0))
        if k < 3:
            dF.close()
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

    if split == "columns":
        my_Sp = torch.zeros(len(my_rows) + 1, dtype=torch.int64, device=dev)
        my_len = torch.zeros(len(my_rows), dtype=torch.int64, device=dev)
        all_len = torch.zeros(world * len(my_rows), dtype=torch.int64, device=dev)

    def step():
        dF.forget()          # a step is the whole of spasm_schur: derived factor state (the back-substituted rows) is rebuilt
        with torch.cuda.stream(stream):
            S, st = spasm_amd.dschur(dA, drows, dF, W, stream=stream.cuda_stream, fetch=False)
            if use_dist and split == "columns":
                # the slabs stay where they are; what every rank needs to address whole rows later is their lengths
                spasm_amd.lib().spasm_hip_dschur_row_pointers(W._h, my_Sp.data_ptr(), stream.cuda_stream)
                torch.sub(my_Sp[1:], my_Sp[:-1], out=my_len)
                dist.all_gather_into_tensor(all_len, my_len)
            elif use_dist:
                full = comm.allgatherv(W, A.m, PRIME, stream=stream.cuda_stream)      # all-gatherv of S on the devices
                assert full.n == len(rows)
        return st

    # Several ranks: the step the line reports is the PRODUCT's -- spasm_hip_schur with the communicator installed, as
    # spasm_hip_echelonize calls it between two rounds: this rank's slab (or slice), the all-gatherv of all of S, the stitching into
    # whole rows on every device.  The device-API step above (slabs left where they are, lengths exchanged) stays as a second object.
    product = None
    if use_dist:
        if args.split != "auto" or world == 1:
            os.environ["SPASM_HIP_EXPERIMENT"] = "1"
            if args.split != "auto":
                os.environ["SPASM_HIP_SHARD"] = args.split
            if world == 1:
                os.environ["SPASM_HIP_SHARD_FORCE"] = "1"          # (SPASM_BENCH_FORCE_DIST=1: the same code in a world of one)
        comm.install()
        product = spasm_amd.ResidentSchur(A_full, rows, F_full)
        product_nnz = [0]

    # (the driver hands spasm_hip_schur its density estimate -- spasm_schur_estimate_density, 100 rows --, from which the row pool of S is
    #  sized; a first call without one finds the size the hard way and gives the estimate of the timed calls)
    product_density = [-1.0]

    def product_step():
        product_nnz[0] = product(forget=True, est_density=product_density[0])
        product_density[0] = 1.02 * product_nnz[0] / (max(len(rows), 1) * float(max(A_full.m - F_full.U.n, 1)))

    def run_steps(count):
        """count timed steps: (seconds, per-kernel device ms summed over the steps, their algorithmic bytes, last stats)"""
        ms = {}
        by = {}
        st = None
        t0 = time.perf_counter()
        for _ in range(count):
            st = step()
            if st.used_sparse_image:
                parts = (("sp_build_kernel", st.ms_sparse_build, st.bytes_sparse_build), ("sp_apply_kernel", st.ms_sparse_apply, st.bytes_sparse_apply),
                         ("sp_gather_kernel", st.ms_sparse_gather, st.bytes_sparse_gather))
            elif st.used_backsolve:
                names = (st.kernel.decode(), st.kernel_other.decode())
                build_name = [x for x in names if x.startswith("backsolve")][0]
                apply_name = [x for x in names if x.startswith("bs_apply")][0]
                parts = ((build_name, st.ms_backsolve, st.bytes_backsolve), (apply_name, st.ms_apply, st.bytes_apply))
                if st.ms_expand > 0:          # staged output (small primes): the entries of S are written by a third kernel
                    parts += ((st.kernel_expand.decode(), st.ms_expand, st.bytes_expand),)
            else:
                # algorithmic bytes of the reference's own algorithm on a dense x: per streamed
                # entry of U' 8 B read + 8 B read-modify-write of x; per elimination 16 B of row extent + 4 B
                # coefficient; per input/output entry 8 B; 20 B per reduced row
                algo = 16 * st.entries_streamed + 8 * (st.input_entries + st.nnz) + 20 * st.eliminations + 20 * st.rows
                name = st.kernel.decode()
                parts = ((name, st.ms_group if st.used_group_kernel and not st.group_aborted else st.ms_tier2, algo),
                         ("schur_lds_kernel<1024>", st.ms_tier0, 0), ("schur_lds_kernel<8192>", st.ms_tier1, 0))
            if st.ms_finalize > 0.05 and not st.used_sparse_image:          # (the image paths write S in its final place: no gather pass)
                parts += (("scan_* + gather_rows_kernel", st.ms_finalize, 16 * st.nnz + 12 * st.rows),)
            for name, m_, b_ in parts:
                ms[name] = ms.get(name, 0.0) + m_
                by[name] = b_
        torch.cuda.synchronize()
        return time.perf_counter() - t0, ms, by, st

    def timed(fn_steps):
        """barrier + synchronize, the steps, barrier + synchronize; seconds (the maximum over the ranks) and what fn_steps returned"""
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t_begin = time.perf_counter()
        got = fn_steps()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        seconds = time.perf_counter() - t_begin
        if dist is not None:
            t = torch.tensor([seconds], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            seconds = float(t.item())
        return seconds, got

    slab_only = None
    if product is not None:
        for _ in range(args.warmup):
            product_step()
        elapsed, _ = timed(lambda: [product_step() for _ in range(args.steps)])
        # the device-API step, for the kernels' own times (the product's call does not hand out its stage timers) and as the
        # second object: no gather of S, no stitching
        for _ in range(min(args.warmup, 2)):
            step()
        n_slab = max(3, min(5, args.steps))
        el_slab, (_, ms_sum, bytes_of, st) = timed(lambda: run_steps(n_slab))
        kernel_steps = n_slab
        slab_only = {"what": "device-API step of the same split: every rank its slab (or slice) through spasm_hip_dschur, %s -- what the product's "
                             "step adds is the all-gatherv of S and the stitching" % ("row lengths all-gathered" if split == "columns" else "all-gatherv of S"),
                     "steps": n_slab, "ms_per_step": 1e3 * el_slab / n_slab, "rows_per_s": len(rows) / (el_slab / n_slab)}
    else:
        for _ in range(args.warmup):
            step()
        elapsed, (_, ms_sum, bytes_of, st) = timed(lambda: run_steps(args.steps))
        kernel_steps = args.steps

    total_nnz_all = int(st.nnz)
    if use_dist and split == "columns":
        total_nnz_all = int(all_len.sum().item())          # entries of the whole Schur complement, from the exchanged lengths
    if product is not None:
        if total_nnz_all != product_nnz[0] and split == "columns":
            raise SystemExit("bench.py: the product's Schur complement holds %d entries, the slabs %d" % (product_nnz[0], total_nnz_all))
        total_nnz_all = product_nnz[0]
    if rank == 0:
        total_rows = len(rows)
        ms_per_step = 1e3 * elapsed / args.steps
        kernels = {}
        for name, total in ms_sum.items():
            k_ms = total / kernel_steps
            if k_ms <= 0.0005 or (k_ms < 0.02 and bytes_of[name] == 0):
                continue
            kernels[name] = {"ms": k_ms, "algorithmic_bytes": int(bytes_of[name]),
                             "GB_per_s": bytes_of[name] / (k_ms * 1e-3) / 1e9}
        dom_name = max(kernels, key=lambda k: kernels[k]["ms"])
        dom = kernels[dom_name]
        kernel_id = dom_name.split(" ")[0]
        traffic, traffic_hi, traffic_src = (None, None, None)
        if world == 1:
            traffic, traffic_hi, traffic_src = quoted_traffic(kernel_id, args.workload, total_rows)
        if world == 1:          # HBM bytes of the other kernels of the step, from the same counter passes
            for kname, kd in kernels.items():
                tr = quoted_traffic(kname.split(" ")[0], args.workload, total_rows)
                if tr[0]:
                    kd["traffic"] = tr[0]
                    kd["hbm_frac"] = tr[0] / (kd["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        step_bytes = sum(k["algorithmic_bytes"] for k in kernels.values())
        roof = {"bound": "hbm", "achieved": dom["GB_per_s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": dom["GB_per_s"] / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": kernel_id, "kernel_ms": dom["ms"], "algorithmic_bytes": dom["algorithmic_bytes"],
                "what_frac_is": "algorithmic bytes of this kernel (DESIGN.md section 4) / its device time / 8 TB/s",
                "hbm_frac": (traffic / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "hbm_frac_fetch_doubled": (traffic_hi / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic_hi else None,
                "traffic_source": traffic_src,
                "kernels": kernels,
                "step_algorithmic_bytes": int(step_bytes),
                "step_GB_per_s": step_bytes / (ms_per_step * 1e-3) / 1e9,
                "step_frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS}
        # The dense image's build is not bound by bytes: its workgroups -- one per slab of columns, 207 on mk13.b5 -- are bound
        # by what ONE compute unit issues.  From the SQ counters of profiles/r06_backsolve_sq.json (tools/profile_backsolve_sq.sh), when
        # they were recorded for this kernel: the time the vector instructions alone need on the compute units the kernel occupies.
        sq_path = os.path.join(ROOT, "profiles", "r06_backsolve_sq.json")
        if st.used_backsolve and os.path.exists(sq_path):
            try:
                sq = json.load(open(sq_path))["kernels"]
                key = [k for k in sq if k.replace(" ", "") == kernel_id.replace(" ", "")]
                if key:
                    pl = sq[key[0]]["per_launch"]
                    words = (int(A_full.m - F_full.U.n) + 1) // 2
                    wgs = min(256, (words + 11) // 12)
                    valu_ms = 1e3 * pl["SQ_INSTS_VALU"] * 4.0 / (wgs * 4) / 2.4e9          # a wave64 vector instruction holds its SIMD for four cycles
                    roof["issue_bound"] = {"what": "vector instructions of the kernel x 4 cycles / (4 SIMDs x the %d compute units its %d workgroups occupy) / 2.4 GHz" % (wgs, wgs),
                                           "valu_wave_instructions": pl["SQ_INSTS_VALU"], "lds_wave_instructions": pl.get("SQ_INSTS_LDS"), "compute_units_used": wgs,
                                           "valu_bound_ms": valu_ms, "frac_of_valu_bound": valu_ms / dom["ms"], "source": "profiles/r06_backsolve_sq.json"}
            except (ValueError, KeyError):
                pass
        if st.used_backsolve and st.bytes_staged > 0:
            roof["staged_bytes"] = int(st.bytes_staged)
            roof["staged_bytes_note"] = "packed rows of S between the apply and the expansion kernels: written once, read once; not algorithmic bytes"

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
            "dtype": "i32 sums of i16 x i16 products" if (st.used_backsolve and "s16" in (st.kernel.decode() + st.kernel_other.decode())) else "u32",
            "data": ("synthetic (%s regenerated from the definition of the matching complex; no network)" % args.workload)
            if source == "generated" else source,
            "config": {"workload": "%s (%dx%d, %d nnz) mod %d, round-0 Schur complement of %d non-pivotal rows "
                                   "w.r.t. %d structural pivots" % (args.workload, A_full.n, A_full.m, A_full.nnz, PRIME,
                                                                   total_rows, F.U.n),
                       "rows_per_step": total_rows, "pivots": int(F.U.n), "levels": dF.levels,
                       "non_pivotal_columns": int(A_full.m - F_full.U.n), "schur_nnz": total_nnz_all,
                       "path": "sparse image" if st.used_sparse_image else "back-substituted factor image" if st.used_backsolve else "row-by-row elimination",
                       "why_this_workload": why,
                       "sharding": ("spasm_hip_schur with the communicator installed (%d rank(s)): slabs of the non-pivotal columns (%d of %d here), "
                                    "all-gatherv of S (%d entries) + stitching in the step" % (world, A.m - F.U.n, A_full.m - F_full.U.n, total_nnz_all)) if (split == "columns" and product is not None)
                       else "rows over %d rank(s)%s" % (world, ", spasm_hip_schur with the communicator installed: all-gatherv of S in the step" if product is not None else "")},
            "roofline": roof,
            "factor_image_ms": statistics.median(image_ms),
            "factor_image_ms_all": image_ms,
            "factor_image_note": "host planning + upload of the factor image (spasm_hip_dfact_create), once per factor, NOT in a step; "
                                 "median of 3 builds after a first one that is not counted (it loads the code object and fills the buffer cache)",
            "rows_per_s_cold": total_rows / (elapsed / args.steps + 1e-3 * statistics.median(image_ms)),
        }
        if slab_only is not None:
            out["dist_product_path"] = {"what": "value / ms_per_step: spasm_hip_schur_resident = spasm_hip_schur as the driver calls it, communicator installed "
                                                "(slab or slice, all-gatherv of S, stitching); roofline.kernels: measured in the slab_only steps",
                                        "ms_per_step": ms_per_step, "slab_only_ms_per_step": slab_only["ms_per_step"], "slab_only_rows_per_s": slab_only["rows_per_s"],
                                        "schur_nnz": total_nnz_all, "split": split}
            out["slab_only"] = slab_only
        extras = world == 1 and not args.no_extras and product is None
        if extras and st.used_backsolve:
            # the same batch through the row-by-row elimination kernels (what round 1 measured), a few steps
            os.environ["SPASM_HIP_BACKSOLVE"] = "0"
            try:
                for _ in range(2):
                    step()
                n_alt = max(3, min(5, args.steps))
                el2, ms2, by2, st2 = run_steps(n_alt)
            finally:
                os.environ.pop("SPASM_HIP_BACKSOLVE", None)
            name2 = st2.kernel.decode()
            k_ms2 = ms2[name2] / n_alt
            out["row_by_row_path"] = {
                "what": "same batch with SPASM_HIP_BACKSOLVE=0 (%d steps)" % n_alt, "ms_per_step": 1e3 * el2 / n_alt,
                "rows_per_s": total_rows / (el2 / n_alt), "kernel": name2, "kernel_ms": k_ms2,
                "eliminations_per_step": int(st2.eliminations), "entries_streamed": int(st2.entries_streamed),
                "group_pivots": int(st2.group_pivots),
                "lane_efficiency": (st2.eliminations / (64.0 * st2.group_pivots)) if st2.group_pivots else None,
                "algorithmic_bytes": int(by2[name2]), "GB_per_s": by2[name2] / (k_ms2 * 1e-3) / 1e9,
                "frac": by2[name2] / (k_ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "same_nnz": bool(st2.nnz == st.nnz)}
            tr2 = quoted_traffic(name2, args.workload, total_rows)
            if tr2[0]:
                out["row_by_row_path"].update({"traffic": tr2[0], "hbm_frac": tr2[0] / (k_ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                               "traffic_source": tr2[2]})
        if extras:
            out["dense_tail"] = dense_tail_probe(torch, spasm_amd, dev)
            out["dense_tail_real"] = dense_tail_real_probe(torch, spasm_amd, dev, dA, drows, dF, int(A_full.m - F_full.U.n))
            # the other half of the headline metric: wall-clock time of the whole rank computation (host I/O excluded:
            # the matrix is already in memory), default options of tools/rank, five calls
            os.environ.pop("SPASM_HIP_THREADS", None)
            # (the tools/rank options of the config: --dense-threshold x, --no-greedy-pivot-search)
            opts = spasm_amd.default_opts()
            rank_args = (workloads.config(args.workload) or {"rank_args": []})["rank_args"]
            for t, a in enumerate(rank_args):
                if a == "--dense-threshold":
                    opts.sparsity_threshold = float(rank_args[t + 1])
                if a == "--no-greedy-pivot-search":
                    opts.enable_greedy_pivot_search = False
            def call():
                fact = spasm_amd.echelonize(A_full, opts)
                return spasm_amd.echelonize_profile(), int(fact.U.n)
            out["end_to_end"] = dict(_calls(call, 5), what="spasm_hip_echelonize on the same matrix, options of the config (%s), 5 calls"
                                                           % (" ".join(rank_args) or "defaults"))
            rounds = out["end_to_end"]["split_of_median_call"].get("sparse_rounds")
            out["config"]["note"] = ("device-API step (spasm_hip_dschur); spasm_hip_echelonize on this matrix runs %s sparse round(s)%s"
                                     % (rounds, ": S is dense, the call goes to the dense finish (summary.end_to_end)" if rounds == 0 else ""))
            # north_star keeps the pivot selection on the host; the library runs its greedy search on the device when it has
            # one.  The same call with the search where north_star puts it:
            os.environ["SPASM_HIP_PIVOT_SEARCH"] = "host"
            try:
                t0 = time.perf_counter()
                fact = spasm_amd.echelonize(A_full, opts)
                out["end_to_end"]["with_the_pivot_search_on_the_host"] = {
                    "what": "one more call with SPASM_HIP_PIVOT_SEARCH=host (%d threads)" % spasm_amd.usable_cpus(),
                    "seconds": time.perf_counter() - t0, "rank": int(fact.U.n), "split": spasm_amd.echelonize_profile()}
            finally:
                os.environ.pop("SPASM_HIP_PIVOT_SEARCH", None)
        if extras and args.workload == "mk13.b5":
            # (the largest object first: what the three paths of mk14.b4 park in the cache of device blocks -- 23 GB of dense R, 24 GB
            #  of accumulator scratch -- had to be given back to the device in the middle of the first mk15.b4 call of one run,
            #  2.6 s instead of 0.9-1.2; emptying the cache between the objects is worse: blocks that come back from the driver
            #  are paid for on first touch, 1.1 s for the pools of one sparse round.  Single calls still take 1-2 s longer now and
            #  then, in whichever stage asks for a large block first -- five calls per object, so that the median does not hang on one)
            # the flow of GL7d19 at its size (no dense image exists for this factor; the row-by-row kernels are left out: minutes)
            out["at_scale"] = sparse_path_probe(torch, spasm_amd, workloads, dev, name="mk15.b4", steps=2, paths=("default",), calls=5, cpu_rows=2000)
            out["sparse_path"] = sparse_path_probe(torch, spasm_amd, workloads, dev)
            out["stand_ins"] = stand_in_runs(spasm_amd, workloads)
        if extras and args.beyond_scale:
            Abig, _ = workloads.load_matrix("mk15.b5", PRIME)
            def call_big():
                fact = spasm_amd.echelonize(Abig)
                return spasm_amd.echelonize_profile(), int(fact.U.n)
            out["beyond_scale"] = dict(_calls(call_big, 2), what="spasm_hip_echelonize on mk15.b5 (%d x %d, %d entries: generated, beyond the size of GL7d19), defaults, 2 calls"
                                                                   % (Abig.n, Abig.m, Abig.nnz))
            del Abig
        out["configs"] = [{"name": c["name"], "status": status, "what": c["what"],
                           "bench": "python bench.py --workload %s" % c["name"],
                           "rank": "./tools/rank --matrix $SPASM_DATA/%s --modulus %d %s" % (c["file"], PRIME, " ".join(c["rank_args"]))}
                          for c, status, _ in workloads.discover()]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(A_full, rows, F_full)
            rt = cpu_rank_time(A_full)
            if rt:
                out["cpu_baseline"]["rank_time"] = rt
                if "end_to_end" in out:
                    out["end_to_end"]["cpu_rank_time"] = {"seconds": float("%.4g" % rt["seconds"]), "projected": rt["projected"], "cores": rt["cores"]}
        emit(out)
    if product is not None:
        product.close()
        comm.uninstall()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
