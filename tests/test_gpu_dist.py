"""The multi-GPU layer of the C ABI (spasm_amd/csrc/dist_api.hip) on ONE GPU: a world of one rank runs the same code --
communicator, size exchange, exact-count broadcasts, row-pointer rebasing, the sharded spasm_hip_schur and the driver on
top of it -- with the HIP kernels doing the compute (the gloo tests in test_dist_cpu.py cover world sizes 2 and 3 with a
stand-in for the compute)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, matrix_path

import spasm_amd
from spasm_amd.dist import Comm, echelonize_dist

pytestmark = pytest.mark.gpu


def _as_product(A):
    return spasm_amd.Csr(A.n, A.m, A.p, A.j, A.x, A.prime)


@pytest.fixture(scope="module")
def comm():
    c = Comm(0, 1)
    yield c
    c.close()


@pytest.mark.parametrize("how", ["p2p", "bcast"])
@pytest.mark.parametrize("name", ["mat364.sms", "medium.sms", "trefethen_500.sms", "void.sms"])
def test_allgatherv_of_a_world_of_one_is_the_identity(oracle, comm, name, how, monkeypatch):
    import torch
    monkeypatch.setenv("SPASM_HIP_ALLGATHERV", how)
    p = 42013
    A = oracle.load_sms(matrix_path(name), p)
    npiv, perm, F = oracle.pivots_extract_structural(A, oracle.empty_fact(A.n, A.m, p))
    rows = perm[npiv:]
    want, _, _ = oracle.schur(A, rows, F)
    dA = spasm_amd.DeviceCsr.from_host(_as_product(A))
    dF = spasm_amd.DeviceFact(spasm_amd.Fact(_as_product(F.U), F.qinv))
    W = spasm_amd.SchurWorkspace(max(len(rows), 1), A.m, 4 * want.nnz + (1 << 20))
    drows = torch.from_numpy(np.ascontiguousarray(rows, np.int32)).cuda()
    S, st = spasm_amd.dschur(dA, drows, dF, W, fetch=False)
    assert st.status == 0
    full = comm.allgatherv(W, A.m, p)
    H = full.to_host()
    assert oracle.same_matrix(oracle.CSR(H.n, H.m, H.p, H.j, H.x, p), want)


@pytest.mark.parametrize("name", ["mat364.sms", "medium.sms", "m1.sms", "singular.sms", "rectangular_l.sms"])
@pytest.mark.parametrize("p", [42013, 4294967291])
@pytest.mark.parametrize("split", ["rows", "columns"])
def test_echelonize_dist_shards_every_round(oracle, comm, name, p, split, monkeypatch):
    """spasm_hip_echelonize_dist: every Schur complement goes through the sharded path even in a world of one -- by rows (slice
    of the row list, all-gatherv on the device, download) and by columns (the slab problem of rank 0 of 1 = the whole problem:
    spasm_hip_column_slab, the one-GPU call on it, columns mapped back, all-gatherv of the slab, rows stitched on the device);
    same rank as the oracle, valid echelon form."""
    monkeypatch.setenv("SPASM_HIP_SHARD_FORCE", "1")
    monkeypatch.setenv("SPASM_HIP_SHARD_MIN_ROWS", "1")
    monkeypatch.setenv("SPASM_HIP_SHARD", split)
    A = oracle.load_sms(matrix_path(name), p)
    want = oracle.echelonize(A).U.n
    o = spasm_amd.default_opts()
    o.enable_dense = False              # keep the sparse rounds going: more sharded Schur complements
    o.enable_tall_and_skinny = False
    F = echelonize_dist(_as_product(A), comm, o)
    assert F.U.n == want
    seen = set()
    for i in range(F.U.n):
        jj, xx = F.U.row(i)
        assert xx[0] == 1 and int(jj[0]) not in seen and F.qinv[jj[0]] == i
        seen.add(int(jj[0]))


@pytest.mark.parametrize("split", ["rows", "auto"])
def test_sharded_driver_with_a_sparse_round_at_scale(comm, split, monkeypatch):
    """mk13.b4 (159,093 x 23,958 Schur complement, 4.4 % dense) with its sparse round forced and every Schur complement
    sharded: by rows (the slice is all-gathered on the device, kept there as the next round's A and downloaded once for the
    host pivot search) and the way the driver chooses by itself -- by COLUMNS, since this factor takes the sparse image: the
    slab is reduced through it, stacked by the all-gatherv, stitched on the device.  Rank = the CPU oracle's."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import workloads
    monkeypatch.setenv("SPASM_HIP_SHARD_FORCE", "1")
    if split != "auto":
        monkeypatch.setenv("SPASM_HIP_SHARD", split)
    A, _ = workloads.load_matrix("mk13.b4")
    o = spasm_amd.default_opts()
    o.sparsity_threshold = 0.1
    F = echelonize_dist(A, comm, o)
    prof = spasm_amd.echelonize_profile()
    assert F.U.n == 111463
    assert prof["sparse_rounds"] >= 1


def test_column_slabs_stay_slabs_until_somebody_needs_whole_rows(comm, monkeypatch):
    """Round 6: between two rounds of the driver a Schur complement computed by column slabs stays on the devices AS SLABS -- the
    census of leftmost entries is a minimum per row over the ranks, the dense finish sums the ranks' columns of every block -- and
    is gathered into whole rows only when another sparse round reads it.  mk13.b4 with its default flow (one sparse round, then the
    low-rank finish: nothing is ever gathered) and without the greedy search (two sparse rounds: the first complement is gathered
    for the second round, the second one is not).  World of one: the same code, collectives of one rank."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import workloads
    monkeypatch.setenv("SPASM_HIP_SHARD_FORCE", "1")
    A, _ = workloads.load_matrix("mk13.b4")
    o = spasm_amd.default_opts()
    o.sparsity_threshold = 0.1
    F = echelonize_dist(A, comm, o)
    ev = spasm_amd.echelonize_counters()
    assert F.U.n == 111463
    assert ev["schur_complements_kept_as_column_slabs"] >= 1 and ev["column_slabs_gathered_into_whole_rows"] == 0, ev
    o = spasm_amd.default_opts()
    o.enable_greedy_pivot_search = 0
    o.sparsity_threshold = 0.05
    F = echelonize_dist(A, comm, o)
    ev = spasm_amd.echelonize_counters()
    prof = spasm_amd.echelonize_profile()
    assert F.U.n == 111463 and prof["sparse_rounds"] >= 2
    assert ev["schur_complements_kept_as_column_slabs"] >= 2 and ev["column_slabs_gathered_into_whole_rows"] >= 1, ev


def test_bench_runs_its_rccl_path_on_one_gpu():
    """bench.py with the distributed path forced on (world of one), split by rows, on a small sibling of the bench matrix: the
    timed step is the product's (spasm_hip_schur_resident with the communicator installed: all-gatherv of S in the step), the
    device-API step comes as a second object."""
    env = dict(os.environ, SPASM_BENCH_FORCE_DIST="1", SPASM_HIP_VERBOSE="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29871",
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "mk11.b4", "--steps", "2", "--warmup", "1",
                          "--no-extras", "--no-cpu-baseline", "--split", "rows"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    assert len([x for x in out.stdout.splitlines() if x.strip()]) == 1, out.stdout[:2000]          # one line, the last thing on stdout
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["value"] > 0 and "all-gatherv" in d["config"]["sharding"]
    prod = d["summary"]["dist_product_path"]
    assert prod["split"] == "rows" and prod["ms_per_step"] >= prod["slab_only_ms_per_step"] * 0.5 and prod["schur_nnz"] == d["config"]["schur_nnz"] > 0


def test_bench_runs_its_column_split_on_one_gpu():
    """bench.py --split columns with the distributed path forced on (world of one): the slab problem of rank 0 of 1 is the
    whole problem.  The timed step is the product's: the slab reduced and kept on the device, all-gatherv, stitching; its entry
    count must equal what the slab-only steps (row lengths through the RCCL all-gather) add up to -- bench.py checks it."""
    env = dict(os.environ, SPASM_BENCH_FORCE_DIST="1", SPASM_HIP_VERBOSE="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29873",
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "mk11.b4", "--steps", "2", "--warmup", "1",
                          "--no-extras", "--no-cpu-baseline", "--split", "columns"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["value"] > 0 and "slabs of the non-pivotal columns" in d["config"]["sharding"] and d["config"]["schur_nnz"] > 0
    assert d["summary"]["dist_product_path"]["split"] == "columns"


@pytest.mark.parametrize("name,parts", [("mk12.b4", 4), ("mk11.b4", 8), ("mk12.b3", 3)])
def test_column_slabs_on_the_gpu_stitch_to_the_full_schur_complement(name, parts):
    """The split that fits the back-substituted path, in a world of one: the slab problems of all `parts` ranks
    (spasm_hip_column_slab) are reduced one after the other by the HIP kernels -- each builds only ITS columns of R -- and
    their rows, mapped back and concatenated, must be the Schur complement of the whole problem entry for entry (values
    included: arithmetic mod p is exact).  What N ranks would each hold is exactly one of these pieces."""
    import torch
    from spasm_amd.dist import column_slab, stitch_column_slabs
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import workloads
    p = 42013
    A, rows, F, _ = workloads.round0(name, p)
    os.environ["SPASM_HIP_BACKSOLVE"] = "1"
    try:
        def reduce(Ax, Fx):
            dA = spasm_amd.DeviceCsr.from_host(Ax)
            dF = spasm_amd.DeviceFact(Fx)
            drows = torch.from_numpy(np.ascontiguousarray(rows, np.int32)).cuda()
            pool = 1 << 24
            while True:
                W = spasm_amd.SchurWorkspace(len(rows), Ax.m, pool)
                S, st = spasm_amd.dschur(dA, drows, dF, W, fetch=True)
                if st.status == 0:
                    break
                W.close()
                pool *= 4
            assert st.used_backsolve == 1
            H = S.to_host()
            W.close()
            dF.close()
            return H
        want = reduce(A, F)
        pieces, maps = [], []
        for k in range(parts):
            As, Fs, cols = column_slab(A, F, k, parts)
            pieces.append(reduce(As, Fs))
            maps.append(cols)
        full = stitch_column_slabs(pieces, maps, len(rows), A.m, p)
    finally:
        os.environ.pop("SPASM_HIP_BACKSOLVE", None)
    assert np.array_equal(full.p, want.p) and np.array_equal(full.j, want.j) and np.array_equal(full.x, want.x)


@pytest.mark.parametrize("name,parts", [("mk12.b4", 4), ("mk12.b3", 3)])
def test_slabs_stitched_on_the_device(name, parts):
    """spasm_hip_dstitch_slabs -- what the column split of the driver runs after its all-gatherv: the slabs of all `parts` ranks,
    computed here one after the other (sparse image or dense image, whatever the library takes), columns mapped back, STACKED
    as the all-gatherv leaves them, stitched by the device kernel: entry for entry the Schur complement of the whole problem."""
    import torch
    from spasm_amd.dist import column_slab
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import workloads
    p = 42013
    A, rows, F, _ = workloads.round0(name, p)
    drows = torch.from_numpy(np.ascontiguousarray(rows, np.int32)).cuda()

    def reduce(Ax, Fx):
        dA = spasm_amd.DeviceCsr.from_host(Ax)
        dF = spasm_amd.DeviceFact(Fx)
        pool = 1 << 24
        while True:
            W = spasm_amd.SchurWorkspace(len(rows), Ax.m, pool)
            S, st = spasm_amd.dschur(dA, drows, dF, W, fetch=True)
            if st.status == 0:
                break
            W.close()
            pool *= 4
        W.close()
        dF.close()
        return S, st
    want, st = reduce(A, F)
    n = len(rows)
    sp, sj, sx = [], [], []
    base = 0
    for k in range(parts):
        As, Fs, cols = column_slab(A, F, k, parts)
        S, st = reduce(As, Fs)
        cmap = torch.from_numpy(np.ascontiguousarray(cols, np.int32)).cuda()
        sp.append(S.p[:-1] + base)
        sj.append(cmap[S.j[:st.nnz].long()])
        sx.append(S.x[:st.nnz])
        base += int(st.nnz)
    gSp = torch.cat(sp + [torch.tensor([base], dtype=torch.int64, device="cuda")])
    gSj = torch.cat(sj).contiguous()
    gSx = torch.cat(sx).contiguous()
    assert base == int(want.p[-1].item())
    Sp = torch.zeros(n + 1, dtype=torch.int64, device="cuda")
    Sj = torch.zeros(max(base, 1), dtype=torch.int32, device="cuda")
    Sx = torch.zeros(max(base, 1), dtype=torch.int32, device="cuda")
    rc = spasm_amd.lib().spasm_hip_dstitch_slabs(gSp.data_ptr(), gSj.data_ptr(), gSx.data_ptr(), n, parts, Sp.data_ptr(), Sj.data_ptr(), Sx.data_ptr(), base, 0)
    assert rc == 0
    assert torch.equal(Sp, want.p) and torch.equal(Sj[:base], want.j[:base]) and torch.equal(Sx[:base], want.x[:base])
    rc = spasm_amd.lib().spasm_hip_dstitch_slabs(gSp.data_ptr(), gSj.data_ptr(), gSx.data_ptr(), n, parts, Sp.data_ptr(), Sj.data_ptr(), Sx.data_ptr(), base - 1, 0)
    assert rc == 1
