"""world_size-2 gloo test of the row sharding + all-gatherv reassembly (no GPU).

The per-shard compute is a stand-in (the oracle) injected through
sharded_schur's reduce_rows hook: what is under test is the N > 1 plumbing.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, matrix_path


def _worker(rank, world, port, name, prime, result_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from spasm_amd.device import DeviceCsr
    from spasm_amd.dist import sharded_schur, shard_bounds
    A = orc.load_sms(matrix_path(name), prime)
    npiv, perm, F = orc.pivots_extract_structural(A, orc.empty_fact(A.n, A.m, prime))
    rows = perm[npiv:]

    def reduce_rows(mine):
        S, _, _ = orc.schur(A, mine, F)
        return DeviceCsr(S.n, S.m, torch.from_numpy(S.p), torch.from_numpy(S.j if S.nnz else np.zeros(1, np.int32)),
                         torch.from_numpy(S.x if S.nnz else np.zeros(1, np.int32)), prime)

    full = sharded_schur(None, rows, None, None, dist, reduce_rows)
    want, _, _ = orc.schur(A, rows, F)
    nnz = int(full.p[full.n])
    ok = (full.n == want.n and np.array_equal(full.p.numpy(), want.p)
          and np.array_equal(full.j.numpy()[:nnz], want.j) and np.array_equal(full.x.numpy()[:nnz], want.x))
    lo, hi = shard_bounds(len(rows), rank, world)
    ok = ok and (hi - lo) in (len(rows) // world, len(rows) // world + 1)
    open(os.path.join(result_dir, "rank%d" % rank), "w").write("ok" if ok else "FAIL")
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("mat364.sms", 2), ("medium.sms", 2), ("small.sms", 3), ("void.sms", 2)])
def test_sharded_schur_gloo(tmp_path, name, world):
    port = 29500 + (os.getpid() + hash(name)) % 2000
    mp.spawn(_worker, args=(world, port, name, 42013, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(os.path.join(str(tmp_path), "rank%d" % r)).read() == "ok"


def test_shard_bounds_cover():
    from spasm_amd.dist import shard_bounds
    for n in (0, 1, 7, 64, 1001):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
