"""Workloads of BASELINE.json: where they come from and how tools/rank prepares them.

Five matrices are named there.  One of them (mk13.b5) has a closed-form
definition and is regenerated here; the other four are data files of the
hpac / SuiteSparse collections that cannot be fetched in this environment.
They are looked up under $SPASM_DATA (default: tests/data/) as
<name>.sms[.gz]; `discover()` says which are present.  Nothing is ever
substituted silently: an absent file is reported as absent.

Naming of the matching-complex boundaries (hpac "Homology/mk"): mkN.bK maps the
(K+1)-edge matchings of the complete graph K_N (rows) to its K-edge matchings
(columns), K+1 entries +-1 per row.  Checked against the published sizes:
mk9.b1 378 x 36 (756 nnz), mk12.b3 51975 x 13860 (207900 nnz), mk13.b5
135135 x 270270 (810810 nnz).
"""
import gzip
import itertools
import os
import shutil
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRIME = 42013

# BASELINE.json "configs", in order.  `file` is the data file looked up under $SPASM_DATA; `generator`
# regenerates the matrix when the file is absent; `rank_args` are the tools/rank options of the config.
CONFIGS = [
    {"name": "kneser_10_4_1", "file": "kneser_10_4_1.sms", "generator": None, "rank_args": [],
     "what": "tools/rank plumbing (CPU reference config)"},
    {"name": "mk13.b5", "file": "mk13.b5.sms", "generator": ("mk", 13, 5), "rank_args": [],
     "what": "sparse Schur complement, single GPU (the bench line)"},
    {"name": "GL7d19", "file": "GL7d19.sms", "generator": None, "rank_args": ["--dense-threshold", "0.01"],
     "what": "sparse rounds + dense tail on the matrix cores"},
    {"name": "relat9", "file": "relat9.sms", "generator": None, "rank_args": [],
     "what": "Schur row batches sharded over 8 GPUs"},
    {"name": "M0,6-D9", "file": "M0,6-D9.sms", "generator": None, "rank_args": ["--no-greedy-pivot-search"],
     "what": "fill-in stress case, greedy pivot search disabled, 8 GPUs"},
]


def data_dir():
    return os.environ.get("SPASM_DATA", os.path.join(ROOT, "tests", "data"))


def find_data(filename):
    """path of <filename> or <filename>.gz under $SPASM_DATA, or None."""
    for cand in (filename, filename + ".gz"):
        path = os.path.join(data_dir(), cand)
        if os.path.exists(path):
            return path
    return None


def config(name):
    for c in CONFIGS:
        if c["name"] == name:
            return c
    return None


def discover():
    """[(config, 'file' | 'generated' | 'absent', path or None)] for every BASELINE config."""
    out = []
    for c in CONFIGS:
        path = find_data(c["file"])
        if path is not None:
            out.append((c, "file", path))
        elif c["generator"] is not None:
            out.append((c, "generated", None))
        else:
            out.append((c, "absent", None))
    return out


# --------------------------------------------------------------------------
# matching complexes
# --------------------------------------------------------------------------
def _matchings(nv, size):
    edges = list(itertools.combinations(range(nv), 2))
    masks = [(1 << a) | (1 << b) for a, b in edges]
    ne = len(edges)
    out, cur = [], []

    def rec(start, used):
        if len(cur) == size:
            out.append(tuple(cur))
            return
        for e in range(start, ne):
            if masks[e] & used:
                continue
            cur.append(e)
            rec(e + 1, used | masks[e])
            cur.pop()
    rec(0, 0)
    return out


def _matchings_vectorised(nv, size):
    """all `size`-edge matchings of K_nv as an array (count, size) of edge numbers, rows in the lexicographic order of
    _matchings() (edges numbered like itertools.combinations(range(nv), 2)), level by level with numpy: the matchings of
    size s + 1 are the matchings of size s extended by a later, disjoint edge.  mk15's 2.8 million 5-matchings take seconds."""
    edges = np.array(list(itertools.combinations(range(nv), 2)), np.int64)
    ne = len(edges)
    emask = (1 << edges[:, 0]) | (1 << edges[:, 1])
    cur = np.zeros((1, 0), np.int16)
    used = np.zeros(1, np.int64)
    last = np.full(1, -1, np.int64)
    for _ in range(size):
        rows, cols = [], []
        step = max(1, (1 << 24) // max(ne, 1))
        for lo in range(0, len(cur), step):
            hi = min(len(cur), lo + step)
            ok = ((used[lo:hi, None] & emask[None, :]) == 0) & (np.arange(ne)[None, :] > last[lo:hi, None])
            r, c = np.nonzero(ok)                      # row-major: matching-major, edge-minor = lexicographic
            rows.append(r + lo)
            cols.append(c)
        r = np.concatenate(rows)
        c = np.concatenate(cols)
        cur = np.concatenate([cur[r], c[:, None].astype(np.int16)], axis=1)
        used = used[r] | emask[c]
        last = c.astype(np.int64)
    return cur, ne


def mk_boundary(nv, K):
    """mk<nv>.b<K> in its published orientation: rows = (K+1)-edge matchings of K_nv, columns = K-edge
    matchings, entry (-1)^t for the face that drops the t-th edge.  Returns (n, m, ti, tj, tx).  Vectorised (numpy);
    tests/test_host.py checks it against the recursive enumeration on the small members of the family."""
    big, ne = _matchings_vectorised(nv, K + 1)
    small, _ = _matchings_vectorised(nv, K)
    # a matching as one number: its edges as digits in base ne (the lexicographic order is the numeric one)
    def key(a):
        k = np.zeros(len(a), np.int64)
        for t in range(a.shape[1]):
            k = k * ne + a[:, t].astype(np.int64)
        return k
    skey = key(small)
    n = len(big)
    ti = np.repeat(np.arange(n, dtype=np.int32), K + 1)
    tj = np.empty((n, K + 1), np.int32)
    for t in range(K + 1):
        face = np.delete(big, t, axis=1)
        pos = np.searchsorted(skey, key(face))
        tj[:, t] = pos
    tx = np.tile(np.array([1 if t % 2 == 0 else -1 for t in range(K + 1)], np.int64), n)
    return n, len(small), ti, tj.reshape(-1), tx


def mk_boundary_reference(nv, K):
    """the same by the recursive enumeration (slow; the check of the vectorised generator)"""
    big = _matchings(nv, K + 1)
    small = _matchings(nv, K)
    index = {s: i for i, s in enumerate(small)}
    ti, tj, tx = [], [], []
    for r, s in enumerate(big):
        for t in range(K + 1):
            ti.append(r)
            tj.append(index[s[:t] + s[t + 1:]])
            tx.append(1 if t % 2 == 0 else -1)
    return len(big), len(small), np.array(ti, np.int32), np.array(tj, np.int32), np.array(tx, np.int64)


# --------------------------------------------------------------------------
# chessboard complexes (STAND-INS, never BASELINE configs)
# --------------------------------------------------------------------------
def _rook_placements(nr, nc, size):
    """non-attacking placements of `size` rooks on an nr x nc board, each as a tuple of (row, column) cells sorted
    by row; the list is in lexicographic order."""
    out = []
    for rows in itertools.combinations(range(nr), size):
        for cols in itertools.permutations(range(nc), size):
            out.append(tuple(zip(rows, cols)))
    return out


def ch_boundary(nr, nc, K):
    """ch<nr>-<nc>.b<K> (hpac "Homology/ch"): rows = placements of K+1 non-attacking rooks on an nr x nc board,
    columns = placements of K rooks, entry (-1)^t for the face that drops the t-th rook.  Checked against the
    published sizes: ch7-8.b5 141120 x 141120 (846720 nnz), ch8-8.b5 564480 x 376320 (3386880 nnz)."""
    big = _rook_placements(nr, nc, K + 1)
    small = _rook_placements(nr, nc, K)
    index = {s: i for i, s in enumerate(small)}
    n = len(big)
    ti = np.repeat(np.arange(n, dtype=np.int32), K + 1)
    tj = np.empty(n * (K + 1), np.int32)
    tx = np.tile(np.array([1 if t % 2 == 0 else -1 for t in range(K + 1)], np.int64), n)
    pos = 0
    for s in big:
        for t in range(K + 1):
            tj[pos] = index[s[:t] + s[t + 1:]]
            pos += 1
    return n, len(small), ti, tj, tx


# Stand-ins for the BASELINE matrices whose files cannot be fetched: same hpac "Homology" collection, closed-form,
# and -- unlike the matching complexes -- their Schur complements stay sparse for several elimination rounds, which
# is the flow GL7d19 takes (spasm_echelonize.c:525-580, then the dense tail at --dense-threshold 0.01).
STAND_INS = {
    "ch7-8.b5": {"for": "GL7d19", "rank_args": ["--dense-threshold", "0.01"], "shape": (141120, 141120), "nnz": 846720},
    "ch8-8.b5": {"for": "GL7d19", "rank_args": ["--dense-threshold", "0.01"], "shape": (564480, 376320), "nnz": 3386880},
}


def _triplets_of(name):
    kind = name.split(".")[0]
    if kind.startswith("mk") and ".b" in name:
        return mk_boundary(int(kind[2:]), int(name.split(".b")[1]))
    if kind.startswith("ch") and "-" in kind and ".b" in name:
        nr, nc = kind[2:].split("-")
        return ch_boundary(int(nr), int(nc), int(name.split(".b")[1]))
    raise ValueError("no generator for %s" % name)


def _open_plain(path):
    """SMS files may be gzipped: the C loader wants a plain file."""
    if not path.endswith(".gz"):
        return path, None
    tmp = tempfile.NamedTemporaryFile(prefix="spasm_amd_", suffix=".sms", delete=False)
    with gzip.open(path, "rb") as src:
        shutil.copyfileobj(src, tmp)
    tmp.close()
    return tmp.name, tmp.name


def load_matrix(name, prime=PRIME, tall=True):
    """(A, source): the matrix as tools/rank works on it (tools/rank.c:76-92: load, transpose when
    n < m unless asked not to).  source = 'file:<path>' or 'generated'.  Raises FileNotFoundError for a
    BASELINE matrix that is neither on disk nor regenerable."""
    import spasm_amd
    c = config(name)
    path = find_data(c["file"] if c else name + ".sms")          # (any mkN.bK can be generated: small siblings for tests)
    if path is not None:
        plain, tmp = _open_plain(path)
        try:
            A = spasm_amd.load(plain, prime, transpose_if_wide=tall)
        finally:
            if tmp:
                os.unlink(tmp)
        return A, "file:" + path
    if c is not None and c["generator"] is None:
        raise FileNotFoundError("%s not found under %s (set SPASM_DATA)" % (c["file"], data_dir()))
    n, m, ti, tj, tx = _triplets_of(name)
    if tall and n < m:
        ti, tj = tj, ti
        n, m = m, n
    return _csr_of_triplets(prime, n, m, ti, tj, tx), "generated"


def _csr_of_triplets(prime, n, m, ti, tj, tx):
    """what spasm_compress (spasm_triplet.c:97) makes of triplets without duplicates or zeros -- rows in order, entries of
    a row in triplet order -- without five million calls through ctypes (checked against spasm_hip_compress in
    tests/test_host.py)."""
    import spasm_amd
    ti = np.asarray(ti)
    order = np.argsort(ti, kind="stable")
    p = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(ti, minlength=n), out=p[1:])
    x = np.asarray(tx, np.int64)[order] % prime
    x = np.where(x > prime // 2, x - prime, x).astype(np.int32)
    return spasm_amd.Csr(n, m, p, np.ascontiguousarray(np.asarray(tj)[order], np.int32), x, prime)


def round0(name, prime=PRIME, cache=True, threads=1, labelled=False):
    """(A, rows, F, source): the matrix, its structural pivots (single-threaded search by default: the same pivots on
    every rank and in every run; threads=0: the library's default thread count -- the pivots then depend on timing, which
    is fine for a timing run on one process) and the non-pivotal rows -- the input of the first Schur complement.
    threads=1: the row-order search whose outcome is the reference's with one thread (the pivot set of the headline step
    since round 1: 140,087 rows on mk13.b5) -- or, labelled=True, the sequential search with depth labels (round 5: as
    deterministic, another valid pivot set, 10-20x faster: what fixes the pivot set of the large stand-ins)."""
    import spasm_amd
    path = os.path.join(tempfile.gettempdir(), "spasm_amd_r0_v4_%s_%d%s.npz" % (name.replace("/", "_"), prime, ("_lab" if labelled else "") if threads == 1 else "_mt"))
    if cache and os.path.exists(path):
        z = np.load(path, allow_pickle=False)
        A = spasm_amd.Csr(int(z["n"]), int(z["m"]), z["Ap"], z["Aj"], z["Ax"], prime)
        F = spasm_amd.Fact(spasm_amd.Csr(int(z["r"]), int(z["m"]), z["Up"], z["Uj"], z["Ux"], prime), z["qinv"])
        return A, z["rows"], F, str(z["source"])
    A, source = load_matrix(name, prime)
    # (both switches are on the supported list of include/spasm_hip.h: read at every call, with or without SPASM_HIP_EXPERIMENT)
    saved = {k: os.environ.get(k) for k in ("SPASM_HIP_THREADS", "SPASM_HIP_PIVOT_LABELS")}
    if threads > 0:
        os.environ["SPASM_HIP_THREADS"] = str(threads)
    else:
        os.environ.pop("SPASM_HIP_THREADS", None)
    if threads == 1 and not labelled:
        os.environ["SPASM_HIP_PIVOT_LABELS"] = "0"
    search = "device / threaded (timing-dependent)" if threads != 1 else "sequential, depth labels" if labelled else "sequential, row order (the reference's single-thread set)"
    try:
        npiv, perm, F = spasm_amd.pivots_extract_structural(A, spasm_amd.empty_fact(A.m, prime))
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    rows = np.ascontiguousarray(perm[npiv:], np.int32)
    if cache:
        tmp = "%s.%d.tmp.npz" % (path, os.getpid())          # ranks build concurrently: publish atomically
        np.savez(tmp, n=A.n, m=A.m, Ap=A.p, Aj=A.j, Ax=A.x, r=F.U.n, Up=F.U.p, Uj=F.U.j, Ux=F.U.x, qinv=F.qinv,
                 rows=rows, source=np.array(source), search=np.array(search))
        os.replace(tmp, path)
    return A, rows, F, source


def save_sms(A, path):
    """writes a Csr as SMS (1-based triplets, terminated by 0 0 0)."""
    with open(path, "w") as f:
        f.write("%d %d M\n" % (A.n, A.m))
        for i in range(A.n):
            for px in range(A.p[i], A.p[i + 1]):
                f.write("%d %d %d\n" % (i + 1, A.j[px] + 1, A.x[px]))
        f.write("0 0 0\n")
