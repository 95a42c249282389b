"""The drop-in command line tools (tools/rank, tools/echelonize) against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, matrix_path

pytestmark = pytest.mark.gpu

RANK = os.path.join(ROOT, "tools", "rank")
ECHELONIZE = os.path.join(ROOT, "tools", "echelonize")


def _need_tools():
    if not (os.path.exists(RANK) and os.path.exists(ECHELONIZE)):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tools")], check=True)


@pytest.mark.parametrize("name", ["mat364.sms", "medium.sms", "rectangular_h.sms", "singular.sms", "void.sms"])
@pytest.mark.parametrize("args", [[], ["--modulus", "65537"], ["--no-transpose", "--dense-threshold", "0.01"],
                                  ["--no-low-rank-mode", "--max-iterations", "1"]])
def test_rank_tool(oracle, name, args):
    _need_tools()
    p = 65537 if "--modulus" in args else 42013
    A = oracle.load_sms(matrix_path(name), p)
    want = oracle.echelonize(A).U.n
    env = dict(os.environ, SPASM_HIP_VERBOSE="0")
    out = subprocess.run([RANK, "--matrix", matrix_path(name)] + args, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert int(out.stdout.strip().split()[-1]) == want
    assert ("rank = %d" % want) in out.stderr


@pytest.mark.parametrize("name", ["mat364.sms", "singular2.sms", "rectangular_l.sms"])
def test_echelonize_tool_outputs_sms(oracle, name):
    """stdin -> stdout, SMS in, SMS out (U, or the RREF with --rref); the output spans the same row space."""
    _need_tools()
    p = 42013
    A = oracle.load_sms(matrix_path(name), p)
    want = oracle.echelonize(A).U.n
    env = dict(os.environ, SPASM_HIP_VERBOSE="0")
    for extra in ([], ["--rref"]):
        with open(matrix_path(name)) as f:
            out = subprocess.run([ECHELONIZE] + extra, stdin=f, capture_output=True, text=True, env=env, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = out.stdout.strip().split("\n")
        n, m, kind = lines[0].split()
        assert (int(n), int(m), kind) == (want, A.m, "M") and lines[-1].split() == ["0", "0", "0"]
        ti, tj, tx = [], [], []
        for line in lines[1:-1]:
            i, j, x = line.split()
            ti.append(int(i) - 1)
            tj.append(int(j) - 1)
            tx.append(int(x))
        U = oracle.compress(p, want, A.m, np.array(ti, np.int32), np.array(tj, np.int32), np.array(tx, np.int64))
        # every row starts with a unit pivot on a fresh column, and rowspan(A) is inside rowspan(U)
        qinv = np.full(A.m, -1, np.int32)
        for i in range(U.n):
            jj, xx = U.row(i)
            assert xx[0] == 1 and qinv[jj[0]] == -1
            qinv[jj[0]] = i
        for i in range(A.n):
            pat, x = oracle.solve_row(U, qinv, A, i)
            assert not any(x[j] != 0 and qinv[j] < 0 for j in pat)
