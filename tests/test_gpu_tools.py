"""The command line tools against the oracle: tools/rank (this repository's), and the reference's own tools/rank.c,
tools/echelonize.c and tools/kernel.c compiled unmodified against the library (oracle/Makefile `dropin`)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, matrix_path

pytestmark = pytest.mark.gpu

RANK = os.path.join(ROOT, "tools", "rank")


REF_BIN = os.path.join(ROOT, "oracle", "_ref")


def _ref_tool(name):
    path = os.path.join(REF_BIN, name)
    if not os.path.exists(path):
        pytest.skip("%s was not built (needs the reference tree at build time)" % name)
    return path


def _need_tools():
    if not os.path.exists(RANK):
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
    """the reference's tools/echelonize.c (through the shim header): stdin -> stdout, SMS in, SMS out (U, or the RREF with
    --rref); the output spans the same row space."""
    ECHELONIZE = _ref_tool("ref_echelonize_shim")
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


# --------------------------------------------------------------------------
# the reference's own tool sources, unmodified, bound to the HIP library (oracle/Makefile `dropin`)
# --------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["mat364.sms", "medium.sms", "rectangular_h.sms", "singular.sms"])
def test_reference_rank_c_linked_against_the_facade(oracle, name):
    """tools/rank.c of the reference, unmodified: its spasm_echelonize is the GPU one."""
    tool = _ref_tool("ref_rank_facade")
    A = oracle.load_sms(matrix_path(name), 42013)
    want = oracle.echelonize(A).U.n
    env = dict(os.environ, SPASM_HIP_VERBOSE="0")
    out = subprocess.run([tool, "--matrix", matrix_path(name), "--modulus", "42013"], capture_output=True, text=True, env=env,
                         timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert ("rank = %d" % want) in out.stderr


def test_reference_rank_c_certificate_through_the_facade(oracle, tmp_path):
    """--certificate: the reference's certificate code (spasm_certificate.c) on top of the L and U of the GPU echelonization,
    verified by the reference's own spasm_certificate_rank_verify (tools/rank.c:107-128)."""
    tool = _ref_tool("ref_rank_facade")
    env = dict(os.environ, SPASM_HIP_VERBOSE="0")
    out = subprocess.run([tool, "--matrix", matrix_path("mat364.sms"), "--modulus", "42013", "--certificate"],
                         capture_output=True, text=True, env=env, timeout=300, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-3000:]
    assert "rank = " in out.stderr


@pytest.mark.parametrize("name", ["mat364.sms", "singular2.sms", "rectangular_l.sms"])
def test_reference_echelonize_c_compiled_with_the_shim(oracle, name):
    tool = _ref_tool("ref_echelonize_shim")
    A = oracle.load_sms(matrix_path(name), 42013)
    want = oracle.echelonize(A).U.n
    env = dict(os.environ, SPASM_HIP_VERBOSE="0")
    with open(matrix_path(name)) as f:
        out = subprocess.run([tool], stdin=f, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    n, m, kind = out.stdout.split("\n")[0].split()
    assert (int(n), int(m)) == (want, A.m)


def test_reference_kernel_c_linked_against_the_facade(oracle):
    """tools/kernel.c: spasm_echelonize + spasm_kernel from the GPU library; K * A^T == 0 is checked on the output."""
    tool = _ref_tool("ref_kernel_facade")
    p = 42013
    name = "singular.sms"
    A = oracle.load_sms(matrix_path(name), p)
    env = dict(os.environ, SPASM_HIP_VERBOSE="0")
    with open(matrix_path(name)) as f:
        out = subprocess.run([tool, "--modulus", str(p)], stdin=f, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().split("\n")
    kn, km, _ = lines[0].split()
    assert int(kn) > 0
