import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# the tests pick kernel variants and debugging aids by name: switches outside the supported list need this (spasm_hip.h, "Environment")
os.environ.setdefault("SPASM_HIP_EXPERIMENT", "1")

GOLDEN = os.path.join(ROOT, "tests", "golden")

# moduli and matrices of the reference's own suite (tests/CMakeLists.txt:45-52, 75-110)
ALL_MODULI = [3, 257, 65537, 67108859, 189812507, 4294967291]
ALL_TEST_MATRICES = [
    "BIOMD0000000424.int.mpl.sms", "cc.sms", "chains.sms", "dm.sms", "example.sms", "G2.sms",
    "lower_trapeze.sms", "mat364.sms", "p3.sms", "rectangular_l.sms", "scc3.sms", "singular2.sms",
    "singular.sms", "t1.sms", "upper_trapeze.sms", "BIOMD0000000525.int.mpl.sms", "chains2.sms",
    "dm2.sms", "empty.sms", "g1.sms", "l1.sms", "m1.sms", "medium.sms", "rectangular_h.sms",
    "scc2.sms", "scc.sms", "singular3.sms", "small.sms", "tree_test.sms", "trefethen_500.sms",
    "u1.sms", "void.sms",
]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def matrix_path(name):
    return os.path.join(GOLDEN, "Matrix", name)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.lib()
    return orc
