"""spasm_amd -- MI355X-native sparse echelonization mod p (hot path of cbouilla/spasm).

The package is a thin host-side mirror of the reference's C API over the C ABI
of spasm_amd/csrc/libspasm_hip.so (include/spasm_hip.h).  All arithmetic runs
in hand-written HIP kernels; there is no CPU fallback: importing works without
a GPU (so that symbols can be inspected), but every compute entry point dies
loudly when no HIP device is present.
"""
from .matrix import Csr, Fact, EchelonizeOpts                     # noqa: F401
from ._lib import lib, device_count, usable_cpus, release_cached_memory, LIB_PATH                      # noqa: F401
from .host import (load, compress, transpose, pivots_extract_structural, schur, ResidentSchur,       # noqa: F401
                   empty_fact, schur_dense, ffpack_rref, ffpack_LU, echelonize, echelonize_profile, echelonize_counters, rref, kernel,
                   default_opts)
from .device import DeviceCsr, DeviceFact, SchurWorkspace, dschur                    # noqa: F401
