"""Loads spasm_amd/csrc/libspasm_hip.so and declares its C ABI (include/spasm_hip.h)."""
import ctypes as C
import os

from .matrix import CCsr, CTriplet, CLu, EchelonizeOpts, CDcsr, CSchurStats, CField

# SPASM_HIP_LIB: load another build of the same library (A/B runs of a kernel variant)
LIB_PATH = os.environ.get("SPASM_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libspasm_hip.so")

_lib = None


def lib():
    """the shared library; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "spasm_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C spasm_amd/csrc).  There is no CPU fallback." % LIB_PATH)
    # torch wheels bundle their own libamdhip64.so.7; it has to be the HIP runtime of the process
    # (two runtimes in one process cannot both own the GPU), so let torch load it first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    i64, ci, vp = C.c_int64, C.c_int, C.c_void_p
    pcsr, ptri, plu = C.POINTER(CCsr), C.POINTER(CTriplet), C.POINTER(CLu)
    pint = C.POINTER(C.c_int)
    sig = {
        "spasm_hip_device_count": (ci, []),
        "spasm_hip_usable_cpus": (ci, []),
        "spasm_hip_release_cached_memory": (None, []),
        "spasm_hip_version": (C.c_char_p, []),
        "spasm_hip_csr_alloc": (pcsr, [ci, ci, i64, i64, C.c_bool]),
        "spasm_hip_csr_free": (None, [pcsr]),
        "spasm_hip_triplet_alloc": (ptri, [ci, ci, i64, i64, C.c_bool]),
        "spasm_hip_triplet_free": (None, [ptri]),
        "spasm_hip_add_entry": (None, [ptri, ci, ci, i64]),
        "spasm_hip_triplet_transpose": (None, [ptri]),
        "spasm_hip_compress": (pcsr, [ptri]),
        "spasm_hip_transpose": (pcsr, [pcsr, ci]),
        "spasm_hip_triplet_load": (ptri, [vp, i64, C.POINTER(C.c_uint8)]),
        "spasm_hip_csr_save": (None, [pcsr, vp]),
        "spasm_hip_triplet_save": (None, [ptri, vp]),
        "spasm_hip_pivots_extract_structural": (ci, [pcsr, pint, plu, pint, C.POINTER(EchelonizeOpts)]),
        "spasm_hip_schur": (pcsr, [pcsr, pint, ci, plu, C.c_double, ptri, pint, pint]),
        "spasm_hip_dfact_create": (vp, [pcsr, pint, vp]),
        "spasm_hip_dfact_destroy": (None, [vp]),
        "spasm_hip_dfact_forget": (None, [vp]),
        "spasm_hip_dfact_hint_density": (None, [vp, C.c_double]),
        "spasm_hip_dfact_hint_eliminations": (None, [vp, C.c_double]),
        "spasm_hip_dfact_rank": (ci, [vp]),
        "spasm_hip_dfact_levels": (ci, [vp]),
        "spasm_hip_dfact_nnz": (i64, [vp]),
        "spasm_hip_dfact_sparse_image_census": (ci, [vp, C.POINTER(i64), vp]),
        "spasm_hip_dwork_create": (vp, [ci, ci, i64]),
        "spasm_hip_dwork_destroy": (None, [vp]),
        "spasm_hip_dschur": (ci, [C.POINTER(CDcsr), vp, ci, vp, vp, vp, C.POINTER(CSchurStats)]),
        "spasm_hip_dschur_fetch": (None, [vp, vp, vp, vp, vp]),
        "spasm_hip_dschur_row_pointers": (None, [vp, vp, vp]),
        "spasm_hip_echelonize_init_opts": (None, [C.POINTER(EchelonizeOpts)]),
        "spasm_hip_echelonize": (plu, [pcsr, C.POINTER(EchelonizeOpts)]),
        "spasm_hip_echelonize_profile": (None, [C.POINTER(C.c_double)]),
        "spasm_hip_echelonize_counters": (ci, [C.POINTER(C.c_longlong), ci]),
        "spasm_hip_rref": (pcsr, [plu, pint]),
        "spasm_hip_kernel": (pcsr, [plu]),
        "spasm_hip_lu_free": (None, [plu]),
        "spasm_hip_schur_estimate_density": (C.c_double, [pcsr, pint, ci, pcsr, pint, ci]),
        "spasm_hip_schur_dense": (None, [pcsr, pint, ci, pint, plu, vp, ci, pint, pint]),
        "spasm_hip_ffpack_rref": (ci, [i64, ci, ci, vp, ci, ci, C.POINTER(C.c_size_t)]),
        "spasm_hip_ffpack_LU": (ci, [i64, ci, ci, vp, ci, ci, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
        "spasm_hip_debug_combine": (None, [pcsr, pint, ci, ci, ci, C.c_uint64, vp]),
        "spasm_hip_dschur_dense": (ci, [C.POINTER(CDcsr), vp, ci, vp, vp, vp, i64, vp]),
        "spasm_hip_drref": (ci, [i64, ci, ci, vp, i64, vp, vp]),
        "spasm_hip_dechelon_extend": (ci, [i64, ci, vp, i64, ci, ci, vp, vp]),
        "spasm_hip_drref_timed": (ci, [i64, ci, ci, vp, i64, vp, vp, ci, C.POINTER(C.c_float)]),
        "spasm_hip_comm_id_bytes": (ci, []),
        "spasm_hip_comm_new_id": (None, [vp]),
        "spasm_hip_comm_create": (vp, [vp, ci, ci]),
        "spasm_hip_comm_destroy": (None, [vp]),
        "spasm_hip_comm_rank": (ci, [vp]),
        "spasm_hip_comm_world": (ci, [vp]),
        "spasm_hip_set_comm": (None, [vp]),
        "spasm_hip_shard": (None, [ci, ci, ci, pint, pint]),
        "spasm_hip_dschur_allgatherv": (ci, [vp, vp, vp, vp, vp, i64, pint, C.POINTER(i64), vp]),
        "spasm_hip_dstitch_slabs": (ci, [vp, vp, vp, ci, ci, vp, vp, vp, i64, vp]),
        "spasm_hip_echelonize_dist": (plu, [pcsr, C.POINTER(EchelonizeOpts), vp]),
        "spasm_hip_schur_resident": (i64, [pcsr, pint, ci, plu, C.c_double]),
        "spasm_hip_forget_cached_images": (None, []),
        "spasm_hip_allgatherv_plan": (ci, [ci, ci, C.POINTER(i64), vp, ci, C.POINTER(i64), C.POINTER(i64)]),
        "spasm_hip_column_slab": (ci, [pcsr, plu, ci, ci, C.POINTER(pcsr), C.POINTER(plu), pint]),
    }
    for name, (res, args) in sig.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            if os.environ.get("SPASM_HIP_LIB"):
                continue                      # an older build loaded for an A/B run: newer entry points are simply absent
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def device_count():
    return lib().spasm_hip_device_count()


def usable_cpus():
    """hardware threads cut down to the CPU quota of the control group (what the threaded host stages use)"""
    return lib().spasm_hip_usable_cpus()


def release_cached_memory():
    """spasm_hip_release_cached_memory: the device memory the library parks between calls goes back to the device"""
    lib().spasm_hip_release_cached_memory()


def require_gpu(what):
    if device_count() < 1:
        raise RuntimeError("spasm_amd.%s needs an MI355X: no HIP device is visible and there is no CPU path" % what)
