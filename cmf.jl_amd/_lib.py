"""ctypes binding of libcmf_hip.so (include/cmf_hip.h), one Python function per C entry.

This is the Python twin of the Julia ``ccall`` layer shown in INTEGRATION.md: the
same symbols, the same argument order, Julia's column-major Float64 arrays
(``order='F'`` here).  There is no fallback: if the shared library is missing or a
call fails, a :class:`CMFError` is raised.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcmf_hip.so")

CMF_OK, CMF_ERR_ARG, CMF_ERR_HIP, CMF_ERR_STATE, CMF_ERR_UNSUPPORTED, CMF_ERR_COMM = 0, 1, 2, 3, 4, 5
CMF_COMM_AUTO, CMF_COMM_RCCL, CMF_COMM_LOOPBACK, CMF_COMM_LOOPBACK_STREAMS, CMF_COMM_PEER = 0, 1, 2, 3, 4
ABI_VERSION = 6  # CMF_ABI_VERSION of include/cmf_hip.h

# host-collective callbacks of cmf_comm_init_callbacks (include/cmf_hip.h)
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_int64)
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.c_int64)

# every symbol include/cmf_hip.h declares (tests check the library exports each one)
SYMBOLS = [
    "cmf_abi_version", "cmf_version", "cmf_source_digest", "cmf_last_error", "cmf_device_count",
    "cmf_create", "cmf_create_shard", "cmf_shard_set_left_data", "cmf_create_multi", "cmf_destroy", "cmf_synchronize", "cmf_set_stream",
    "cmf_rccl_version", "cmf_get_counter",
    "cmf_comm_unique_id", "cmf_comm_init_rccl", "cmf_comm_init_overlap", "cmf_comm_init_callbacks", "cmf_comm_info", "cmf_shard_bounds",
    "cmf_set_option", "cmf_option_names", "cmf_get_data_sumsq",
    "cmf_set_factors", "cmf_get_factors", "cmf_arm_writeback", "cmf_fingerprint",
    "cmf_update_motifs", "cmf_update_feature_maps", "cmf_compute_loss", "cmf_iterate", "cmf_fit", "cmf_converged",
    "cmf_hals_update_motifs", "cmf_hals_update_feature_maps",
    "cmf_pgd_reset", "cmf_set_mask", "cmf_pgd_set_loss", "cmf_pgd_update_motifs", "cmf_pgd_update_feature_maps", "cmf_pgd_get_steps",
    "cmf_tensor_conv", "cmf_tensor_transconv", "cmf_init_rand", "cmf_gen_synthetic",
    "cmf_time_kernel", "cmf_kernel_times",
]


class CMFError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libcmf_hip error {code}: {msg}")
        self.code = code


_lib = None


def load():
    """Load libcmf_hip.so; raises CMFError if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # Build provenance: the library carries a digest of the sources it was compiled from (cmf_source_digest).  With the
    # tree at hand, a binary that does not match it is rebuilt -- or refused when that is impossible; never loaded silently.
    from . import build as _build

    want = _build.source_digest() if all(os.path.exists(d) for d in _build.DEPS) else None
    if want is not None and _build.embedded_digest(LIB_PATH) != want:
        try:
            _build.build_lib(force=True)
        except Exception as e:  # noqa: BLE001
            raise CMFError(CMF_ERR_HIP, f"{LIB_PATH} is missing or was not built from this tree (source digest {want}) and "
                                        f"cannot be rebuilt here: {e!r} (the HIP path is the only implementation)") from e
    if not os.path.exists(LIB_PATH):
        raise CMFError(CMF_ERR_HIP, f"{LIB_PATH} not found: build it with `python cmf.jl_amd/build.py` "
                                    "(the HIP path is the only implementation)")
    lib = ctypes.CDLL(LIB_PATH)
    lib.cmf_source_digest.restype = ctypes.c_char_p
    have = lib.cmf_source_digest().decode()
    if want is not None and have != want:
        raise CMFError(CMF_ERR_HIP, f"{LIB_PATH} carries source digest {have}, the tree has {want}")
    i64, u64, dbl, cint = ctypes.c_int64, ctypes.c_uint64, ctypes.c_double, ctypes.c_int
    pd, vp = ctypes.POINTER(ctypes.c_double), ctypes.c_void_p
    pvp, pi64 = ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)

    def sig(name, argtypes, restype=cint):
        f = getattr(lib, name)
        f.argtypes = argtypes
        f.restype = restype

    sig("cmf_abi_version", [])
    sig("cmf_version", [], ctypes.c_char_p)
    sig("cmf_source_digest", [], ctypes.c_char_p)
    sig("cmf_synchronize", [vp])
    sig("cmf_rccl_version", [ctypes.POINTER(cint), ctypes.c_char_p, i64])
    sig("cmf_get_counter", [vp, ctypes.c_char_p, pi64])
    sig("cmf_last_error", [], ctypes.c_char_p)
    sig("cmf_device_count", [])
    sig("cmf_create", [pvp, cint, i64, i64, i64, i64, pd])
    sig("cmf_create_shard", [pvp, cint, i64, i64, i64, i64, pd, i64, i64])
    sig("cmf_create_multi", [pvp, cint, ctypes.POINTER(cint), cint, i64, i64, i64, i64, pd])
    sig("cmf_comm_unique_id", [vp])
    sig("cmf_comm_init_rccl", [vp, cint, cint, vp])
    sig("cmf_comm_init_overlap", [vp, vp])
    sig("cmf_comm_init_callbacks", [vp, cint, cint, ALLREDUCE_FN, ALLGATHER_FN, vp])
    sig("cmf_comm_info", [vp, ctypes.c_char_p, i64])
    sig("cmf_shard_bounds", [vp, cint, pi64, pi64])
    sig("cmf_shard_set_left_data", [vp, pd])
    sig("cmf_destroy", [vp])
    sig("cmf_set_stream", [vp, vp])
    sig("cmf_set_option", [vp, ctypes.c_char_p, cint])
    sig("cmf_option_names", [ctypes.c_char_p, i64])
    sig("cmf_get_data_sumsq", [vp, pd])
    sig("cmf_set_factors", [vp, pd, pd])
    sig("cmf_get_factors", [vp, pd, pd])
    sig("cmf_arm_writeback", [vp, pd, pd])
    sig("cmf_fingerprint", [pd, i64, i64, ctypes.POINTER(u64)])
    sig("cmf_update_motifs", [vp, dbl, dbl])
    sig("cmf_update_feature_maps", [vp, dbl, dbl, pd])
    sig("cmf_compute_loss", [vp, pd])
    sig("cmf_iterate", [vp, i64, cint, dbl, dbl, dbl, dbl, pd, pd])
    sig("cmf_fit", [vp, i64, dbl, cint, i64, dbl, cint, dbl, dbl, dbl, dbl, pd, pd, pi64, ctypes.POINTER(cint)])
    sig("cmf_converged", [pd, i64, i64, dbl])
    sig("cmf_hals_update_motifs", [vp, dbl, dbl])
    sig("cmf_hals_update_feature_maps", [vp, dbl, dbl, pd])
    sig("cmf_pgd_reset", [vp])
    sig("cmf_set_mask", [vp, pd])
    sig("cmf_pgd_set_loss", [vp, cint])
    sig("cmf_pgd_update_motifs", [vp, dbl, dbl, cint])
    sig("cmf_pgd_update_feature_maps", [vp, dbl, dbl, cint, pd])
    sig("cmf_pgd_get_steps", [vp, pd, pd])
    sig("cmf_tensor_conv", [cint, i64, i64, i64, i64, pd, pd, pd])
    sig("cmf_tensor_transconv", [cint, i64, i64, i64, i64, pd, pd, pd])
    sig("cmf_init_rand", [cint, i64, i64, i64, i64, u64, pd, pd, pd])
    sig("cmf_gen_synthetic", [cint, i64, i64, i64, i64, dbl, dbl, dbl, dbl, u64, pd, pd, pd])
    sig("cmf_kernel_times", [vp, ctypes.c_char_p, pd, pi64])
    sig("cmf_time_kernel", [vp, ctypes.c_char_p, cint, pd, pd])
    if lib.cmf_abi_version() != ABI_VERSION:
        raise CMFError(CMF_ERR_STATE, f"{LIB_PATH} implements ABI {lib.cmf_abi_version()}, this binding expects {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc):
    if rc != CMF_OK:
        raise CMFError(rc, load().cmf_last_error().decode("utf-8", "replace"))


def farr(a, shape=None):
    """Float64, Fortran-contiguous (Julia memory order) view/copy of `a`."""
    a = np.asarray(a)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return np.asfortranarray(a, dtype=np.float64)


def ptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def default_device():
    """One process per GPU: LOCAL_RANK picks the device (0 when unset)."""
    return int(os.environ.get("CMF_DEVICE", os.environ.get("LOCAL_RANK", "0")))
