"""ctypes binding of the engine's C ABI (include/rustsasa_amd.h).

The shared library is built in-tree by rustsasa_amd/csrc/Makefile
(`__graft_entry__.build()`); there is no fallback of any kind: if the library
is missing or no GPU is usable, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "librustsasa_amd.so")

RSASA_OK = 0
RSASA_ERR_INVALID_ARGUMENT = -1
RSASA_ERR_NO_DEVICE = -2
RSASA_ERR_HIP = -3
RSASA_ERR_OUT_OF_MEMORY = -4
RSASA_ERR_GRID_TOO_LARGE = -5
RSASA_ERR_INTERNAL = -6
RSASA_ERR_QUEUE_FULL = -7

# numpy image of rsasa_atom_t (mirrors `Atom`, reference src/structures/atomic.rs:13-24)
ATOM_DTYPE = np.dtype([("position", np.float32, (3,)), ("radius", np.float32), ("id", np.uint64)],
                      align=True)
assert ATOM_DTYPE.itemsize == 24


class RsasaError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"rustsasa_amd: {message} (status {status})")
        self.status = status


class DeviceBatch(C.Structure):
    """rsasa_device_batch_t"""
    _fields_ = [
        ("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p), ("radius", C.c_void_p),
        ("id", C.c_void_p),
        ("structure_offsets_host", C.c_void_p),
        ("n_structures", C.c_size_t),
        ("n_atoms", C.c_size_t),
        ("residue_offsets", C.c_void_p),
        ("n_residues", C.c_size_t),
        ("out_atom_sasa", C.c_void_p),
        ("out_residue_sasa", C.c_void_p),
        ("out_neighbor_counts", C.c_void_p),
    ]


class Timings(C.Structure):
    """rsasa_timings_t"""
    _fields_ = [("grid_build_ms", C.c_float), ("occlusion_ms", C.c_float),
                ("aggregate_ms", C.c_float), ("total_ms", C.c_float),
                ("n_cells", C.c_uint64), ("n_atoms", C.c_uint64), ("n_deferred", C.c_uint64)]


# every symbol include/rustsasa_amd.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
SYMBOLS = {
    "rsasa_abi_version": (C.c_int, []),
    "rsasa_status_string": (C.c_char_p, [C.c_int]),
    "rsasa_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rsasa_context_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "rsasa_context_destroy": (C.c_int, [_vp]),
    "rsasa_context_last_error": (C.c_char_p, [_vp]),
    "rsasa_context_get_device": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "rsasa_context_set_simd_width": (C.c_int, [_vp, C.c_int]),
    "rsasa_context_get_simd_width": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "rsasa_context_bind_thread": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "rsasa_calculate_sasa_internal": (C.c_int, [_vp, _vp, C.c_size_t, C.c_float, C.c_size_t,
                                                C.c_ssize_t, _vp]),
    "rsasa_calculate_sasa_soa": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, C.c_float,
                                           C.c_size_t, _vp]),
    "rsasa_calculate_sasa_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t,
                                             C.c_float, C.c_size_t, _vp, _vp, C.c_size_t, _vp]),
    "rsasa_calculate_sasa_trajectory": (C.c_int, [_vp, _vp, C.c_size_t, C.c_size_t, _vp, _vp, C.c_float,
                                                  C.c_size_t, _vp, _vp, C.c_size_t, _vp]),
    "rsasa_segment_sums": (C.c_int, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp]),
    "rsasa_batch_enqueue": (C.c_int, [_vp, C.POINTER(DeviceBatch), C.c_float, C.c_size_t, _vp]),
    "rsasa_batch_wait": (C.c_int, [_vp]),
    "rsasa_batch_wait_all": (C.c_int, [_vp]),
    "rsasa_host_batch_enqueue": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t,
                                           C.c_float, C.c_size_t, _vp, _vp, C.c_size_t, _vp]),
    "rsasa_host_batch_wait": (C.c_int, [_vp]),
    "rsasa_host_batch_wait_all": (C.c_int, [_vp]),
    "rsasa_context_clone_settings": (C.c_int, [_vp, _vp]),
    "rsasa_context_enable_timing": (C.c_int, [_vp, C.c_int]),
    "rsasa_context_get_timings": (C.c_int, [_vp, C.POINTER(Timings)]),
    "rsasa_context_ids_dropped": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "rsasa_context_ids_kept": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "rsasa_context_set_call_combining": (C.c_int, [_vp, C.c_int]),
    "rsasa_call_combining_stats": (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rsasa_sphere_points": (C.c_int, [C.c_size_t, _vp, _vp, _vp]),
}

_lib = None


def load():
    """Loads librustsasa_amd.so; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch wheels bundle their own HIP runtime under the same SONAME
        # (libamdhip64.so.7) as /opt/rocm's.  A process must hold exactly one of
        # them, so when torch is installed it is imported first and this library
        # binds to the runtime torch loaded; without torch it uses /opt/rocm's.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def status_string(status: int) -> str:
    return load().rsasa_status_string(status).decode()


def check(status: int, ctx=None):
    if status == RSASA_OK:
        return
    msg = status_string(status)
    if ctx:
        detail = load().rsasa_context_last_error(ctx).decode()
        if detail:
            msg = f"{msg}: {detail}"
    raise RsasaError(status, msg)


def ptr(a):
    """Host pointer of a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data
