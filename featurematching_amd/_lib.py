"""ctypes binding of libfmatch_hip.so (the C ABI declared in include/fmatch.h).

The library is mandatory: there is no CPU or eager fallback.  ``load()`` raises if the
shared object is missing (build it with ``python -c 'import __graft_entry__ as g;
g.build()'`` or ``make -C featurematching_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfmatch_hip.so")

FM_OK = 0
FM_F32, FM_F16, FM_BF16 = 0, 1, 2     # enum fm_dtype
FM_E_CAPACITY = -5
FM_E_CANDIDATES = -6
FM_E_RANGE = -7
FM_E_DENSE = -8
FM_E_INTERNAL = -9
FM_E_STEP = -10
FM_DEV_RANGE = 4                                    # device status bit
FM_DEV_ALL_DENSE = 256                              # informational device status bit (fm_read_count_info)
FM_MODE_EXACT_SCREENING, FM_MODE_DENSE, FM_MODE_NO_CELL_MAPS, FM_MODE_EXACT_STEP, FM_MODE_STATS = 1, 2, 4, 8, 16   # `mode` bits
FM_MODE_FLAT = 32
FM_MODE_ALONE = 64      # hint about the device: this call has the GPU to itself (grids sized for the kernel alone)
FM_LAYOUT_NCHW_PREPARED = 2                         # fm_fine_match_maps*: image 1's channels-last copy is already in `scratch`

_lib = None

_p = C.c_void_p
_i = C.c_int
_f = C.c_float

# name -> (restype, argtypes); mirrors include/fmatch.h one to one
SIGNATURES = {
    "fm_version": (_i, []),
    "fm_strerror": (C.c_char_p, [_i]),
    "fm_default_cand_slots": (_i, [_f]),
    "fm_coarse_workspace_bytes": (_i, [_i, _i, _i, _i, _i, C.POINTER(C.c_size_t)]),
    "fm_coarse_workspace_bytes_mode": (_i, [_i, _i, _i, _i, _i, _i, _i, C.POINTER(C.c_size_t)]),
    "fm_coarse_match": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _i, _f, _p, _p,
                             _p, C.c_size_t, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p]),
    "fm_coarse_match_dtype": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _i, _f, _p, _p,
                                   _p, C.c_size_t, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p]),
    "fm_coarse_match_maps": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _i, _f, _p, _p,
                                  _p, C.c_size_t, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "fm_coarse_workspace_bytes_auto": (_i, [_i, _i, _i, _i, _i, C.POINTER(C.c_size_t)]),
    "fm_coarse_match_auto": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _i, _f, _p, _p,
                                  _p, C.c_size_t, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p,
                                  C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), _p]),
    "fm_debug_coarse_layout": (_i, [_i, _i, _i, _i, _i, C.POINTER(C.c_int64), _i]),
    "fm_debug_launch_corr": (_i, [_p, _i, _i, _i, _i, _i, _f, _f, _i, _p]),
    "fm_debug_launch_screen": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _p]),
    "fm_debug_launch_prep": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "fm_debug_launch_prep_f16": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "fm_debug_reset_counters": (_i, [_p, _i, _i, _i, _i, _i, _p]),
    "fm_read_count": (_i, [_p, _i, C.POINTER(C.c_int32), _p]),
    "fm_read_count_info": (_i, [_p, _i, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _p]),
    "fm_debug_launch_flat": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _i, _p]),
    "fm_coarse_softmax_stats": (_i, [_p, _i, _i, _i, _i, _i, C.POINTER(_p), C.POINTER(_p), C.POINTER(_i), C.POINTER(_p),
                                     C.POINTER(_p), C.POINTER(_i)]),
    "fm_dual_softmax_conf_at": (_i, [_p, _p, _i, _i, _i, _i, _f, _p, _p, _i, _p, _p, _i, _p, _p, _p, _i, _p, _p]),
    "fm_dual_softmax_backward_workspace_bytes": (C.c_size_t, [_i, _i, _i, _i]),
    "fm_dual_softmax_backward": (_i, [_p, _p, _i, _i, _i, _i, _f, _p, _p, _i, _p, _p, _i, _p, _p, _p, _p, _i, _p, C.c_size_t, _p,
                                      _p, _p]),
    "fm_dual_softmax_backward_dense": (_i, [_p, _p, _i, _i, _i, _i, _f, _p, _p, _i, _p, _p, _i, _p, _p, C.c_size_t, _p, _p, _p]),
    "fm_gather_windows": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p]),
    "fm_gather_windows_dtype": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _p]),
    "fm_coarse_cell_maps": (_i, [_p, _i, _i, _i, _i, _i, C.POINTER(_p), C.POINTER(_i), C.POINTER(_p),
                                 C.POINTER(_p), C.POINTER(_i), C.POINTER(_p)]),
    "fm_gather_windows_cells": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _i, _p, _p]),
    "fm_merge_pack_weights": (_i, [_p, _i, _p, _p]),
    "fm_gather_merge_windows": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p]),
    "fm_gather_windows_pair": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _i, _p,
                                    _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p]),
    "fm_coarse_tf_packed_bytes": (C.c_size_t, [_i]),
    "fm_coarse_tf_workspace_bytes": (_i, [_i, _i, _i, C.POINTER(C.c_size_t)]),
    "fm_coarse_tf_pack_weights": (_i, [_p, _i, _p, _p]),
    "fm_coarse_transformer": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _i, _p, _p, C.c_size_t, _p, _p, _p]),
    "fm_coarse_transformer_masked": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p, _p, C.c_size_t, _p, _p, _p]),
    "fm_fine_tf_packed_bytes": (C.c_size_t, []),
    "fm_fine_tf_pack_weights": (_i, [_p, _p, _p, _p]),
    "fm_fine_transformer": (_i, [_p, _p, _i, _p, _i, _i, _p, _p, _p, _p]),
    "fm_fine_transformer_status": (_i, [_p, _p, _i, _p, _i, _i, _p, _p, _p, _p, _p]),
    "fm_fine_transformer_start": (_i, [_p, _p, _i, _p, _i, _i, _p, _p, _p, _p, _i, _p, _p]),
    "fm_epipolar_errors": (_i, [_p, _p, _i, _p, _p, _i, _i, _p, _p, _p, _f, _p, _p, _p, _p]),
    "fm_fine_match": (_i, [_p, _p, _i, _p, _i, _i, _p, _p, _p, _p, _f, _p, _p, _p]),
    "fm_fine_maps_scratch_bytes": (C.c_size_t, [_i, _i, _i, _i, _i, _i, _i]),
    "fm_fine_maps_scratch_bytes_dtype": (C.c_size_t, [_i, _i, _i, _i, _i, _i, _i, _i]),
    "fm_fine_match_maps_dtype": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p,
                                      _p, _p, _f, _p, _p, _p, _p]),
    "fm_fine_match_maps": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p,
                                _p, _p, _f, _p, _p, _p, _p]),
}


class FMatchError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        msg = load().fm_strerror(status).decode()
        super().__init__(f"{where}: {msg} (status {status})")


# the diagnostic entry points (declared in csrc/fm_debug.h, not in the public header): bench.py and tools/ only
DEBUG_SIGNATURES = {k: v for k, v in SIGNATURES.items() if k.startswith("fm_debug_")}
SIGNATURES = {k: v for k, v in SIGNATURES.items() if not k.startswith("fm_debug_")}
ALL_SIGNATURES = {**SIGNATURES, **DEBUG_SIGNATURES}


def load(path=None):
    """Load (once) and return the ctypes handle of featurematching_amd/lib/libfmatch_hip.so.  `path` is for the
    tuning tools under tools/ only: they may load an experimental build of the same ABI BEFORE anything else has
    loaded the library (no environment variable can redirect the product's loader).  torch must already be imported
    by the caller so that the HIP runtime the library binds to is the one torch uses."""
    global _lib
    if _lib is None:
        lib_path = path or LIB_PATH
        if not os.path.exists(lib_path):
            raise RuntimeError(
                f"{lib_path} is missing: the HIP library is the product and has no fallback. "
                "Build it with `make -C featurematching_amd/csrc` (hipcc, --offload-arch=gfx950).")
        import torch  # noqa: F401  (loads torch's libamdhip64 first; same soname is then reused)
        lib = C.CDLL(lib_path)
        for name, (res, args) in ALL_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    elif path is not None and os.path.abspath(path) != os.path.abspath(getattr(_lib, '_name', '')):
        raise RuntimeError("the library is already loaded; load an experimental build before any other use")
    return _lib


def check(status: int, where: str) -> None:
    if status != FM_OK:
        raise FMatchError(status, where)
