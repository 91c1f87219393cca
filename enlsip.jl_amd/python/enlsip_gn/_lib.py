"""ctypes binding of libenlsip_gn.so (include/enlsip_gn.h).

The product path is the HIP library: if it is missing or cannot be loaded this module raises —
there is no CPU fallback (the CPU oracle under /oracle is test infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG_ROOT = Path(__file__).resolve().parents[2]          # .../enlsip.jl_amd
LIB_PATH = Path(os.environ.get("ENLSIP_GN_LIB", _PKG_ROOT / "lib" / "libenlsip_gn.so"))

FACTOR_A, FACTOR_L11, FACTOR_J2 = 0, 1, 2
FLAG_UPDATE_MFMA, FLAG_UPDATE_REFLECTORS = 1, 2
STAGE_NAMES = ("constraint", "jq1", "panel", "update", "pivot", "total")


class Opts(C.Structure):
    _fields_ = [("device", C.c_int32), ("flags", C.c_int32), ("panel_width", C.c_int32),
                ("tile_rows", C.c_int32), ("stream", C.c_void_p)]


class Info(C.Structure):
    _fields_ = [("rankA", C.c_int64), ("rankJ2", C.c_int64), ("code", C.c_int64),
                ("dimA", C.c_int64), ("dimJ2", C.c_int64), ("status", C.c_int64)]


# enlsip_gn_allgather_fn: int (*)(void* ctx, const void* dsend, void* drecv, size_t bytes_per_rank, void* hip_stream)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_i64 = C.c_int64
_h = C.c_void_p

# name -> (restype, argtypes); every symbol include/enlsip_gn.h declares
PROTOTYPES = {
    "enlsip_gn_version": (C.c_int, []),
    "enlsip_gn_create": (C.c_int, [C.POINTER(_h), C.POINTER(Opts)]),
    "enlsip_gn_destroy": (C.c_int, [_h]),
    "enlsip_gn_last_error": (C.c_char_p, [_h]),
    "enlsip_gn_synchronize": (C.c_int, [_h]),
    "enlsip_gn_solve": (C.c_int, [_h, _i64, _i64, _i64, C.c_void_p, _i64, C.c_void_p, C.c_void_p, _i64,
                                  C.c_void_p, C.c_double, _i64, _i64, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.POINTER(Info), C.c_void_p, C.c_void_p, C.c_void_p]),
    "enlsip_gn_factor_constraints": (C.c_int, [_h, _i64, _i64, _i64, C.c_void_p, _i64, C.c_void_p, C.c_double,
                                               C.POINTER(Info)]),
    "enlsip_gn_solve_factored": (C.c_int, [_h, _i64, _i64, _i64, C.c_void_p, _i64, C.c_void_p, C.c_double, _i64,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Info), C.c_void_p, C.c_void_p,
                                           C.c_void_p]),
    "enlsip_gn_solve_batched": (C.c_int, [_h, _i64, _i64, _i64, _i64, C.c_void_p, _i64, _i64, C.c_void_p,
                                          C.c_void_p, _i64, _i64, C.c_void_p, C.c_double, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "enlsip_gn_solve_batched_dev": (C.c_int, [_h, _i64, _i64, _i64, _i64, C.c_void_p, _i64, _i64, C.c_void_p,
                                              C.c_void_p, _i64, _i64, C.c_void_p, C.c_double, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                              C.c_void_p]),
    "enlsip_gn_factor_shape": (C.c_int, [_h, C.c_int, _i64, _ip, _ip]),
    "enlsip_gn_get_R": (C.c_int, [_h, C.c_int, _i64, C.c_void_p, _i64]),
    "enlsip_gn_get_diagR": (C.c_int, [_h, C.c_int, _i64, C.c_void_p]),
    "enlsip_gn_get_jpvt": (C.c_int, [_h, C.c_int, _i64, C.c_void_p]),
    "enlsip_gn_apply_qt": (C.c_int, [_h, C.c_int, _i64, C.c_void_p]),
    "enlsip_gn_apply_q": (C.c_int, [_h, C.c_int, _i64, C.c_void_p]),
    "enlsip_gn_get_JQ1": (C.c_int, [_h, _i64, C.c_void_p, _i64]),
    "enlsip_gn_resolve": (C.c_int, [_h, _i64, _i64, _i64, _i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "enlsip_gn_gradient": (C.c_int, [_h, _i64, C.c_void_p]),
    "enlsip_gn_jacobian_times": (C.c_int, [_h, _i64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "enlsip_gn_first_lagrange": (C.c_int, [_h, _i64, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                           C.POINTER(C.c_double)]),
    "enlsip_gn_second_lagrange": (C.c_int, [_h, _i64, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "enlsip_gn_newton_direction": (C.c_int, [_h, _i64, C.c_void_p, _i64, C.c_void_p, _ip]),
    "enlsip_gn_tsqr_local_dev": (C.c_int, [_h, _i64, _i64, _i64, C.c_void_p, _i64, C.c_void_p, C.c_void_p, _i64,
                                           C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, _dp, _ip]),
    "enlsip_gn_tsqr_combine_dev": (C.c_int, [_h, _i64, _i64, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                             C.c_void_p, _dp, C.POINTER(Info), C.c_void_p]),
    "enlsip_gn_tsqr_unique_id": (C.c_int, [C.c_void_p]),
    "enlsip_gn_tsqr_init_rccl": (C.c_int, [_h, C.c_void_p, C.c_int, C.c_int]),
    "enlsip_gn_tsqr_set_comm": (C.c_int, [_h, C.c_void_p, C.c_int, C.c_int]),
    "enlsip_gn_tsqr_set_exchange": (C.c_int, [_h, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "enlsip_gn_solve_tsqr": (C.c_int, [_h, _i64, _i64, _i64, C.c_void_p, _i64, C.c_void_p, C.c_void_p, _i64, C.c_void_p,
                                       C.c_double, C.c_void_p, C.c_void_p, _dp, C.POINTER(Info), C.c_void_p]),
    "enlsip_gn_tsqr_get_stage_ms": (C.c_int, [_h, C.POINTER(C.c_float)]),
    "enlsip_gn_tsqr_get_transport": (C.c_int, [_h, C.POINTER(C.c_int)]),
    "enlsip_gn_tsqr_get_exchange": (C.c_int, [_h, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "enlsip_gn_set_profiling": (C.c_int, [_h, C.c_int]),
    "enlsip_gn_get_stage_ms": (C.c_int, [_h, C.POINTER(C.c_float)]),
    "enlsip_gn_get_update_stats": (C.c_int, [_h, C.POINTER(C.c_float), _ip, _dp]),
    "enlsip_gn_get_update_table": (C.c_int, [_h, C.c_int64, _dp, C.POINTER(C.c_float), _ip]),
    "enlsip_gn_get_launch_plan": (C.c_int, [_h, _ip, C.POINTER(C.c_int), _ip]),
    "enlsip_gn_get_update_totals": (C.c_int, [_h, C.POINTER(C.c_float), C.POINTER(C.c_float), _ip, _dp]),
    "enlsip_gn_measure_stream": (C.c_int, [_h, C.c_int64, C.c_int, _dp]),
    "enlsip_gn_debug_copy_W": (C.c_int, [_h, C.c_int64, _dp, _ip, C.c_int64]),
    "enlsip_gn_full_constraints_times": (C.c_int, [_h, _i64, _i64, C.c_void_p, _i64, C.c_void_p, C.c_void_p]),
    "enlsip_gn_matrix_times_QA": (C.c_int, [_h, _i64, _i64, C.c_void_p, _i64, C.c_void_p, _i64]),
    "enlsip_gn_get_route": (C.c_int, [_h, C.POINTER(C.c_uint64)]),
    "enlsip_gn_route_name": (C.c_char_p, [C.c_int]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library and bind every prototype.  Raises OSError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise OSError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
