"""ctypes binding of libpsolve_hip.so (include/psolve_hip.h).  There is no CPU fallback: if the HIP
library is missing this module raises, and every call on a box without a GPU fails in
psolve_hip_create with PSOLVE_HIP_EDEVICE."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libpsolve_hip.so")

OK, EINVAL, EDEVICE, ENUMERIC, ECOMM, ERANGE = 0, -1, -2, -3, -4, -5
UNIQUE_ID_BYTES = 128

STATUS_STRINGS = {  # MASSolver.hpp:18-33 strings
    0: "Running",
    1: "Reach relative tolerance",
    2: "Reach absolute tolerance",
    3: "Reach max iterations",
    4: "Non-finite residual",
}


class Info(C.Structure):
    _fields_ = [
        ("solver_iter", C.c_int64),
        ("num_iterations", C.c_int64),
        ("solver_error", C.c_double),
        ("final_res_norm", C.c_double),
        ("true_residual", C.c_double),
        ("rhs_norm", C.c_double),
        ("solver_status", C.c_int32),
        ("amg_levels", C.c_int32),
        ("time_analyze", C.c_double),
        ("time_factorize", C.c_double),
        ("time_solve", C.c_double),
        ("time_solve_device", C.c_double),
        ("spmv_ms_avg", C.c_double),
        ("spmv_samples", C.c_int64),
    ]


# every symbol include/psolve_hip.h declares: (restype, argtypes)
_vp, _i64, _i32, _dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
SIGNATURES = {
    "psolve_hip_abi_version": (_i32, []),
    "psolve_hip_device_count": (_i32, [C.POINTER(C.c_int)]),
    "psolve_hip_create": (_i32, [C.POINTER(_vp), _i32]),
    "psolve_hip_create_multi": (_i32, [C.POINTER(_vp), C.POINTER(C.c_int), _i32]),
    "psolve_hip_shard_rows": (_i32, [_vp, _i32, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(C.c_int)]),
    "psolve_hip_destroy": (None, [_vp]),
    "psolve_hip_last_error": (C.c_char_p, [_vp]),
    "psolve_hip_set_stream": (_i32, [_vp, _vp]),
    "psolve_hip_synchronize": (_i32, [_vp]),
    "psolve_hip_set_param": (_i32, [_vp, C.c_char_p, _dbl]),
    "psolve_hip_get_param": (_i32, [_vp, C.c_char_p, C.POINTER(_dbl)]),
    "psolve_hip_default_param": (_i32, [C.c_char_p, C.POINTER(_dbl)]),
    "psolve_hip_matrix_copy": (_i32, [_vp, _vp, _vp, _vp]),
    "psolve_hip_host_pattern_hash": (_i32, [_i64, _i64, _vp, _vp, _i32, _vp]),
    "psolve_hip_amd_order": (_i32, [_i64, _vp, _vp, _vp]),
    "psolve_hip_analyze_pattern": (_i32, [_vp, _i64, _i64, _vp, _vp, _i32]),
    "psolve_hip_factorize": (_i32, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "psolve_hip_solve": (_i32, [_vp, _vp, _vp]),
    "psolve_hip_get_info": (_i32, [_vp, C.POINTER(Info)]),
    "psolve_hip_factorize_device": (_i32, [_vp, _i64, _i64, _vp, _vp, _vp]),
    "psolve_hip_solve_device": (_i32, [_vp, _vp, _vp]),
    "psolve_hip_generate_poisson7": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32]),
    "psolve_hip_generate_elasticity_q1": (_i32, [_vp, _i32, _dbl, _dbl]),
    "psolve_hip_generate_elasticity_q1_permuted": (_i32, [_vp, _i32, _dbl, _dbl, _i32, _i64, C.c_uint64]),
    "psolve_hip_generate_poisson7_permuted": (_i32, [_vp, _i32, _i32, _i32, _i32, _i64, C.c_uint64]),
    "psolve_hip_permutation": (_i32, [_i64, _i32, _i64, C.c_uint64, _vp]),
    "psolve_hip_generate_rhs": (_i32, [_vp, C.c_uint64, _vp, _vp]),
    "psolve_hip_spmv_device": (_i32, [_vp, _vp, _vp]),
    "psolve_hip_spmv_dot_device": (_i32, [_vp, _vp, _vp, C.POINTER(_dbl)]),
    "psolve_hip_dot_device": (_i32, [_vp, _i64, _vp, _vp, C.POINTER(_dbl)]),
    "psolve_hip_axpby_device": (_i32, [_vp, _i64, _dbl, _vp, _dbl, _vp]),
    "psolve_hip_precond_apply_device": (_i32, [_vp, _vp, _vp]),
    "psolve_hip_time_spmv": (_i32, [_vp, _vp, _vp, _i32, C.POINTER(_dbl)]),
    "psolve_hip_time_vecops": (_i32, [_vp, _i32, C.POINTER(_dbl), C.POINTER(_dbl)]),
    "psolve_hip_box_probe": (_i32, [_vp, C.POINTER(_dbl), _i32]),
    "psolve_hip_amg_time_level_ops": (_i32, [_vp, _i32, _i32, C.POINTER(_dbl)]),
    "psolve_hip_malloc": (_i32, [_vp, C.POINTER(_vp), C.c_size_t]),
    "psolve_hip_free": (_i32, [_vp, _vp]),
    "psolve_hip_memcpy_h2d": (_i32, [_vp, _vp, _vp, C.c_size_t]),
    "psolve_hip_memcpy_d2h": (_i32, [_vp, _vp, _vp, C.c_size_t]),
    "psolve_hip_matrix_shape": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "psolve_hip_amg_level_info": (_i32, [_vp, _i32, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_dbl)]),
    "psolve_hip_amg_level_matrix_shape": (_i32, [_vp, _i32, _i32, C.POINTER(_i64)]),
    "psolve_hip_amg_level_matrix_copy": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "psolve_hip_ic_host_factorize": (_i32, [_i64, _vp, _vp, _vp, _dbl, _vp, _vp, _vp, _vp, C.POINTER(_dbl), C.POINTER(C.c_int)]),
    "psolve_hip_amg_level_perm": (_i32, [_vp, _i32, _vp, C.POINTER(C.c_int)]),
    "psolve_hip_reorder_perm": (_i32, [_vp, _vp, C.POINTER(C.c_int)]),
    "psolve_hip_amg_host_build": (_i32, [C.POINTER(_vp), _i64, _i64, _vp, _vp, _vp, _i32, _i32, _dbl, _dbl, _i32, _i32,
                                         C.POINTER(C.c_int)]),
    "psolve_hip_last_spmv_kernel": (_i32, [_vp, C.c_char_p, _i32]),
    "psolve_hip_last_pcg_kernel": (_i32, [_vp, _i32, C.c_char_p, _i32]),
    "psolve_hip_trim": (_i32, [_vp]),
    "psolve_hip_amg_host_build2": (_i32, [C.POINTER(_vp), _i64, _i64, _vp, _vp, _vp, _i32, _i32, _dbl, _dbl, _i32, _i32,
                                          _i32, _i32, _dbl, C.POINTER(C.c_int)]),
    "psolve_hip_amg_host_level_shape": (_i32, [_vp, _i32, _i32, _vp, C.POINTER(_dbl)]),
    "psolve_hip_amg_host_level_copy": (_i32, [_vp, _i32, _i32, _vp, _vp, _vp]),
    "psolve_hip_amg_host_free": (None, [_vp]),
    "psolve_hip_comm_unique_id": (_i32, [C.c_char_p, C.c_char_p]),
    "psolve_hip_comm_init": (_i32, [_vp, _i32, _i32, C.c_char_p, C.c_char_p]),
    "psolve_hip_local_group_create": (_i32, [C.POINTER(_vp), _i32]),
    "psolve_hip_local_group_destroy": (None, [_vp]),
    "psolve_hip_comm_init_local": (_i32, [_vp, _vp, _i32]),
    "psolve_hip_set_partition": (_i32, [_vp, _i64, _i64, _i64]),
    "psolve_hip_partition_rows": (_i32, [_i64, _vp, _i32, _i32, _vp]),
    "psolve_hip_plan_halo": (_i32, [_i32, _i32, _vp, _i64, _vp, _vp, C.POINTER(_i64), _vp]),
}

_lib = None


def load() -> C.CDLL:
    """Load the HIP library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m polysolve_amd.build` "
                "(the HIP backend has no CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
