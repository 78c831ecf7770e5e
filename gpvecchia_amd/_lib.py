"""ctypes binding of libgpvecchia_hip.so (include/gpvecchia.h).

The shared library is the product; this module only loads it and declares the
prototypes.  There is deliberately no Python/NumPy compute fallback: if the
library is missing it is built (hipcc), and if that fails the import error
propagates; if no GPU is present every compute entry raises GpvError.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPV_LIB") or os.path.join(_HERE, "libgpvecchia_hip.so")   # GPV_LIB: tuning variants only
NSUMS = 8

GPV_WANT_U = 1
GPV_WANT_LOGLIK_Z = 2
GPV_WANT_NUMERATOR = 4
GPV_WANT_DENOM = 8
GPV_WANT_MEAN = 16
GPV_WANT_MEAN_B = 32

# every symbol include/gpvecchia.h declares (tests check the library exports all of them)
EXPORTS = [
    "gpv_status_string", "gpv_last_hip_error", "gpv_version", "gpv_device_count", "gpv_max_p",
    "gpv_U_NZentries", "gpv_U_NZentries_mat", "gpv_MaternFun", "gpv_EsqeFun",
    "gpv_plan_create", "gpv_plan_destroy", "gpv_plan_set_data", "gpv_plan_eval",
    "gpv_plan_get_sums", "gpv_plan_get_Lentries", "gpv_plan_get_Zentries",
    "gpv_plan_Lentries_device", "gpv_plan_rows", "gpv_plan_dims", "gpv_plan_last_kernel_ms", "gpv_plan_set_kernel_timing",
    "gpv_loglik_z_from_sums", "gpv_numerator_from_sums", "gpv_whichCondOnLatent",
    "gpv_plan_build_posterior", "gpv_plan_build_posterior_fill", "gpv_plan_posterior_levels", "gpv_loglik_from_sums", "gpv_plan_get_posterior_mean", "gpv_find_ordered_nn", "gpv_order_maxmin_exact", "gpv_ic0",
    "gpv_plan_cache_clear", "gpv_plan_cache_stats", "gpv_hash_bytes", "gpv_plan_vl_begin", "gpv_plan_vl_step", "gpv_plan_vl_get",
    "gpv_plan_set_user_order", "gpv_plan_vl_begin_user", "gpv_plan_vl_restart", "gpv_plan_vl_get_user", "gpv_plan_vl_loglik",
    "gpv_mplan_create", "gpv_mplan_destroy", "gpv_mplan_set_data", "gpv_mplan_eval", "gpv_mplan_get_Lentries",
    "gpv_mplan_create_replicas", "gpv_mplan_count", "gpv_mplan_set_data_one", "gpv_mplan_build_posterior",
    "gpv_mplan_eval_each", "gpv_mplan_vl_begin_one", "gpv_mplan_vl_step_each", "gpv_mplan_vl_get_one",
    "gpv_plan_set_observed", "gpv_rccl_version", "gpv_comm_unique_id", "gpv_comm_create", "gpv_comm_destroy", "gpv_plan_set_comm",
]


class GpvError(RuntimeError):
    def __init__(self, status: int, where: str = ""):
        self.status = status
        msg = lib().gpv_status_string(status).decode()
        if status == 6:                                   # GPV_ERR_HIP: say which HIP call failed and how
            buf = C.create_string_buffer(256)
            code = lib().gpv_last_hip_error(buf, 256)
            msg += f" -- hipError_t {code}: {buf.value.decode(errors='replace')}"
        super().__init__(f"libgpvecchia_hip: {where}: {msg} (status {status})")


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        from . import build as _build
        _build.build()
    L = C.CDLL(LIB_PATH)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    i64 = C.c_int64
    L.gpv_status_string.restype = C.c_char_p
    L.gpv_status_string.argtypes = [C.c_int]
    L.gpv_version.restype = C.c_int
    L.gpv_last_hip_error.restype = C.c_int
    L.gpv_last_hip_error.argtypes = [C.c_char_p, C.c_int]
    L.gpv_device_count.argtypes = [ip]
    L.gpv_max_p.restype = C.c_int
    L.gpv_U_NZentries.restype = None
    L.gpv_U_NZentries.argtypes = [ip, ip, ip, ip, ip, dp, ip, ip, dp, dp, C.POINTER(C.c_char_p), dp, ip, dp, dp, ip, ip]
    L.gpv_U_NZentries_mat.restype = None
    L.gpv_U_NZentries_mat.argtypes = [ip, ip, ip, ip, ip, dp, dp, dp, dp, ip, ip]
    L.gpv_MaternFun.restype = None
    L.gpv_MaternFun.argtypes = [dp, ip, dp, dp, ip]
    L.gpv_EsqeFun.restype = None
    L.gpv_EsqeFun.argtypes = [dp, ip, dp, dp, ip]
    L.gpv_plan_create.argtypes = [C.POINTER(vp), C.c_int, i64, C.c_int, C.c_int, dp, ip, ip, i64, i64]
    L.gpv_plan_destroy.argtypes = [vp]
    L.gpv_plan_set_data.argtypes = [vp, dp]
    L.gpv_plan_eval.argtypes = [vp, C.c_char_p, vp, C.c_int, vp, i64, C.c_int, vp, vp]   # (data pointers as integers: cheaper marshalling)
    L.gpv_plan_get_sums.argtypes = [vp, dp]
    L.gpv_plan_get_Lentries.argtypes = [vp, dp]
    L.gpv_plan_get_Zentries.argtypes = [vp, dp]
    L.gpv_plan_Lentries_device.argtypes = [vp, C.POINTER(vp), C.POINTER(i64)]
    L.gpv_plan_rows.argtypes = [vp, C.POINTER(i64), C.POINTER(i64)]
    L.gpv_plan_dims.argtypes = [vp, C.POINTER(i64), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.gpv_plan_last_kernel_ms.argtypes = [vp, dp]
    L.gpv_plan_set_kernel_timing.argtypes = [vp, C.c_int]
    L.gpv_loglik_z_from_sums.argtypes = [dp, i64, dp]
    L.gpv_numerator_from_sums.argtypes = [dp, dp, dp]
    L.gpv_plan_build_posterior.argtypes = [vp, ip, ip]
    L.gpv_plan_posterior_levels.argtypes = [vp, ip]
    L.gpv_plan_build_posterior_fill.argtypes = [vp, ip, ip, C.c_double, dp]
    L.gpv_plan_get_posterior_mean.argtypes = [vp, dp]
    L.gpv_loglik_from_sums.argtypes = [dp, i64, dp]
    L.gpv_mplan_create.argtypes = [C.POINTER(vp), ip, C.c_int, i64, C.c_int, C.c_int, dp, ip, ip]
    L.gpv_mplan_destroy.argtypes = [vp]
    L.gpv_mplan_set_data.argtypes = [vp, dp]
    L.gpv_mplan_eval.argtypes = [vp, C.c_char_p, dp, C.c_int, dp, i64, C.c_int, dp]
    L.gpv_mplan_get_Lentries.argtypes = [vp, dp]
    L.gpv_mplan_create_replicas.argtypes = [C.POINTER(vp), ip, C.c_int, i64, C.c_int, C.c_int, dp, ip, ip]
    L.gpv_mplan_count.argtypes = [vp, ip]
    L.gpv_mplan_set_data_one.argtypes = [vp, C.c_int, dp]
    L.gpv_mplan_build_posterior.argtypes = [vp, ip, ip]
    L.gpv_mplan_eval_each.argtypes = [vp, C.c_char_p, dp, C.c_int, dp, C.c_int, dp]
    L.gpv_mplan_vl_begin_one.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, dp]
    L.gpv_mplan_vl_step_each.argtypes = [vp, C.c_char_p, dp, C.c_int, ip, dp, ip]
    L.gpv_mplan_vl_get_one.argtypes = [vp, C.c_int, dp, dp, dp]
    L.gpv_plan_set_observed.argtypes = [vp, ip]
    L.gpv_comm_unique_id.argtypes = [vp]
    L.gpv_comm_create.argtypes = [C.POINTER(vp), C.c_int, C.c_int, C.c_int, vp]
    L.gpv_comm_destroy.argtypes = [vp]
    L.gpv_plan_set_comm.argtypes = [vp, vp]
    L.gpv_order_maxmin_exact.argtypes = [dp, i64, C.c_int, ip]
    L.gpv_ic0.argtypes = [i64, ip, ip, dp, C.POINTER(C.c_int64)]
    L.gpv_find_ordered_nn.argtypes = [C.c_int, dp, i64, C.c_int, C.c_int, i64, i64, ip]
    L.gpv_whichCondOnLatent.argtypes = [ip, i64, C.c_int, i64, ip]
    L.gpv_plan_cache_stats.argtypes = [C.POINTER(i64), C.POINTER(i64)]
    L.gpv_hash_bytes.argtypes = [vp, i64, C.c_uint64, C.POINTER(C.c_uint64)]
    L.gpv_plan_vl_begin.argtypes = [vp, C.c_int, dp, dp, dp, dp]
    L.gpv_plan_vl_step.argtypes = [vp, C.c_char_p, dp, C.c_int, dp, ip]
    L.gpv_plan_vl_get.argtypes = [vp, dp, dp, dp]
    L.gpv_plan_set_user_order.argtypes = [vp, ip]
    L.gpv_plan_vl_begin_user.argtypes = [vp, C.c_int, dp, dp, dp, dp]
    L.gpv_plan_vl_restart.argtypes = [vp, dp]
    L.gpv_plan_vl_get_user.argtypes = [vp, dp, dp, dp]
    L.gpv_plan_vl_loglik.argtypes = [vp, C.c_char_p, dp, C.c_int, dp]
    _lib = L
    return L


def dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def device_count() -> int:
    n = C.c_int(0)
    lib().gpv_device_count(C.byref(n))
    return int(n.value)


def check(status: int, where: str):
    if status != 0:
        raise GpvError(int(status), where)


NA_INTEGER = -2147483648


def as_r_int_matrix(a):
    """float/int matrix with NaN (R's NA) -> Fortran-ordered int32 with NA_INTEGER."""
    a = np.asarray(a)
    if a.dtype == np.int32 and a.flags.f_contiguous:
        return a                                                 # already what R's .C() would pass
    if a.dtype.kind == "f":
        out = np.where(np.isnan(a), NA_INTEGER, a)
    else:
        out = a
    return np.asarray(out, dtype=np.int32, order="F")            # one pass: cast and transpose together
