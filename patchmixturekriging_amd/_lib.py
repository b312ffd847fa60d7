"""ctypes loader of libpmk_hip.so (the C ABI of include/pmk.h).

The library is the product: there is no CPU fallback.  If it cannot be built or loaded the import
of this module's `lib()` raises, and every operator of the package fails loudly.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(CSRC, "libpmk_hip.so")
_LIB = None


class PmkError(RuntimeError):
    pass


class KernelDesc(C.Structure):
    """pmk_kernel_desc of include/pmk.h"""
    _fields_ = [("family", C.c_int32), ("flags", C.c_int32), ("p", C.c_double * 4)]


def build(force=False, jobs=6):
    """compile the HIP kernels for gfx950 in-tree (hipcc cross-compiles without a GPU)"""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
            if f.endswith((".hip", ".cpp", ".h"))] + [
        os.path.join(_HERE, "..", "include", "pmk.h"), os.path.join(_HERE, "..", "include", "pmk_test.h")]
    stale = force or not os.path.exists(SO_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in srcs if os.path.exists(s))
    if stale:
        if not os.path.exists("/opt/rocm/bin/hipcc"):
            raise PmkError("libpmk_hip.so is missing or stale and hipcc is not available to build it")
        subprocess.check_call(["make", "-C", CSRC, "-j%d" % jobs, "libpmk_hip.so"])
    return SO_PATH


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)
_kp = C.POINTER(KernelDesc)
_vp = C.c_void_p
_vpp = C.POINTER(C.c_void_p)
_dpp = C.POINTER(_dp)

# name -> (restype, argtypes): exactly the symbols declared in include/pmk.h and include/pmk_test.h
SIGNATURES = {
    "pmk_version": (C.c_int, []),
    "pmk_last_error": (C.c_char_p, []),
    "pmk_ctx_create": (C.c_int, [C.c_int, _vpp]),
    "pmk_ctx_set_stream": (C.c_int, [_vp, _vp]),
    "pmk_ctx_set_stream_null": (C.c_int, [_vp]),
    "pmk_ctx_synchronize": (C.c_int, [_vp]),
    "pmk_ctx_destroy": (None, [_vp]),
    "pmk_ctx_enable_timers": (C.c_int, [_vp, C.c_int]),
    "pmk_ctx_set_pipeline": (C.c_int, [_vp, C.c_int]),
    "pmk_ctx_shader_clock": (C.c_int, [_vp, C.c_int, _dp]),
    "pmk_ctx_timer_ms": (C.c_int, [_vp, C.c_char_p, _dp]),
    "pmk_bsp_build": (C.c_int, [C.c_int, C.c_int64, _dp, C.c_int, C.c_int, C.c_int, _vpp]),
    "pmk_bsp_build_device": (C.c_int, [_vp, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, _vpp]),
    "pmk_bsp_from_hyperplanes": (C.c_int, [C.c_int, C.c_int, _dp, _dp, C.c_int, _vpp]),
    "pmk_bsp_destroy": (None, [_vp]),
    "pmk_bsp_dim": (C.c_int, [_vp]),
    "pmk_bsp_levels": (C.c_int, [_vp]),
    "pmk_bsp_dot_mode": (C.c_int, [_vp]),
    "pmk_bsp_num_leaves": (C.c_int64, [_vp]),
    "pmk_bsp_num_points": (C.c_int64, [_vp]),
    "pmk_bsp_arrays": (C.c_int, [_vp, _dp, _dp, _ip, _ip]),
    "pmk_bsp_assign": (C.c_int, [_vp, C.c_int64, _dp, C.c_double, _ip, _ip, _ip, _ip]),
    "pmk_bsp_assign_device": (C.c_int, [_vp, _vp, C.c_int64, C.c_void_p, C.c_double, _ip, _ip, _ip, _ip]),
    "pmk_bsp_findpartition": (C.c_int64, [_vp, _dp]),
    "pmk_bsp_neighbours": (C.c_int64, [_vp, _dp, C.c_double, C.c_double, C.c_int64, _ip, _dp, _dp, _bp]),
    "pmk_kernel_matrix": (C.c_int, [_vp, _kp, C.c_int, C.c_int64, _dp, C.c_int64, _dp, _dp, C.c_int64]),
    "pmk_model_create": (C.c_int, [_vp, C.c_int, C.c_int64, _ip, _dpp, _dpp, _vpp]),
    "pmk_model_create_ex": (C.c_int, [_vp, C.c_int, C.c_int64, _ip, _dpp, _dpp, C.c_int, _vpp]),
    "pmk_model_fit": (C.c_int, [_vp, _kp, C.c_double]),
    "pmk_model_info": (C.c_int, [_vp, _i32p]),
    "pmk_model_set_targets": (C.c_int, [_vp, _dpp]),
    "pmk_model_set_diag": (C.c_int, [_vp, C.c_void_p]),
    "pmk_query_set_diag": (C.c_int, [_vp, _dp]),
    "pmk_model_get": (C.c_int, [_vp, C.c_int64, C.c_int, _dp, C.c_int64]),
    "pmk_model_num_patches": (C.c_int64, [_vp]),
    "pmk_model_destroy": (None, [_vp]),
    "pmk_fit_batched": (C.c_int, [_vp, _kp, C.c_double, C.c_int, C.c_int64, _ip, _dpp, _dpp, _vpp, _dpp, _i32p]),
    "pmk_model_load": (C.c_int, [_vp, C.c_int, C.c_int64, _ip, _dpp, _dpp, _dpp, _ip, _vpp]),
    "pmk_model_queryinner": (C.c_int, [_vp, C.c_int64, _kp, C.c_int64, _dp, _dp, _dp]),
    "pmk_model_queryinner_ex": (C.c_int, [_vp, C.c_int64, _kp, C.c_int64, _dp, C.c_double, _dp, _dp]),
    "pmk_model_set_weights": (C.c_int, [_vp, _dpp]),
    "pmk_model_get_weights": (C.c_int, [_vp, _dpp]),
    "pmk_model_set_bsp": (C.c_int, [_vp, _vp, C.c_int64]),
    "pmk_query_create": (C.c_int, [_vp, C.c_int64, _dp, _vpp]),
    "pmk_query_create_items": (C.c_int, [_vp, C.c_int64, C.c_void_p, C.c_void_p, _vpp]),
    "pmk_query_export_requests": (C.c_int, [_vp, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "pmk_query_export_results": (C.c_int, [_vp, C.c_void_p, C.c_void_p]),
    "pmk_query_plan": (C.c_int, [_vp, C.c_double, C.c_double]),
    "pmk_query_counts": (C.c_int, [_vp, _ip, _ip, _ip]),
    "pmk_query_region_offsets": (C.c_int, [_vp, _ip]),
    "pmk_query_items": (C.c_int, [_vp, _kp]),
    "pmk_query_item_buffers": (C.c_int, [_vp, _vpp, _vpp]),
    "pmk_query_mix": (C.c_int, [_vp, _kp, C.c_int64, C.c_int64]),
    "pmk_comm_unique_id": (C.c_int, [C.c_void_p]),
    "pmk_comm_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_void_p, _vpp]),
    "pmk_comm_rank": (C.c_int, [_vp]),
    "pmk_comm_size": (C.c_int, [_vp]),
    "pmk_comm_destroy": (None, [_vp]),
    "pmk_shard_segments": (C.c_int, [_ip, C.c_int64, C.c_int, _ip, _ip]),
    "pmk_query_predict_sharded": (C.c_int, [_vp, _vp, _kp, _kp, C.c_double, C.c_double, _ip]),
    "pmk_query_predict_allgather": (C.c_int, [_vp, _vp, _kp, _kp, C.c_double, C.c_double, _ip]),
    "pmk_comm_last_bytes": (C.c_int, [_vp, _ip, _ip]),
    "pmk_query_fetch": (C.c_int, [_vp, _dp, _dp]),
    "pmk_query_debug": (C.c_int, [_vp, _ip, _ip, _ip, _dp, _dp, _dp, _dp]),
    "pmk_query_destroy": (None, [_vp]),
    "pmk_predict_mixture": (C.c_int, [_vp, _kp, _kp, C.c_int64, _dp, C.c_double, C.c_double, _dp, _dp]),
    "pmk_query_mean": (C.c_int, [_vp, _kp, C.c_int, C.c_int64, _dp, _dp, C.c_int64, _dp, _dp]),
    "pmk_query_mean_multi": (C.c_int, [_vp, _kp, C.c_int, C.c_int64, _dp, _dp, C.c_int64, _dp, _dp]),
    # include/pmk_test.h
    "pmk_selftest_gemm": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp]),
    "pmk_selftest_trisolve": (C.c_int, [_vp, _dp, _dp, _dp, _dp]),
    "pmk_selftest_mfma_peak": (C.c_int, [_vp, _dp]),
    "pmk_test_comm_force_exchange": (C.c_int, [_vp, C.c_int]),
    "pmk_test_model_set_split": (C.c_int, [_vp, C.c_int]),
}


def lib():
    """load (building if needed) libpmk_hip.so; raises PmkError when that is impossible"""
    global _LIB
    if _LIB is not None:
        return _LIB
    try:
        # share torch's HIP runtime when torch is (or will be) in the process: both must use one
        # libamdhip64 for streams and device pointers to be interchangeable
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is plumbing, not a requirement
        pass
    path = os.environ.get("PMK_LIB") or build()       # PMK_LIB: load an A/B variant build instead
    try:
        L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError as e:
        raise PmkError("cannot load %s: %s" % (path, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            f = getattr(L, name)
        except AttributeError:
            raise PmkError("libpmk_hip.so does not export %s" % name)
        f.restype = res
        f.argtypes = args
    _LIB = L
    return L


def check(rc, what="pmk call"):
    """negative status -> exception with the library's message"""
    if rc < 0:
        raise PmkError("%s failed (%d): %s" % (what, rc, lib().pmk_last_error().decode()))
    return rc
