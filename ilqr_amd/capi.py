"""ctypes binding of include/ilqr_amd.h (libilqr_amd.so).  Mirrors the C declarations 1:1."""
import ctypes as C
import os

from . import _build

ABI_VERSION = 5
MODEL_ACROBOT, MODEL_DOUBLE_INTEGRATOR, MODEL_LQ, MODEL_HOST, MODEL_USER = 0, 1, 2, 3, 4
FLAG_FIXED_WORK, FLAG_BACKWARD_THREAD_PER_TRAJ, _FLAG_RESERVED_4, FLAG_UNFUSED, FLAG_ANALYTIC_DERIVATIVES, FLAG_STAGED, FLAG_REFERENCE_FIXES, FLAG_REGULARIZE_VXX = 1, 2, 4, 8, 16, 32, 64, 128
DTYPE_F64, DTYPE_F32 = 0, 1
# enum ilqr_route (include/ilqr_amd.h): which of several equivalent kernels a handle uses; 0 = by batch size
ROUTE_TILE_PER_CU, ROUTE_TWO_TILES_PER_CU, ROUTE_WIDE_TILES = 1, 2, 3
ROUTE_WIDE_ONE_PER_CU, ROUTE_WIDE_TWO_PER_CU, ROUTE_NO_COMPACTION, ROUTE_FULL_RECORDS, ROUTE_LQ_THREAD_ROLLOUT, ROUTE_QUAD_CHAIN, ROUTE_LQ_RECOMMIT, ROUTE_BACKWARD_W2, ROUTE_LQ_DENSE_FD = 4, 8, 16, 32, 64, 256, 512, 1024, 2048  # (128: retired in ABI 5)
ROUTE_WAVE_PER_TRAJECTORY = 4096
NUM_STAGES = 5
STAGE_NAMES = ("derivatives", "backward", "rollout", "accept", "solve")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class Params(C.Structure):
    _fields_ = [("max_iter", C.c_int), ("tol_fun", C.c_double), ("tol_grad", C.c_double),
                ("lambda_init", C.c_double), ("dlambda_init", C.c_double),
                ("lambda_factor", C.c_double), ("lambda_max", C.c_double),
                ("lambda_min", C.c_double), ("z_min", C.c_double)]


class Desc(C.Structure):
    _fields_ = [("abi_version", C.c_int), ("model", C.c_int), ("nx", C.c_int), ("nu", C.c_int),
                ("T", C.c_int), ("B", C.c_int), ("dt", C.c_double), ("device", C.c_int),
                ("flags", C.c_int), ("dtype", C.c_int), ("u_min", _dp), ("u_max", _dp), ("goal", _dp),
                ("lq_A", _dp), ("lq_B", _dp), ("lq_Q", _dp), ("lq_R", _dp), ("lq_Qf", _dp),
                ("stream", C.c_void_p), ("params", C.POINTER(Params)), ("user_params", _dp), ("n_user_params", C.c_int),
                ("route", C.c_int), ("assume_cus", C.c_int)]


# every symbol include/ilqr_amd.h declares: name -> (restype, argtypes)
_H = C.c_void_p
SYMBOLS = {
    "ilqr_last_error": (C.c_char_p, []),
    "ilqr_abi_version": (C.c_int, []),
    "ilqr_has_user_model": (C.c_int, []),
    "ilqr_default_params": (None, [C.POINTER(Params)]),
    "ilqr_create": (C.c_int, [C.POINTER(Desc), C.POINTER(_H)]),
    "ilqr_destroy": (None, [_H]),
    "ilqr_set_stream": (C.c_int, [_H, C.c_void_p]),
    "ilqr_synchronize": (C.c_int, [_H]),
    "ilqr_init_traj": (C.c_int, [_H, _dp, _dp, _dp]),
    "ilqr_generate_trajectory": (C.c_int, [_H]),
    "ilqr_solve": (C.c_int, [_H, _dp, _dp]),
    "ilqr_warm_start": (C.c_int, [_H, _dp]),
    "ilqr_iterate": (C.c_int, [_H, C.c_int]),
    "ilqr_compute_derivatives": (C.c_int, [_H]),
    "ilqr_backward_pass": (C.c_int, [_H, _ip]),
    "ilqr_backward_step": (C.c_int, [_H]),
    "ilqr_rollout_candidates": (C.c_int, [_H, _dp]),
    "ilqr_line_search": (C.c_int, [_H]),
    "ilqr_accept_candidates": (C.c_int, [_H, _dp, _ip]),
    "ilqr_reset_state": (C.c_int, [_H, C.c_int]),
    "ilqr_set_trajectory": (C.c_int, [_H, _dp, _dp, _dp, _dp]),
    "ilqr_set_gains": (C.c_int, [_H, _dp, _dp]),
    "ilqr_set_derivatives": (C.c_int, [_H] + [_dp] * 7),
    "ilqr_set_lambda": (C.c_int, [_H, _dp, _dp]),
    "ilqr_get_trajectory": (C.c_int, [_H, _dp, _dp]),
    "ilqr_get_gains": (C.c_int, [_H, _dp, _dp]),
    "ilqr_get_derivatives": (C.c_int, [_H] + [_dp] * 7),
    "ilqr_get_cost": (C.c_int, [_H, _dp]),
    "ilqr_get_lambda": (C.c_int, [_H, _dp, _dp]),
    "ilqr_get_dV": (C.c_int, [_H, _dp]),
    "ilqr_get_gnorm": (C.c_int, [_H, _dp]),
    "ilqr_get_status": (C.c_int, [_H, _ip, _ip, _ip]),
    "ilqr_get_candidate": (C.c_int, [_H, C.c_int, _dp, _dp]),
    "ilqr_count_running": (C.c_int, [_H, _ip]),
    "ilqr_copy_cost_to_device": (C.c_int, [_H, C.c_void_p]),
    "ilqr_copy_trajectory_to_device": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "ilqr_copy_gains_to_device": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "ilqr_get_results_async": (C.c_int, [_H, _dp, _dp, _dp, _dp, _dp]),
    "ilqr_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "ilqr_host_unregister": (C.c_int, [C.c_void_p]),
    "ilqr_group_create": (C.c_int, [C.POINTER(_H), C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "ilqr_group_destroy": (None, [C.c_void_p]),
    "ilqr_group_gather_costs": (C.c_int, [C.c_void_p, _dp]),
    "ilqr_group_uses_rccl": (C.c_int, [C.c_void_p, _ip]),
    "ilqr_profile_enable": (C.c_int, [_H, C.c_int]),
    "ilqr_profile_reset": (C.c_int, [_H]),
    "ilqr_profile_read": (C.c_int, [_H, _dp, _ip]),
    "ilqr_profile_shader_clock": (C.c_int, [_H, _dp]),
    "ilqr_stage_kernel_name": (C.c_char_p, [_H, C.c_int]),
}

_lib = None
_libs = {}   # path -> loaded library (builds with a user device model live next to the stock one)


class ILQRError(RuntimeError):
    pass


def lib_path():
    return _build.LIB


def _bind(path):
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def load(build_if_missing=True, path=None):
    """Load libilqr_amd.so (building it in-tree first if it is missing or stale).  path: another build of the same
    sources, e.g. one with a user device model (ilqr_amd._build.build_user)."""
    global _lib
    if path is not None:
        path = os.path.abspath(path)
        if path not in _libs:
            if not os.path.exists(path):
                raise ILQRError("%s is missing: build it with ilqr_amd._build.build_user(header, out)" % path)
            _libs[path] = _bind(path)
        return _libs[path]
    if _lib is None:
        if build_if_missing and os.path.exists(_build.HIPCC):
            _build.build()
        if not os.path.exists(_build.LIB):
            raise ILQRError("libilqr_amd.so is missing (%s): build it with "
                            "`python -m ilqr_amd._build`; there is no fallback path" % _build.LIB)
        # ILQR_AMD_LIB selects an experiment build of the same sources (e.g. -DILQR_PHASE_TIMING)
        lib = C.CDLL(os.environ.get("ILQR_AMD_LIB", _build.LIB))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc, lib=None):
    if rc != 0:
        msg = (lib or load()).ilqr_last_error()
        raise ILQRError("libilqr_amd error %d: %s" % (rc, msg.decode() if msg else "?"))
