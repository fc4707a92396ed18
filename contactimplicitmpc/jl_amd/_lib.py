"""ctypes binding of the C ABI declared in include/cimpc.h (libcimpc_hip.so).

The library is built in-tree by `make -C contactimplicitmpc/jl_amd/csrc` (see
__graft_entry__.build).  There is no fallback: if the shared object is missing or a
symbol is absent this module raises, and creating a solver without a gfx950 device
raises `CimpcError`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CIMPC_LIB") or os.path.join(_HERE, "libcimpc_hip.so")   # CIMPC_LIB: A/B builds


class CimpcError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [("nq", C.c_int), ("nu", C.c_int), ("nw", C.c_int), ("nc", C.c_int),
                ("nb", C.c_int), ("mode", C.c_int), ("H_ref", C.c_int), ("H", C.c_int),
                ("B", C.c_int)]


class IpOpts(C.Structure):
    _fields_ = [("r_tol", C.c_double), ("kappa_tol", C.c_double), ("undercut", C.c_double),
                ("gamma_reg", C.c_double), ("kappa_reg", C.c_double), ("eps_min", C.c_double),
                ("ls_scale", C.c_double), ("max_iter", C.c_int), ("max_ls", C.c_int),
                ("stall_alpha", C.c_double), ("max_time", C.c_double)]


class NewtonOpts(C.Structure):
    _fields_ = [("r_tol", C.c_double), ("beta_init", C.c_double), ("max_time", C.c_double),
                ("kappa", C.c_double), ("max_iter", C.c_int), ("kkt_backend", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("newton_iters", C.c_longlong), ("sweeps", C.c_longlong),
                ("ip_solves", C.c_longlong), ("ip_iters", C.c_longlong),
                ("ip_failures", C.c_longlong), ("rounds", C.c_longlong)]


class Profile(C.Structure):
    _fields_ = [("ip_sweep_ms", C.c_double), ("ip_sweep_launches", C.c_longlong),
                ("ip_sweep_problems", C.c_longlong),
                ("kkt_ms", C.c_double), ("kkt_launches", C.c_longlong), ("kkt_systems", C.c_longlong),
                ("resid_ms", C.c_double), ("resid_launches", C.c_longlong),
                ("other_ms", C.c_double), ("other_launches", C.c_longlong),
                ("async_ms", C.c_double), ("async_launches", C.c_longlong), ("async_problems", C.c_longlong)]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_h = C.c_void_p

# every symbol include/cimpc.h declares: (restype, argtypes)
SIGNATURES = {
    "cimpc_default_ip_opts": (None, [C.POINTER(IpOpts)]),
    "cimpc_default_newton_opts": (None, [C.POINTER(NewtonOpts)]),
    "cimpc_last_error": (C.c_char_p, [_h]),
    "cimpc_version": (C.c_int, []),
    "cimpc_create": (C.c_int, [C.POINTER(Dims), C.POINTER(IpOpts), C.POINTER(NewtonOpts), C.c_int,
                               C.POINTER(_h)]),
    "cimpc_destroy": (C.c_int, [_h]),
    "cimpc_set_stream": (C.c_int, [_h, C.c_void_p]),
    "cimpc_synchronize": (C.c_int, [_h]),
    "cimpc_set_linearization": (C.c_int, [_h, C.c_int, _dp, _dp, _dp, _dp, _dp]),
    "cimpc_set_objective": (C.c_int, [_h, _dp, _dp, _dp, _dp, _dp, _dp, _dp]),
    "cimpc_set_altitude": (C.c_int, [_h, _dp]),
    "cimpc_set_window": (C.c_int, [_h, _ip]),
    "cimpc_set_reference": (C.c_int, [_h, _dp, _dp, _dp, _dp, _dp, _dp]),
    "cimpc_implicit_dynamics": (C.c_int, [_h, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _dp]),
    "cimpc_ip_residual": (C.c_int, [_h, C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, _dp]),
    "cimpc_ip_linear_solve": (C.c_int, [_h, C.c_int, C.c_int, _dp, _dp, C.c_double, _dp]),
    "cimpc_kkt_solve": (C.c_int, [_h, _dp, C.c_double, _dp]),
    "cimpc_kkt_solve_rho": (C.c_int, [_h, _dp, C.c_double, _dp]),
    "cimpc_newton_solve": (C.c_int, [_h, _dp, _dp, C.c_int, _dp, _ip, _dp]),
    "cimpc_mpc_solve": (C.c_int, [_h, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp, _ip, _dp, _dp, _dp, _dp, _dp, _dp]),
    "cimpc_newton_solve_dev": (C.c_int, [_h, C.c_void_p, C.c_void_p, C.c_int]),
    "cimpc_get_trajectory": (C.c_int, [_h, _dp, _dp, _dp, _dp, _dp]),
    "cimpc_get_newton_info": (C.c_int, [_h, _ip, _dp, _dp]),
    "cimpc_mpc_advance": (C.c_int, [_h, _dp]),
    "cimpc_set_gait": (C.c_int, [_h, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip]),
    "cimpc_plant_step": (C.c_int, [C.c_int, C.c_int, _dp, _dp, _dp, _dp, C.c_double, C.c_double, C.POINTER(IpOpts),
                                   _dp, _dp, _dp, _ip, _ip]),
    "cimpc_get_reference": (C.c_int, [_h, _dp, _dp, _dp, _dp, _dp, _dp, _ip]),
    "cimpc_get_stats": (C.c_int, [_h, C.POINTER(Stats)]),
    "cimpc_get_newton_log": (C.c_int, [_h, _dp, C.c_int]),
    "cimpc_get_kkt_fallbacks": (C.c_int, [_h, C.POINTER(C.c_longlong)]),
    "cimpc_get_kkt_twisted": (C.c_int, [_h, C.POINTER(C.c_longlong)]),
    "cimpc_get_kkt_twisted_fallbacks": (C.c_int, [_h, C.POINTER(C.c_longlong)]),
    "cimpc_debug_set_tw_spins": (C.c_int, [_h, C.c_int]),
    "cimpc_linear_solve_csc": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), _dp, _dp, _dp]),
    "cimpc_get_rollout_counters": (C.c_int, [_h, _ip, _ip, _ip]),
    "cimpc_profile_enable": (C.c_int, [_h, C.c_int]),
    "cimpc_profile_reset": (C.c_int, [_h]),
    "cimpc_profile_read": (C.c_int, [_h, C.POINTER(Profile)]),
    "cimpc_query_sizes": (C.c_int, [_h, _ip, _ip]),
}

_lib = None


def load():
    """Load libcimpc_hip.so and bind every declared symbol (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CimpcError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C contactimplicitmpc/jl_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
