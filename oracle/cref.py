"""ctypes wrapper of oracle/libcimpc_ref.so (the single-thread C restatement,
oracle/cimpc_ref.c).  Test infrastructure / CPU baseline only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libcimpc_ref.so")
_lib = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    if force or not os.path.exists(_PATH) or os.path.getmtime(_PATH) < os.path.getmtime(os.path.join(_HERE, "cimpc_ref.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])


def load():
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(_PATH)
        lib.ref_create.restype = C.c_void_p
        lib.ref_create.argtypes = [C.c_int] * 8
        lib.ref_destroy.argtypes = [C.c_void_p]
        lib.ref_set_opts.argtypes = [C.c_void_p] + [C.c_double] * 7 + [C.c_int, C.c_int] + [C.c_double] * 3 + [C.c_int]
        lib.ref_set_stall.argtypes = [C.c_void_p, C.c_double]
        lib.ref_set_noise.argtypes = [C.c_void_p, C.c_ulonglong]
        lib.ref_set_linearization.argtypes = [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp]
        lib.ref_set_objective.argtypes = [C.c_void_p, _dp, _dp]
        lib.ref_implicit_dynamics.argtypes = [C.c_void_p, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _dp]
        lib.ref_kkt_solve.argtypes = [C.c_void_p, _dp, C.c_double, _dp, _dp, C.c_int]
        lib.ref_newton_solve.argtypes = [C.c_void_p, _ip, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _ip, _dp]
        _lib = lib
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


class CRef:
    def __init__(self, dims, H_ref, H, prob=None, obj=None, ip_opts=None, newton_opts=None, kappa=2e-4):
        self.lib = load()
        d = dims
        self.d, self.H_ref, self.H = d, H_ref, H
        self.h = C.c_void_p(self.lib.ref_create(d.nq, d.nu, d.nw, d.nc, d.nb, d.mode, H_ref, H))
        io = ip_opts
        no = newton_opts
        if io is not None or no is not None:
            from .ip import IPOptions
            from .newton import NewtonOptions
            io = io or IPOptions()
            no = no or NewtonOptions(r_tol=3e-4, max_iter=5)
            self.lib.ref_set_opts(self.h, io.r_tol, io.kappa_tol, io.undercut, io.gamma_reg, io.kappa_reg, io.eps_min,
                                  io.ls_scale, io.max_iter, io.max_ls, no.r_tol, no.beta_init, kappa, no.max_iter)
            self.lib.ref_set_stall(self.h, io.stall_alpha)
        if prob is not None:
            for t in range(H_ref):
                rz = np.asfortranarray(prob["rz0"][t]); rt = np.asfortranarray(prob["rth0"][t])
                rc = self.lib.ref_set_linearization(self.h, t, _p(np.ascontiguousarray(prob["z0"][t])),
                                                    _p(np.ascontiguousarray(prob["th0"][t])),
                                                    _p(np.ascontiguousarray(prob["r0"][t])), _p(rz), _p(rt))
                assert rc == 0
        if obj is not None:
            Q = np.ascontiguousarray(np.transpose(obj.q, (0, 2, 1))); R = np.ascontiguousarray(np.transpose(obj.u, (0, 2, 1)))
            assert self.lib.ref_set_objective(self.h, _p(Q), _p(R)) == 0

    def __del__(self):
        try:
            if self.h:
                self.lib.ref_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def set_noise(self, seed):
        """Arbiter mode: theta of EVERY interior-point solve is perturbed in the last place (0 = off)."""
        self.lib.ref_set_noise(self.h, int(seed))

    def implicit_dynamics(self, window, q, theta, gamma=None, b=None):
        d, H = self.d, self.H
        nths = 2 * d.nq + d.nu
        w = np.ascontiguousarray(window, dtype=np.int32)
        q = np.ascontiguousarray(q); theta = np.ascontiguousarray(theta)
        dv = np.zeros((H, d.nd)); dz = np.zeros((H, nths, d.nd)); st = np.zeros(H, dtype=np.int32); it = np.zeros(H, dtype=np.int32)
        z = np.zeros((H, d.nz))
        g = np.ascontiguousarray(gamma) if gamma is not None else None
        bb = np.ascontiguousarray(b) if b is not None else None
        self.lib.ref_implicit_dynamics(self.h, w.ctypes.data_as(_ip), _p(q), _p(theta), _p(g), _p(bb), None, _p(dv), _p(dz),
                                       st.ctypes.data_as(_ip), it.ctypes.data_as(_ip), _p(z))
        dzt = np.transpose(dz, (0, 2, 1))
        return dict(d=dv, dq0=dzt[..., :d.nq], dq1=dzt[..., d.nq:2 * d.nq], du1=dzt[..., 2 * d.nq:], status=st, iters=it, z=z,
                    dz_raw=dz)

    def kkt_solve(self, dz_raw, beta, r, solver):
        out = np.zeros_like(r)
        rc = self.lib.ref_kkt_solve(self.h, _p(np.ascontiguousarray(dz_raw)), float(beta), _p(np.ascontiguousarray(r)), _p(out), int(solver))
        assert rc == 0
        return out

    def newton_solve(self, window, ref, q0, q1, solver=1):
        d, H = self.d, self.H
        w = np.ascontiguousarray(window, dtype=np.int32)
        q = np.zeros((H + 2, d.nq)); u = np.zeros((H, d.nu)); nu = np.zeros((H, d.nd))
        info = np.zeros(4, dtype=np.int32); rinfo = np.zeros(1)
        self.lib.ref_newton_solve(self.h, w.ctypes.data_as(_ip), _p(np.ascontiguousarray(ref.q)), _p(np.ascontiguousarray(ref.u)),
                                  _p(np.ascontiguousarray(ref.w)), _p(np.ascontiguousarray(ref.theta)),
                                  _p(np.ascontiguousarray(q0)), _p(np.ascontiguousarray(q1)), int(solver), _p(q), _p(u), _p(nu),
                                  info.ctypes.data_as(_ip), _p(rinfo))
        return dict(q=q, u=u, nu=nu, iters=int(info[0]), sweeps=int(info[1]), ip_iters=int(info[2]), ip_fail=int(info[3]),
                    r_norm=float(rinfo[0]))
