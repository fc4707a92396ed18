"""Host-side mirror of the reference's solver interface for the MI355X path.

Names, argument meaning and error behaviour follow the reference (file:line relative to
/root/reference):
    ImplicitTrajectory / implicit_dynamics!   src/controller/implicit_dynamics.jl:21-90, 156-192
    Newton / newton_solve!                    src/controller/newton.jl:37-91, 169-288
    linear_solve!(solver, D, R, r)            src/controller/newton.jl:218 (LinearSolver seam)
    TrackingObjective                         src/controller/objective.jl:3-16
    NewtonOptions / InteriorPointOptions      src/controller/newton.jl:2-11, policy.jl:54-61
All arithmetic happens in libcimpc_hip.so (HIP, gfx950); this module only marshals
numpy arrays across the C ABI.  Batched: every trajectory argument carries a leading
rollout dimension B (B = 1 is the reference's call).
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from ._lib import CimpcError

MODE_CONFIGURATION = 0
MODE_CONFIGURATIONFORCE = 1


@dataclass
class InteriorPointOptions:
    r_tol: float = 1.0e-8
    kappa_tol: float = 2.0e-4
    undercut: float = 5.0
    gamma_reg: float = 0.1
    kappa_reg: float = 1.0e-3
    eps_min: float = 0.05
    ls_scale: float = 0.5
    max_iter: int = 100
    max_ls: int = 3
    stall_alpha: float = 1.0e-13
    max_time: float = 0.0          # seconds per solve (policy.jl:9,61 ip_max_time); <= 0 or >= 1000: unlimited


@dataclass
class NewtonOptions:
    r_tol: float = 3.0e-4
    max_iter: int = 5
    beta_init: float = 1.0e-5
    max_time: float = 0.0
    kappa: float = 2.0e-4
    kkt_backend: int = 0      # 3 = the condensed solve in mixed precision (fp32-MFMA Schur products + fp64 refinement + fp64 fallback); 0 auto: condensed MFMA solve (TrackingObjective), banded LDL^T (velocity objective), dense LU (:configurationforce); 1 dense LU (reference default); 2 banded LDL^T (:configuration, any objective)


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ipt(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


def _f64(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a


def linear_solve_csc(A, b, device=0):
    """`linear_solve!(solver, x, A, b)` with the sparse matrix passed in (B1 stand-alone, lu.jl:4-12): A is a
    scipy.sparse matrix (converted to CSC with Julia's 1-based Int64 index arrays), returns x."""
    import scipy.sparse as sp
    A = sp.csc_matrix(A)
    A.sum_duplicates()
    n = A.shape[0]
    if A.shape != (n, n):
        raise ValueError("square matrix expected")
    cp = np.ascontiguousarray(A.indptr.astype(np.int64) + 1)
    rv = np.ascontiguousarray(A.indices.astype(np.int64) + 1)
    nz = np.ascontiguousarray(A.data, dtype=np.float64)
    bb = _f64(b, (n,))
    x = np.zeros(n)
    pll = lambda a: a.ctypes.data_as(C.POINTER(C.c_longlong))
    rc = _lib.load().cimpc_linear_solve_csc(int(device), n, pll(cp), pll(rv), _dp(nz), _dp(bb), _dp(x))
    if rc == -5:      # CIMPC_ERR_SINGULAR: the reference's lu_solver throws SingularException here (lu.jl:4-12)
        raise np.linalg.LinAlgError("cimpc_linear_solve_csc: singular matrix (zero or non-finite pivot)")
    if rc != 0:
        raise CimpcError(f"cimpc_linear_solve_csc failed ({rc})")
    return x


class CIMPCSolver:
    """One handle = one GPU.  Wraps the C ABI; raises CimpcError on any non-zero status."""

    def __init__(self, nq, nu, nw, nc, nb, H_ref, H, B=1, mode=MODE_CONFIGURATION,
                 ip_opts: InteriorPointOptions = None, newton_opts: NewtonOptions = None, device=0):
        self.lib = _lib.load()
        self.dims = _lib.Dims(nq, nu, nw, nc, nb, mode, H_ref, H, B)
        ip_opts = ip_opts or InteriorPointOptions()
        newton_opts = newton_opts or NewtonOptions()
        self._ip = _lib.IpOpts(ip_opts.r_tol, ip_opts.kappa_tol, ip_opts.undercut, ip_opts.gamma_reg,
                               ip_opts.kappa_reg, ip_opts.eps_min, ip_opts.ls_scale, ip_opts.max_iter,
                               ip_opts.max_ls, ip_opts.stall_alpha, ip_opts.max_time)
        self._nt = _lib.NewtonOpts(newton_opts.r_tol, newton_opts.beta_init, newton_opts.max_time,
                                   newton_opts.kappa, newton_opts.max_iter, newton_opts.kkt_backend)
        self.h = C.c_void_p()
        rc = self.lib.cimpc_create(C.byref(self.dims), C.byref(self._ip), C.byref(self._nt), device,
                                   C.byref(self.h))
        if rc != 0:
            msg = self.lib.cimpc_last_error(None).decode()
            self.h = None
            raise CimpcError(f"cimpc_create failed ({rc}): {msg}")
        self.nq, self.nu, self.nw, self.nc, self.nb = nq, nu, nw, nc, nb
        self.H_ref, self.H, self.B, self.mode = H_ref, H, B, mode
        self.ny = 2 * nc + nb
        self.nz = nq + 4 * nc + 2 * nb
        self.nth = 2 * nq + nu + nw + 2
        self.nths = 2 * nq + nu
        self.nd = nq if mode == MODE_CONFIGURATION else nq + nc + nb
        self.nr = nq + nu if mode == MODE_CONFIGURATION else nq + nu + nc + nb
        self.N = H * (self.nr + self.nd)

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            raise CimpcError(f"{what} failed ({rc}): {self.lib.cimpc_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.cimpc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        self._check(self.lib.cimpc_set_stream(self.h, C.c_void_p(stream_ptr)), "set_stream")

    def synchronize(self):
        self._check(self.lib.cimpc_synchronize(self.h), "synchronize")

    # -- A1 ---------------------------------------------------------------------------------
    def set_linearization(self, t, z0, th0, r0, rz0, rth0):
        """LinearizedStep + RLin/RZLin/RthLin of knot t (1-based).  rz0 (nz,nz), rth0 (nz,nth)."""
        z0 = _f64(z0, (self.nz,)); th0 = _f64(th0, (self.nth,)); r0 = _f64(r0, (self.nz,))
        rz = np.asfortranarray(np.asarray(rz0, dtype=np.float64))
        rt = np.asfortranarray(np.asarray(rth0, dtype=np.float64))
        if rz.shape != (self.nz, self.nz) or rt.shape != (self.nz, self.nth):
            raise ValueError("rz0 / rth0 shape mismatch")
        self._check(self.lib.cimpc_set_linearization(self.h, int(t), _dp(z0), _dp(th0), _dp(r0),
                                                     _dp(rz), _dp(rt)), "set_linearization")

    def set_objective(self, Q, R, Cg=None, Cb=None, V=None, q_target=None, v_target=None):
        """TrackingObjective weights, per horizon step: Q (H,nq,nq), R (H,nu,nu) [row-major numpy
        arrays of symmetric blocks are transposed to the column-major ABI]."""
        def blk(a, n):
            if a is None:
                return None
            a = np.asarray(a, dtype=np.float64)
            if a.shape != (self.H, n, n):
                raise ValueError(f"objective block shape {a.shape} != {(self.H, n, n)}")
            return np.ascontiguousarray(np.transpose(a, (0, 2, 1)))
        Qc, Rc, Gc, Bc, Vc = blk(Q, self.nq), blk(R, self.nu), blk(Cg, self.nc), blk(Cb, self.nb), blk(V, self.nq)
        def tgt(a):      # (H, nq) per-step targets; a single (nq,) vector is broadcast over the horizon
            if a is None:
                return None
            a = np.asarray(a, dtype=np.float64)
            if a.shape == (self.nq,):
                a = np.tile(a[None], (self.H, 1))
            return _f64(a, (self.H, self.nq))
        qt = tgt(q_target); vt = tgt(v_target)
        self._check(self.lib.cimpc_set_objective(self.h, _dp(Qc), _dp(Rc), _dp(Gc), _dp(Bc), _dp(Vc),
                                                 _dp(qt), _dp(vt)), "set_objective")

    def set_altitude(self, alt):
        alt = _f64(alt, (self.B, self.nc)) if alt is not None else None
        self._check(self.lib.cimpc_set_altitude(self.h, _dp(alt)), "set_altitude")

    def set_window(self, window):
        """window: (B, H+2) 1-based reference-knot indices (policy.jl:154-171)."""
        w = np.ascontiguousarray(window, dtype=np.int32)
        if w.shape != (self.B, self.H + 2):
            raise ValueError(f"window shape {w.shape} != {(self.B, self.H + 2)}")
        self._check(self.lib.cimpc_set_window(self.h, _ipt(w)), "set_window")

    def set_reference(self, q_ref, u_ref, w_ref, gamma_ref, b_ref, theta_ref):
        B, H = self.B, self.H
        self._check(self.lib.cimpc_set_reference(
            self.h, _dp(_f64(q_ref, (B, H + 2, self.nq))), _dp(_f64(u_ref, (B, H, self.nu))),
            _dp(_f64(w_ref, (B, H, self.nw)) if w_ref is not None else None),
            _dp(_f64(gamma_ref, (B, H, self.nc)) if gamma_ref is not None else None),
            _dp(_f64(b_ref, (B, H, self.nb)) if b_ref is not None else None),
            _dp(_f64(theta_ref, (B, H, self.nth)))), "set_reference")

    # -- B3: implicit_dynamics! ---------------------------------------------------------------
    def implicit_dynamics(self, q, theta, gamma=None, b=None, want_z=False):
        """implicit_dynamics!(im_traj, traj; window).  Returns dict with d (B,H,nd),
        dq0/dq1 (B,H,nd,nq), du1 (B,H,nd,nu), status, iters (B,H) [, z (B,H,nz)]."""
        B, H, nd, nq, nu = self.B, self.H, self.nd, self.nq, self.nu
        q = _f64(q, (B, H + 2, nq)); theta = _f64(theta, (B, H, self.nth))
        gamma = _f64(gamma, (B, H, self.nc)) if gamma is not None else None
        b = _f64(b, (B, H, self.nb)) if b is not None else None
        d = np.zeros((B, H, nd)); dz = np.zeros((B, H, self.nths, nd))
        status = np.zeros((B, H), dtype=np.int32); iters = np.zeros((B, H), dtype=np.int32)
        z = np.zeros((B, H, self.nz)) if want_z else None
        self._check(self.lib.cimpc_implicit_dynamics(self.h, _dp(q), _dp(theta), _dp(gamma), _dp(b),
                                                     _dp(d), _dp(dz), _ipt(status), _ipt(iters), _dp(z)),
                    "implicit_dynamics")
        dzt = np.transpose(dz, (0, 1, 3, 2))       # (B,H,nd,nths): column-major -> row-major view
        out = dict(d=d, dq0=dzt[..., 0:nq], dq1=dzt[..., nq:2 * nq], du1=dzt[..., 2 * nq:2 * nq + nu],
                   status=status, iters=iters)
        if want_z:
            out["z"] = z
        return out

    # -- B2: the per-knot interior-point callbacks (linearized_solver.jl:364-444) on caller-supplied points --------
    def ip_residual(self, t, z, theta, kappa, alt=None):
        """rlin!(r, z, theta, kappa) for n points of knot t (1-based): z (n, nz), theta (n, nth) -> r (n, nz)."""
        z = np.ascontiguousarray(np.atleast_2d(z), dtype=np.float64)
        n = z.shape[0]
        z = _f64(z, (n, self.nz)); theta = _f64(np.atleast_2d(theta), (n, self.nth))
        alt = _f64(np.atleast_2d(alt), (n, self.nc)) if alt is not None else None
        r = np.zeros((n, self.nz))
        self._check(self.lib.cimpc_ip_residual(self.h, int(t), n, _dp(z), _dp(theta), _dp(alt), float(kappa), _dp(r)), "ip_residual")
        return r

    def ip_linear_solve(self, t, z, r, reg=0.0):
        """rzlin!(rz, z; reg) + linear_solve!(Delta, rz, r; reg) for n points of knot t (1-based) -> Delta (n, nz)."""
        z = np.ascontiguousarray(np.atleast_2d(z), dtype=np.float64)
        n = z.shape[0]
        z = _f64(z, (n, self.nz)); r = _f64(np.atleast_2d(r), (n, self.nz))
        delta = np.zeros((n, self.nz))
        self._check(self.lib.cimpc_ip_linear_solve(self.h, int(t), n, _dp(z), _dp(r), float(reg), _dp(delta)), "ip_linear_solve")
        return delta

    # -- B1: linear_solve!(solver, Delta, R, r) -----------------------------------------------
    def kkt_solve(self, r, beta):
        r = _f64(r, (self.B, self.N))
        delta = np.zeros_like(r)
        self._check(self.lib.cimpc_kkt_solve(self.h, _dp(r), float(beta), _dp(delta)), "kkt_solve")
        return delta

    def kkt_solve_rho(self, r, rho):
        """The same with the dual regularisation rho = -R[N, N] given directly (what a LinearSolver sees at newton.jl:218)."""
        r = _f64(r, (self.B, self.N))
        delta = np.zeros_like(r)
        self._check(self.lib.cimpc_kkt_solve_rho(self.h, _dp(r), float(rho), _dp(delta)), "kkt_solve_rho")
        return delta

    # -- B4: newton_solve! ----------------------------------------------------------------------
    def newton_solve(self, q0, q1, warm_start=False):
        """newton_solve!(core, s, q0, q1, window, im_traj, ref_traj; warm_start).
        Returns (u1 (B,nu), newton_iters (B,), r_norm (B,))."""
        q0 = _f64(q0, (self.B, self.nq)); q1 = _f64(q1, (self.B, self.nq))
        u1 = np.zeros((self.B, self.nu)); it = np.zeros(self.B, dtype=np.int32); rn = np.zeros(self.B)
        self._check(self.lib.cimpc_newton_solve(self.h, _dp(q0), _dp(q1), int(bool(warm_start)),
                                                _dp(u1), _ipt(it), _dp(rn)), "newton_solve")
        return u1, it, rn

    def mpc_solve(self, q0, q1, window=None, reference=None, alt=None, warm_start=False, which=("q", "u", "nu")):
        """One MPC step in ONE C call (cimpc_mpc_solve): optional new window (B, H+2) 1-based, reference = dict / tuple of
        (q, u, w, gamma, b, theta) as `set_reference`, altitude (B, nc); returns (u1, newton_iters, r_norm, trajectory dict)."""
        B, H = self.B, self.H
        q0 = _f64(q0, (B, self.nq)); q1 = _f64(q1, (B, self.nq))
        w = None
        if window is not None:
            w = np.ascontiguousarray(window, dtype=np.int32)
            if w.shape != (B, H + 2):
                raise ValueError(f"window shape {w.shape} != {(B, H + 2)}")
        ref = [None] * 6
        if reference is not None:
            r = [reference[k] for k in ("q", "u", "w", "gamma", "b", "theta")] if isinstance(reference, dict) else list(reference)
            shp = [(B, H + 2, self.nq), (B, H, self.nu), (B, H, self.nw), (B, H, self.nc), (B, H, self.nb), (B, H, self.nth)]
            ref = [None if a is None else _f64(a, s) for a, s in zip(r, shp)]
        al = _f64(alt, (B, self.nc)) if alt is not None else None
        u1 = np.zeros((B, self.nu)); it = np.zeros(B, dtype=np.int32); rn = np.zeros(B)
        shapes = dict(q=(B, H + 2, self.nq), u=(B, H, self.nu), gamma=(B, H, self.nc), b=(B, H, self.nb), nu=(B, H, self.nd))
        out = {k: np.zeros(shapes[k]) for k in ("q", "u", "gamma", "b", "nu") if k in which}
        P = lambda k: _dp(out[k]) if k in out else None
        self._check(self.lib.cimpc_mpc_solve(self.h, _ipt(w), *[_dp(a) for a in ref], _dp(al), _dp(q0), _dp(q1), int(bool(warm_start)),
                                             _dp(u1), _ipt(it), _dp(rn), P("q"), P("u"), P("gamma"), P("b"), P("nu")), "mpc_solve")
        return u1, it, rn, out

    def newton_solve_dev(self, q0_ptr, q1_ptr, warm_start=False):
        """Same, q0/q1 already in device memory (raw pointers, e.g. torch_tensor.data_ptr())."""
        self._check(self.lib.cimpc_newton_solve_dev(self.h, C.c_void_p(q0_ptr), C.c_void_p(q1_ptr),
                                                    int(bool(warm_start))), "newton_solve_dev")

    def newton_info(self):
        u1 = np.zeros((self.B, self.nu)); it = np.zeros(self.B, dtype=np.int32); rn = np.zeros(self.B)
        self._check(self.lib.cimpc_get_newton_info(self.h, _ipt(it), _dp(rn), _dp(u1)), "get_newton_info")
        return u1, it, rn

    def trajectory(self, which=("q", "u", "gamma", "b", "nu")):
        """core.traj after the last solve.  `which` limits the download (cimpc_get_trajectory skips NULL outputs: a control loop
        that only wants the planned configurations asks for ("q",) - one device-to-host copy instead of five)."""
        B, H = self.B, self.H
        shapes = dict(q=(B, H + 2, self.nq), u=(B, H, self.nu), gamma=(B, H, self.nc), b=(B, H, self.nb), nu=(B, H, self.nd))
        out = {k: np.zeros(shapes[k]) for k in ("q", "u", "gamma", "b", "nu") if k in which}
        P = lambda k: _dp(out[k]) if k in out else None
        self._check(self.lib.cimpc_get_trajectory(self.h, P("q"), P("u"), P("gamma"), P("b"), P("nu")), "get_trajectory")
        return out

    def mpc_advance(self, stride):
        """rot_n_stride!(p.traj, ..., p.stride, p.window) + update_window! (policy.jl:136-139) on the device."""
        st = _f64(stride, (self.nq,))
        self._check(self.lib.cimpc_mpc_advance(self.h, _dp(st)), "mpc_advance")

    def set_gait(self, q, u, theta, stride, w=None, gamma=None, b=None, phase=None):
        """The controller's full reference trajectory (p.traj of the CIMPC policy): H_ref knots shared by the
        rollouts; builds every rollout's window and horizon reference on the device (phase = steps already
        rotated, per rollout) and switches `mpc_advance` to regenerate them from the gait."""
        K = self.H_ref
        opt = lambda a, n: None if a is None else _f64(a, (K, n))
        qa, ua, tha = _f64(q, (K + 2, self.nq)), _f64(u, (K, self.nu)), _f64(theta, (K, self.nth))
        wa, ga, ba = opt(w, self.nw), opt(gamma, self.nc), opt(b, self.nb)
        st = _f64(stride, (self.nq,))
        ph = None if phase is None else np.ascontiguousarray(np.asarray(phase, dtype=np.int32).reshape(self.B))
        P = lambda a: None if a is None else _dp(a)
        self._check(self.lib.cimpc_set_gait(self.h, _dp(qa), _dp(ua), P(wa), P(ga), P(ba), _dp(tha), _dp(st),
                                            None if ph is None else _ipt(ph)), "set_gait")

    def reference(self):
        B, H = self.B, self.H
        q = np.zeros((B, H + 2, self.nq)); u = np.zeros((B, H, self.nu)); w = np.zeros((B, H, self.nw))
        g = np.zeros((B, H, self.nc)); b = np.zeros((B, H, self.nb)); th = np.zeros((B, H, self.nth))
        win = np.zeros((B, H + 2), dtype=np.int32)
        self._check(self.lib.cimpc_get_reference(self.h, _dp(q), _dp(u), _dp(w), _dp(g), _dp(b), _dp(th), _ipt(win)),
                    "get_reference")
        return dict(q=q, u=u, w=w, gamma=g, b=b, theta=th, window=win)

    def rollout_counters(self):
        sw = np.zeros(self.B, dtype=np.int32); it = np.zeros(self.B, dtype=np.int32)
        fl = np.zeros(self.B, dtype=np.int32)
        self._check(self.lib.cimpc_get_rollout_counters(self.h, _ipt(sw), _ipt(it), _ipt(fl)),
                    "get_rollout_counters")
        return dict(sweeps=sw, ip_iters=it, ip_failures=fl)

    def newton_log(self, max_entries=16):
        """Status lines of the last newton_solve! (print_status, newton.jl:290-301): (B, max_entries, 4) =
        alpha, |r|_1/N before, after, line-search iterate per accepted Newton iteration."""
        log = np.zeros((self.B, max_entries, 4))
        self._check(self.lib.cimpc_get_newton_log(self.h, _dp(log), int(max_entries)), "get_newton_log")
        return log

    def kkt_fallbacks(self):
        """kkt_backend = 3 (mixed precision): KKT systems that went to the fp64 fallback since the handle was created."""
        n = C.c_longlong()
        self._check(self.lib.cimpc_get_kkt_fallbacks(self.h, C.byref(n)), "get_kkt_fallbacks")
        return n.value

    def kkt_twisted(self):
        """KKT launches since the handle was created that took the twisted (two-ended) condensed solve."""
        n = C.c_longlong()
        self._check(self.lib.cimpc_get_kkt_twisted(self.h, C.byref(n)), "get_kkt_twisted")
        return n.value

    def kkt_twisted_fallbacks(self):
        """Hand-overs of the twisted kernels that timed out (their KKT stages were repeated on the one-ended kernels)."""
        n = C.c_longlong()
        self._check(self.lib.cimpc_get_kkt_twisted_fallbacks(self.h, C.byref(n)), "get_kkt_twisted_fallbacks")
        return n.value

    def debug_set_tw_spins(self, spins):
        self._check(self.lib.cimpc_debug_set_tw_spins(self.h, int(spins)), "debug_set_tw_spins")

    def stats(self):
        s = _lib.Stats()
        self._check(self.lib.cimpc_get_stats(self.h, C.byref(s)), "get_stats")
        return {k: getattr(s, k) for k, _ in s._fields_}

    # -- measurement ----------------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self.lib.cimpc_profile_enable(self.h, int(on)), "profile_enable")

    def profile_reset(self):
        self._check(self.lib.cimpc_profile_reset(self.h), "profile_reset")

    def profile_read(self):
        p = _lib.Profile()
        self._check(self.lib.cimpc_profile_read(self.h, C.byref(p)), "profile_read")
        return {k: getattr(p, k) for k, _ in p._fields_}

    def query_sizes(self):
        a = C.c_int(); b = C.c_int()
        self._check(self.lib.cimpc_query_sizes(self.h, C.byref(a), C.byref(b)), "query_sizes")
        return a.value, b.value
