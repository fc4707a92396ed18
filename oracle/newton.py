"""Horizon Newton / KKT layer: CPU restatement (numpy, fp64); test infrastructure only.

Follows:
  src/controller/trajectory.jl:21-82        ContactTraj, update_z!/update_theta!
  src/controller/objective.jl:1-47          TrackingObjective / TrackingVelocityObjective
  src/controller/newton_residual.jl:18-98   residual layout (views)
                                 :113-138   residual!
                                 :140-176   update_traj!
                                 :178-281   gradient! (4 variants)
  src/controller/newton_jacobian.jl:24-134  KKT matrix layout (views)
                                 :148-198   initialize_jacobian!/update_jacobian!/jacobian!
                                 :200-248   hessian! (4 variants)
  src/controller/newton.jl:130-167          reset!
                          :169-288          newton_solve!
  src/solver/lu.jl:4-12                     default KKT backend = dense LU of Array(R)
  src/controller/newton_structure_solver/methods.jl:386-557  condensed solve (kkt_solve_condensed)
"""
from dataclasses import dataclass, field

import numpy as np

from . import ip as ipm
from .dims import Dims, MODE_CONFIGURATION, MODE_CONFIGURATIONFORCE


# ----------------------------------------------------------------------------
# containers: the checker's own ContactTraj / copy_traj! / objective holders (trajectory.jl:1-82, newton.jl:105-128,
# objective.jl:3-47).  Not imported from the product package - see oracle/dims.py.
# ----------------------------------------------------------------------------
@dataclass
class Traj:
    """ContactTraj (trajectory.jl:1-49) restricted to what the path reads."""
    q: np.ndarray       # (H+2, nq)
    u: np.ndarray       # (H, nu)
    w: np.ndarray       # (H, nw)
    gamma: np.ndarray   # (H, nc)
    b: np.ndarray       # (H, nb)
    theta: np.ndarray   # (H, nth)

    @property
    def H(self):
        return self.u.shape[0]

    def copy(self):
        return Traj(*(a.copy() for a in (self.q, self.u, self.w, self.gamma, self.b, self.theta)))

    def update_theta(self, dims: Dims, t=None):
        """update_theta!, trajectory.jl:67-82 (0-based t; None = all)."""
        ts = range(self.H) if t is None else [t]
        for k in ts:
            if 0 <= k < self.H:
                self.theta[k, dims.iq0] = self.q[k]
                self.theta[k, dims.iq1] = self.q[k + 1]
                self.theta[k, dims.iu1] = self.u[k]
                self.theta[k, dims.iw1] = self.w[k]


def copy_traj(dst: Traj, src: Traj, H):
    """copy_traj!, newton.jl:105-128 (first H steps)."""
    dst.q[:H + 2] = src.q[:H + 2]
    dst.u[:H] = src.u[:H]
    dst.w[:H] = src.w[:H]
    dst.gamma[:H] = src.gamma[:H]
    dst.b[:H] = src.b[:H]
    dst.theta[:H] = src.theta[:H]


@dataclass
class Objective:
    """objective.jl:3-47.  q/u/gamma/b: per-step square matrices (dense allowed).
    velocity objective iff v is not None."""
    q: np.ndarray                      # (H, nq, nq)
    u: np.ndarray                      # (H, nu, nu)
    gamma: np.ndarray = None           # (H, nc, nc)
    b: np.ndarray = None               # (H, nb, nb)
    v: np.ndarray = None               # (H, nq, nq) or None
    v_target: np.ndarray = None        # (H, nq)
    q_target: np.ndarray = None        # (H, nq)

    @classmethod
    def tracking(cls, dims, H, q=None, u=None, gamma=None, b=None):
        """TrackingObjective(model, env, H; q, u, γ, b) (objective.jl:9-22): per-step weight matrices, zeros where not given."""
        z = lambda n, M: np.zeros((H, n, n)) if M is None else np.asarray(M, dtype=np.float64).reshape(H, n, n)
        return cls(q=z(dims.nq, q), u=z(dims.nu, u), gamma=z(dims.nc, gamma), b=z(dims.nb, b))

    def __post_init__(self):
        if self.v is not None:
            H, nq = self.q.shape[0], self.q.shape[1]
            if self.v_target is None:
                self.v_target = np.zeros((H, nq))
            if self.q_target is None:     # objective.jl:36-45
                if np.any(self.v_target != 0.0):
                    qt = np.zeros((H, nq))
                    for t in range(1, H):
                        qt[t] = qt[t - 1] + self.v_target[t - 1]
                    self.q_target = qt
                else:
                    self.q_target = np.zeros((H, nq))


# ----------------------------------------------------------------------------
# containers
# ----------------------------------------------------------------------------
@dataclass
class NewtonOptions:
    """newton.jl:2-11 (+ policy.jl:48-52 defaults)."""
    r_tol: float = 1.0e-5
    max_iter: int = 10
    beta_init: float = 1.0e-5
    solver: str = "lu"        # "lu" (reference default, dense) | "condensed"


@dataclass
class NewtonStats:
    iters: int = 0
    sweeps: int = 0
    ip_iters: int = 0
    ip_fail: int = 0
    r_norm: float = 0.0
    alphas: list = field(default_factory=list)


# ----------------------------------------------------------------------------
# layout helpers (0-based)
# ----------------------------------------------------------------------------
class Layout:
    """Offsets inside one primal block [u | (gamma | b) | q2] and the dual segment.
    newton_residual.jl:18-54 / 69-98, newton_jacobian.jl:24-70 / 93-134."""

    def __init__(self, dims: Dims, H: int):
        d = dims
        self.dims, self.H = d, H
        self.nr, self.nd = d.nr, d.nd
        self.N = H * (d.nr + d.nd)
        off = 0
        self.iu = np.arange(off, off + d.nu); off += d.nu
        if d.mode == MODE_CONFIGURATIONFORCE:
            self.ig = np.arange(off, off + d.nc); off += d.nc
            self.ib = np.arange(off, off + d.nb); off += d.nb
        else:
            self.ig = np.arange(0, 0)
            self.ib = np.arange(0, 0)
        self.iq = np.arange(off, off + d.nq); off += d.nq
        self.iz = np.concatenate([self.iq, self.ig, self.ib])    # [q2, gamma1, b1]

    def pu(self, i):
        return i * self.nr + self.iu

    def pg(self, i):
        return i * self.nr + self.ig

    def pb(self, i):
        return i * self.nr + self.ib

    def pq(self, i):
        return i * self.nr + self.iq

    def pz(self, i):
        return i * self.nr + self.iz

    def dual(self, i):
        return self.H * self.nr + i * self.nd + np.arange(self.nd)


# ----------------------------------------------------------------------------
# residual! / gradient!
# ----------------------------------------------------------------------------
def gradient(lay: Layout, obj: Objective, traj: Traj, ref: Traj, r):
    """gradient!, newton_residual.jl:178-281."""
    d = lay.dims
    H = lay.H
    vel = obj.v is not None
    for t in range(H):
        dq = traj.q[t + 2] - (ref.q[t + 2] + (obj.q_target[t] if vel else 0.0))
        du = traj.u[t] - ref.u[t]
        r[lay.pq(t)] += obj.q[t] @ dq
        r[lay.pu(t)] += obj.u[t] @ du
        if d.mode == MODE_CONFIGURATIONFORCE:
            r[lay.pg(t)] += obj.gamma[t] @ (traj.gamma[t] - ref.gamma[t])
            r[lay.pb(t)] += obj.b[t] @ (traj.b[t] - ref.b[t])
        if vel:
            r[lay.pq(t)] += obj.v[t] @ traj.q[t + 2]
            r[lay.pq(t)] -= obj.v[t] @ traj.q[t + 1]
            if d.mode == MODE_CONFIGURATION:      # v_target only in the configuration variant (:264,:270)
                r[lay.pq(t)] -= obj.v[t] @ obj.v_target[t]
            if t == 0:
                continue
            r[lay.pq(t - 1)] -= obj.v[t] @ traj.q[t + 2]
            r[lay.pq(t - 1)] += obj.v[t] @ traj.q[t + 1]
            if d.mode == MODE_CONFIGURATION:
                r[lay.pq(t - 1)] += obj.v[t] @ obj.v_target[t]


def residual(lay: Layout, obj: Objective, nu_dual, im, traj: Traj, ref: Traj):
    """residual!, newton_residual.jl:113-138.  `im` = implicit_dynamics output dict
    (already indexed by horizon position i; the reference indexes by knot t = window[i])."""
    r = np.zeros(lay.N)
    gradient(lay, obj, traj, ref, r)
    for i in range(lay.H):
        if i >= 2:
            r[lay.pq(i - 2)] += im["dq0"][i].T @ nu_dual[i]
        if i >= 1:
            r[lay.pq(i - 1)] += im["dq1"][i].T @ nu_dual[i]
        r[lay.pu(i)] += im["du1"][i].T @ nu_dual[i]
        r[lay.dual(i)] += im["d"][i]
        r[lay.pz(i)] -= nu_dual[i]
    return r


# ----------------------------------------------------------------------------
# jacobian!
# ----------------------------------------------------------------------------
def jacobian(lay: Layout, obj: Objective, im, beta, kappa):
    """jacobian! = initialize_jacobian! + update_jacobian! (newton_jacobian.jl:148-198),
    dense N x N.  Dual regularisation quirk: reg_du -= beta*kappa once per window
    step, i.e. the whole dual diagonal ends at -H*beta*kappa (:169-186)."""
    d = lay.dims
    H, N = lay.H, lay.N
    R = np.zeros((N, N))
    vel = obj.v is not None
    for t in range(H):                                          # hessian! :200-248
        R[np.ix_(lay.pq(t), lay.pq(t))] += obj.q[t]
        R[np.ix_(lay.pu(t), lay.pu(t))] += obj.u[t]
        if d.mode == MODE_CONFIGURATIONFORCE:
            R[np.ix_(lay.pg(t), lay.pg(t))] += obj.gamma[t]
            R[np.ix_(lay.pb(t), lay.pb(t))] += obj.b[t]
        if vel:
            R[np.ix_(lay.pq(t), lay.pq(t))] += obj.v[t]
            if t == 0:
                continue
            R[np.ix_(lay.pq(t - 1), lay.pq(t - 1))] += obj.v[t]
            R[np.ix_(lay.pq(t - 1), lay.pq(t))] -= obj.v[t]
            R[np.ix_(lay.pq(t), lay.pq(t - 1))] -= obj.v[t]
    for t in range(H):                                          # IV / ITV :157-161
        R[lay.pz(t), lay.dual(t)] -= 1.0
        R[lay.dual(t), lay.pz(t)] -= 1.0
    for i in range(H):                                          # update_jacobian! :166-189
        if i >= 2:
            R[np.ix_(lay.dual(i), lay.pq(i - 2))] += im["dq0"][i]
            R[np.ix_(lay.pq(i - 2), lay.dual(i))] += im["dq0"][i].T
        if i >= 1:
            R[np.ix_(lay.dual(i), lay.pq(i - 1))] += im["dq1"][i]
            R[np.ix_(lay.pq(i - 1), lay.dual(i))] += im["dq1"][i].T
        R[np.ix_(lay.dual(i), lay.pu(i))] += im["du1"][i]
        R[np.ix_(lay.pu(i), lay.dual(i))] += im["du1"][i].T
        idx = np.arange(H * lay.nr, N)
        R[idx, idx] -= beta * kappa
    return R


# ----------------------------------------------------------------------------
# KKT solves
# ----------------------------------------------------------------------------
def kkt_solve_lu(R, r):
    """Reference default `:lu_solver` (newton.jl:10, lu.jl:4-12): dense LU with
    partial pivoting of the densified matrix."""
    return np.linalg.solve(R, r)


def kkt_solve_condensed(lay: Layout, obj: Objective, im, beta, kappa, r):
    """Structure-exploiting solve of the SAME system R*Delta = r (result-equivalent
    to kkt_solve_lu): eliminate the primal block P (block diagonal when there is no
    velocity objective) and factor the dual Schur complement
        Y = C P^-1 C^T + rho I   (block penta-diagonal, nd x nd blocks)
    by block Cholesky - the condensing of newton_structure_solver/methods.jl:386-557
    (Y = C S^-1 C^T, compute_L!, compute_beta!, compute_y!, compute_Dnu!, compute_Dz!)
    applied to the :direct layout of newton_jacobian.jl.  Requires
    mode == :configuration and a TrackingObjective (P block diagonal, P > 0)."""
    H, nd = lay.H, lay.dims.nd
    Y, bet, recover = _condensed_system(lay, obj, im, beta, kappa, r)      # Y[i,0]=Y_ii, Y[i,1]=Y_{i,i-1}, Y[i,2]=Y_{i,i-2}
    # block Cholesky of the block-pentadiagonal SPD matrix Y = L L^T
    L0 = np.zeros((H, nd, nd)); L1 = np.zeros((H, nd, nd)); L2 = np.zeros((H, nd, nd))
    for i in range(H):
        if i >= 2:
            L2[i] = np.linalg.solve(L0[i - 2], Y[i, 2].T).T          # Y_{i,i-2} L_{i-2,i-2}^-T
        if i >= 1:
            M = Y[i, 1].copy()
            if i >= 2:
                M -= L2[i] @ L1[i - 1].T
            L1[i] = np.linalg.solve(L0[i - 1], M.T).T
        D = Y[i, 0].copy()
        if i >= 1:
            D -= L1[i] @ L1[i].T
        if i >= 2:
            D -= L2[i] @ L2[i].T
        L0[i] = np.linalg.cholesky(D)
    yv = np.zeros((H, nd))
    for i in range(H):
        v = bet[i].copy()
        if i >= 1:
            v -= L1[i] @ yv[i - 1]
        if i >= 2:
            v -= L2[i] @ yv[i - 2]
        yv[i] = np.linalg.solve(L0[i], v)
    dnu = np.zeros((H, nd))
    for i in range(H - 1, -1, -1):
        v = yv[i].copy()
        if i + 1 < H:
            v -= L1[i + 1].T @ dnu[i + 1]
        if i + 2 < H:
            v -= L2[i + 2].T @ dnu[i + 2]
        dnu[i] = np.linalg.solve(L0[i].T, v)
    return recover(dnu)                                          # primal recovery  Delta_x = P^-1 (r_p - C^T dnu)


def _condensed_system(lay: Layout, obj: Objective, im, beta, kappa, r):
    """Y (block penta-diagonal: Y[i,0] = Y_ii, Y[i,1] = Y_{i,i-1}, Y[i,2] = Y_{i,i-2}), beta and the primal recovery
    Delta_x = P^-1 (r_p - C^T dnu) of the condensed solves (compute_Y!, compute_beta!, compute_Dz!: methods.jl:386-446,
    476-500, 540-557 applied to the :direct layout).  Row block i of C: [du1_i @ u_i, -I @ q_{i+2} (blk i), dq1_i @ blk i-1,
    dq0_i @ blk i-2]."""
    d = lay.dims
    assert d.mode == MODE_CONFIGURATION and obj.v is None
    H, nd = lay.H, d.nd
    rho = H * beta * kappa
    Qi = [np.linalg.inv(obj.q[t]) for t in range(H)]
    Ri = [np.linalg.inv(obj.u[t]) for t in range(H)]
    rp_u = [r[lay.pu(i)] for i in range(H)]
    rp_q = [r[lay.pq(i)] for i in range(H)]
    rd = [r[lay.dual(i)] for i in range(H)]
    Y = np.zeros((H, 3, nd, nd))
    bet = np.zeros((H, nd))
    for i in range(H):
        A0 = im["du1"][i]
        Yii = A0 @ Ri[i] @ A0.T + Qi[i] + rho * np.eye(nd)
        bi = A0 @ (Ri[i] @ rp_u[i]) - Qi[i] @ rp_q[i] - rd[i]
        if i >= 1:
            A1 = im["dq1"][i]
            Yii += A1 @ Qi[i - 1] @ A1.T
            bi += A1 @ (Qi[i - 1] @ rp_q[i - 1])
            Y[i, 1] = -A1 @ Qi[i - 1]
        if i >= 2:
            A2 = im["dq0"][i]
            Yii += A2 @ Qi[i - 2] @ A2.T
            bi += A2 @ (Qi[i - 2] @ rp_q[i - 2])
            Y[i, 1] += A2 @ Qi[i - 2] @ im["dq1"][i - 1].T
            Y[i, 2] = -A2 @ Qi[i - 2]
        Y[i, 0] = Yii
        bet[i] = bi

    def recover(dnu):
        Delta = np.zeros(lay.N)
        for i in range(H):
            Delta[lay.pu(i)] = Ri[i] @ (rp_u[i] - im["du1"][i].T @ dnu[i])
            cq = -dnu[i].copy()
            if i + 1 < H:
                cq += im["dq1"][i + 1].T @ dnu[i + 1]
            if i + 2 < H:
                cq += im["dq0"][i + 2].T @ dnu[i + 2]
            Delta[lay.pq(i)] = Qi[i] @ (rp_q[i] - cq)
            Delta[lay.dual(i)] = dnu[i]
        return Delta
    return Y, bet, recover


def _penta_eliminate(Yd, Y1, Y2, b, n):
    """Block Cholesky of the first n block rows of a block penta-diagonal SPD matrix (Yd[i] = Y_ii, Y1[i] = Y_{i,i-1},
    Y2[i] = Y_{i,i-2}) with the right-hand side riding along, and its traces in the two block rows that follow: returns
    (L0, L1, L2, y, S, c) with S = the 2 x 2 block Schur complement on rows n, n + 1 restricted to what the eliminated rows
    contribute (to be SUBTRACTED from the matrix there) and c likewise for the right-hand side."""
    nd = Yd.shape[1]
    K = n + 2
    L0 = np.zeros((K, nd, nd)); L1 = np.zeros((K, nd, nd)); L2 = np.zeros((K, nd, nd)); y = np.zeros((K, nd))
    S = np.zeros((2, 2, nd, nd)); c = np.zeros((2, nd))
    for i in range(min(K, len(Yd))):
        if i >= 2 and i - 2 < n:
            L2[i] = np.linalg.solve(L0[i - 2], Y2[i].T).T
        if i >= 1 and i - 1 < n:
            M = Y1[i].copy()
            if i >= 2 and i - 2 < n:
                M -= L2[i] @ L1[i - 1].T
            L1[i] = np.linalg.solve(L0[i - 1], M.T).T
        if i < n:
            D = Yd[i].copy()
            v = b[i].copy()
            if i >= 1:
                D -= L1[i] @ L1[i].T; v -= L1[i] @ y[i - 1]
            if i >= 2:
                D -= L2[i] @ L2[i].T; v -= L2[i] @ y[i - 2]
            L0[i] = np.linalg.cholesky(D)
            y[i] = np.linalg.solve(L0[i], v)
    # what rows n, n + 1 receive from the eliminated columns (n - 2, n - 1)
    a, bb = n, n + 1
    if a < len(Yd):
        if n >= 1:
            S[0, 0] += L1[a] @ L1[a].T; c[0] += L1[a] @ y[n - 1]
        if n >= 2:
            S[0, 0] += L2[a] @ L2[a].T; c[0] += L2[a] @ y[n - 2]
    if bb < len(Yd) and n >= 1:
        S[1, 1] += L2[bb] @ L2[bb].T; c[1] += L2[bb] @ y[n - 1]
        S[1, 0] += L2[bb] @ L1[a].T
    return L0, L1, L2, y, S, c


def kkt_solve_condensed_twisted(lay: Layout, obj: Objective, im, beta, kappa, r, split=None):
    """The condensed solve with a TWISTED (two-ended) block factorisation of Y: block rows 0 .. m-1 are eliminated from the top,
    rows H-1 .. m+2 from the bottom (the same recurrence on the reversed matrix), the two meet in a 2 x 2 block system for rows
    m, m+1, and the substitutions run outwards from there - two independent chains of about H / 2 block steps instead of one of H
    (the factor recurrences are those of newton_structure_solver/methods.jl:466-557).  Result-equivalent to
    `kkt_solve_condensed`; groundwork for a two-wave device kernel."""
    H, nd = lay.H, lay.dims.nd
    if H < 4:
        return kkt_solve_condensed(lay, obj, im, beta, kappa, r)
    Y, bet, recover = _condensed_system(lay, obj, im, beta, kappa, r)
    m = (H - 2) // 2 if split is None else split
    assert 0 <= m <= H - 2
    nb = H - m - 2                                               # rows eliminated from the bottom
    # top: the matrix as it is
    T = _penta_eliminate(Y[:, 0], Y[:, 1], Y[:, 2], bet, m)
    # bottom: the reversed matrix  Y'_{a,b} = Y_{H-1-a, H-1-b}:  Y'_{i,i-1} = Y_{H-i, H-1-i}^T,  Y'_{i,i-2} = Y_{H+1-i, H-1-i}^T
    Yd_r = Y[::-1, 0].copy()
    Y1_r = np.zeros_like(Yd_r); Y2_r = np.zeros_like(Yd_r)
    for i in range(1, H):
        Y1_r[i] = Y[H - i, 1].T
    for i in range(2, H):
        Y2_r[i] = Y[H + 1 - i, 2].T
    Bm = _penta_eliminate(Yd_r, Y1_r, Y2_r, bet[::-1].copy(), nb)
    # middle: rows m, m+1 (reversed indices: m+1 -> nb, m -> nb + 1)
    S = np.zeros((2 * nd, 2 * nd)); c = np.zeros(2 * nd)
    S[:nd, :nd] = Y[m, 0] - T[4][0, 0] - Bm[4][1, 1]
    S[nd:, nd:] = Y[m + 1, 0] - T[4][1, 1] - Bm[4][0, 0]
    S[nd:, :nd] = Y[m + 1, 1] - T[4][1, 0] - Bm[4][1, 0].T
    S[:nd, nd:] = S[nd:, :nd].T
    c[:nd] = bet[m] - T[5][0] - Bm[5][1]
    c[nd:] = bet[m + 1] - T[5][1] - Bm[5][0]
    mid = np.linalg.solve(S, c)
    dnu = np.zeros((H, nd))
    dnu[m], dnu[m + 1] = mid[:nd], mid[nd:]
    # outwards: the top chain upwards ...
    L0, L1, L2, y = T[0], T[1], T[2], T[3]
    for i in range(m - 1, -1, -1):
        v = y[i].copy()
        v -= L1[i + 1].T @ dnu[i + 1]
        v -= L2[i + 2].T @ dnu[i + 2]
        dnu[i] = np.linalg.solve(L0[i].T, v)
    # ... and the bottom chain downwards (reversed indices)
    L0, L1, L2, y = Bm[0], Bm[1], Bm[2], Bm[3]
    dr = dnu[::-1]                                               # a view: dr[a] = dnu[H-1-a]
    for i in range(nb - 1, -1, -1):
        v = y[i].copy()
        v -= L1[i + 1].T @ dr[i + 1]
        v -= L2[i + 2].T @ dr[i + 2]
        dr[i] = np.linalg.solve(L0[i].T, v)
    return recover(dnu)


def kkt_tw_split(H):
    """Rows the top chain of the device's twisted solve eliminates before the two middle rows (csrc/newton_impl.h: kkt_tw_split):
    the bottom chain (its stage A forms two more products per step) gets the shorter half, and the top chain reaches the
    middle rows only after the bottom chain's traces can have arrived."""
    nb = max(2, (H - 8) // 2)
    return H - 2 - nb


def kkt_solve_condensed_twisted_device(lay: Layout, obj: Objective, im, beta, kappa, r, split=None):
    """The twisted condensed solve AS THE DEVICE KERNEL RUNS IT (csrc/newton_impl.h: kkt_body<..., TW = 1 / 2>,
    kkt_kernel_twisted): same mathematics as `kkt_solve_condensed_twisted`, restated with the kernel's data flow so that
    its index algebra is checked on the CPU -
      * the BOTTOM chain walks the rows downwards (chain step i = row j = H-1-i) and forms its blocks of the reversed matrix
        directly from the operands it streams: Y'_ii = Y_jj, Y'_{i,i-1} = Y_{j+1,j}^T = -Qi_j dq1_{j+1}^T + (dq1_j Qi_{j-1}) dq0_{j+1}^T,
        Y'_{i,i-2} = Y_{j+2,j}^T = -Qi_j dq0_{j+2}^T, with the weights and r_p(q) of row j-2 entering two rows ahead (rings of three
        indexed by the chain step modulo 3, the dq0 ring likewise, the dq1 ring by parity);
      * its last two steps (rows m+1, m) are TRACE steps: L2', L1' only, from which S00 = L1 L1^T + L2 L2^T, S11 = L2 L2^T,
        S10^T = L1'(nb) L2'(nb+1)^T, c0 = L1 y'(nb-1) + L2 y'(nb-2), c1 = L2'(nb+1) y'(nb-1) go to the top chain;
      * the TOP chain is the one-ended recursion stopped after row m+1, with Y_mm -= S11, beta_m -= c1, Y_{m+1,m+1} -= S00,
        Y_{m+1,m} -= S10^T, beta_{m+1} -= c0 before the factor of those rows;
      * both backward passes run on the records [W1 | W2 | yhat] of stage C: dnu_i = yhat_i - W1_i^T dnu_{i+1} - W2_i^T dnu_{i+2}
        with W1_i = L1_{i+1} L0_i^-1, W2_i = L2_{i+2} L0_i^-1, yhat_i = L0_i^-T y_i (chain-local indices; the bottom chain starts
        from dnu_{m+1}, dnu_m handed over by the top chain);
      * the top chain recovers the primal rows 0 .. m-1, the bottom chain rows m .. H-1."""
    d = lay.dims
    assert d.mode == MODE_CONFIGURATION and obj.v is None
    H, nd, nq = lay.H, d.nd, d.nq
    m = kkt_tw_split(H) if split is None else split
    nb = H - m - 2
    assert nb >= 2 and m >= 2
    rho = H * beta * kappa
    Qi = [np.linalg.inv(obj.q[t]) for t in range(H)]
    Ri = [np.linalg.inv(obj.u[t]) for t in range(H)]
    du1, dq1, dq0 = im["du1"], im["dq1"], im["dq0"]
    rpu = [r[lay.pu(i)] for i in range(H)]; rpq = [r[lay.pq(i)] for i in range(H)]; rd = [r[lay.dual(i)] for i in range(H)]
    eye = np.eye(nd)

    class Chain:      # the rings of one chain (slot = chain step modulo ring length), its records and its y ring
        def __init__(self, n):
            self.Li = [None] * 3; self.LiT = [None] * 4; self.L1 = [None] * 2; self.L2 = [None] * 3; self.y = [None] * 3
            self.bet = [None] * 3; self.Y0 = [None] * 2; self.Y1 = [None] * 2
            self.rec = [dict(W1=np.zeros((nd, nd)), W2=np.zeros((nd, nd)), yh=np.zeros(nd)) for _ in range(n)]

    def stage_b(c, i, full=True):                 # the dependency chain: L1_i, L0_i and its inverse
        y0, y1 = c.Y0[i & 1], c.Y1[i & 1]
        if i >= 1:
            if i >= 2:
                y1 = y1 - c.L2[i % 3] @ c.L1[(i + 1) & 1].T
            c.L1[i & 1] = y1 @ c.Li[(i + 2) % 3].T
        if not full:
            return
        if i >= 1:
            y0 = y0 - c.L1[i & 1] @ c.L1[i & 1].T
        if i >= 2:
            y0 = y0 - c.L2[i % 3] @ c.L2[i % 3].T
        L0 = np.linalg.cholesky(y0)
        c.Li[i % 3] = np.linalg.inv(L0)
        c.LiT[i % 4] = c.Li[i % 3].T

    def stage_c(c, i, full=True, skip_w1=False):  # off the chain: records of the backward pass, forward substitution
        if i >= 1 and not skip_w1:
            c.rec[i - 1]["W1"] = c.L1[i & 1] @ c.LiT[(i - 1) % 4].T
        if i >= 2:
            c.rec[i - 2]["W2"] = c.L2[i % 3] @ c.LiT[(i - 2) % 4].T
        if not full:
            return
        s = c.bet[i % 3].copy()
        if i >= 1:
            s -= c.L1[i & 1] @ c.y[(i + 2) % 3]
        if i >= 2:
            s -= c.L2[i % 3] @ c.y[(i + 1) % 3]
        c.y[i % 3] = c.Li[i % 3] @ s
        c.rec[i]["yh"] = c.Li[i % 3].T @ c.y[i % 3]

    # ---- bottom chain: stage A of the reversed matrix, rings indexed by the chain step --------------------------------
    bot = Chain(nb)
    Qr = [None] * 3; rqr = [None] * 3; A1r = [None] * 2; A2r = [None] * 3
    Qr[0], Qr[1] = Qi[H - 1], Qi[H - 2]; rqr[0], rqr[1] = rpq[H - 1], rpq[H - 2]          # the preamble
    S00 = S11 = S10T = c0 = c1 = None
    for i in range(nb + 2):
        j = H - 1 - i
        m0, m1, m2, p0, p1 = i % 3, (i + 2) % 3, (i + 1) % 3, i & 1, (i + 1) & 1
        A0 = du1[j]; A1r[p0] = dq1[j]; A2r[m0] = dq0[j]
        if j >= 2:
            Qr[m1] = Qi[j - 2]; rqr[m1] = rpq[j - 2]                                       # entering two rows ahead
        Qj, Qj1, Qj2 = Qr[m0], Qr[m2], Qr[m1]
        T0 = A0 @ Ri[j]; T1 = A1r[p0] @ Qj1; T2 = A2r[m0] @ Qj2
        y0 = Qj + rho * eye + T0 @ A0.T + T1 @ A1r[p0].T + T2 @ A2r[m0].T
        y1 = np.zeros((nd, nd))
        if i >= 1:
            y1 = -(Qj @ A1r[p1].T) + T1 @ A2r[m1].T
        if i >= 2:
            Tq = Qj @ A2r[m2].T
            bot.L2[m0] = -(Tq @ bot.Li[(i + 1) % 3].T)
        bot.bet[m0] = T0 @ rpu[j] - Qj @ rqr[m0] + T1 @ rqr[m2] + T2 @ rqr[m1] - rd[j]
        bot.Y0[p0], bot.Y1[p0] = y0, y1
        if i < nb:
            stage_b(bot, i); stage_c(bot, i)
        elif i == nb:                                                                     # trace step of row m+1
            stage_b(bot, i, full=False)
            S00 = bot.L1[p0] @ bot.L1[p0].T + bot.L2[m0] @ bot.L2[m0].T
            stage_c(bot, i, full=False)
            c0 = bot.L1[p0] @ bot.y[m1] + bot.L2[m0] @ bot.y[m2]
        else:                                                                             # trace step of row m
            S11 = bot.L2[m0] @ bot.L2[m0].T
            S10T = bot.L1[p1] @ bot.L2[m0].T
            stage_c(bot, i, full=False, skip_w1=True)
            c1 = bot.L2[m0] @ bot.y[m2]
    # ---- top chain: the one-ended recursion up to row m+1 ------------------------------------------------------------
    top = Chain(m + 2)
    for i in range(m + 2):
        m0, p0 = i % 3, i & 1
        A0 = du1[i]
        T0 = A0 @ Ri[i]
        y0 = Qi[i] + rho * eye + T0 @ A0.T
        y1 = np.zeros((nd, nd))
        b = T0 @ rpu[i] - Qi[i] @ rpq[i] - rd[i]
        if i >= 1:
            T1 = dq1[i] @ Qi[i - 1]
            y0 += T1 @ dq1[i].T; y1 = -T1; b += T1 @ rpq[i - 1]
        if i >= 2:
            T2 = dq0[i] @ Qi[i - 2]
            y0 += T2 @ dq0[i].T; y1 = y1 + T2 @ dq1[i - 1].T; b += T2 @ rpq[i - 2]
            top.L2[m0] = -(T2 @ top.Li[(i + 1) % 3].T)
        if i == m:
            y0 = y0 - S11; b = b - c1
        if i == m + 1:
            y0 = y0 - S00; y1 = y1 - S10T; b = b - c0
        top.Y0[p0], top.Y1[p0], top.bet[m0] = y0, y1, b
        stage_b(top, i); stage_c(top, i)
    # ---- backward passes on the records --------------------------------------------------------------------------------
    dnu = np.zeros((H, nd))
    dt = np.zeros((m + 2, nd))
    for i in range(m + 1, -1, -1):
        s = top.rec[i]["yh"].copy()
        if i + 1 < m + 2:
            s -= top.rec[i]["W1"].T @ dt[i + 1]
        if i + 2 < m + 2:
            s -= top.rec[i]["W2"].T @ dt[i + 2]
        dt[i] = s
    dnu[:m + 2] = dt
    db = np.zeros((nb + 2, nd))
    db[nb], db[nb + 1] = dt[m + 1], dt[m]                                                  # the hand-over
    for i in range(nb - 1, -1, -1):
        db[i] = bot.rec[i]["yh"] - bot.rec[i]["W1"].T @ db[i + 1] - bot.rec[i]["W2"].T @ db[i + 2]
    for i in range(nb):
        dnu[H - 1 - i] = db[i]
    # ---- primal recovery (top chain: rows 0 .. m-1, bottom chain: rows m .. H-1) ----------------------------------------
    Delta = np.zeros(lay.N)
    for i in range(H):
        Delta[lay.pu(i)] = Ri[i] @ (rpu[i] - du1[i].T @ dnu[i])
        cq = -dnu[i].copy()
        if i + 1 < H:
            cq += dq1[i + 1].T @ dnu[i + 1]
        if i + 2 < H:
            cq += dq0[i + 2].T @ dnu[i + 2]
        Delta[lay.pq(i)] = Qi[i] @ (rpq[i] - cq)
        Delta[lay.dual(i)] = dnu[i]
    return Delta


# ----------------------------------------------------------------------------
# update_traj! / reset! / newton_solve!
# ----------------------------------------------------------------------------
def update_traj(lay: Layout, cand: Traj, traj: Traj, nu_cand, nu_dual, Delta, alpha):
    """update_traj!, newton_residual.jl:140-176 (x <- x - alpha*Delta) + update_theta!."""
    d = lay.dims
    for t in range(lay.H):
        cand.q[t + 2] = traj.q[t + 2] - alpha * Delta[lay.pq(t)]
        cand.u[t] = traj.u[t] - alpha * Delta[lay.pu(t)]
        if d.mode == MODE_CONFIGURATIONFORCE:
            cand.gamma[t] = traj.gamma[t] - alpha * Delta[lay.pg(t)]
            cand.b[t] = traj.b[t] - alpha * Delta[lay.pb(t)]
        nu_cand[t] = nu_dual[t] - alpha * Delta[lay.dual(t)]
    cand.update_theta(d)


class Newton:
    """Newton (newton.jl:17-91): workspace of one rollout."""

    def __init__(self, dims: Dims, H: int, obj: Objective, opts: NewtonOptions,
                 ip_opts: ipm.IPOptions, kappa: float, template: Traj):
        self.dims, self.H, self.obj, self.opts, self.ip_opts = dims, H, obj, opts, ip_opts
        self.kappa = kappa
        self.lay = Layout(dims, H)
        z = lambda *s: np.zeros(s)
        self.traj = Traj(z(H + 2, dims.nq), z(H, dims.nu), z(H, dims.nw), z(H, dims.nc),
                         z(H, dims.nb), template.theta[:H].copy())
        self.traj_cand = self.traj.copy()
        self.nu = z(H, dims.nd)
        self.nu_cand = z(H, dims.nd)
        self.beta = opts.beta_init
        self.im = None
        self.store = None          # per-knot sensitivity memory (ip[t].dz), created at the first sweep, kept across solves
        self.Delta = z(self.lay.N)
        self.res = z(self.lay.N)

    def reset(self, ref: Traj, q0, q1, warm_start):
        """reset!, newton.jl:130-167."""
        self.beta = self.opts.beta_init
        if not warm_start:
            self.nu[:] = 0.0
            self.nu_cand[:] = 0.0
            copy_traj(self.traj, ref, self.H)
        self.traj.q[0] = q0
        self.traj.q[1] = q1
        self.traj.update_theta(self.dims, 0)
        self.traj.update_theta(self.dims, 1)
        copy_traj(self.traj_cand, self.traj, self.H)

    def _sweep(self, tables, window, traj: Traj, stats: NewtonStats):
        if self.store is None:
            self.store = ipm.knot_store(self.dims, len(tables))
        im = ipm.implicit_dynamics(self.dims, tables, window, traj.q, traj.theta, self.ip_opts,
                                   gamma=traj.gamma, b=traj.b, store=self.store)
        stats.sweeps += 1
        stats.ip_iters += int(im["iters"].sum())
        stats.ip_fail += int((im["status"] == 0).sum())
        return im


def newton_solve(core: Newton, q0, q1, window, tables, ref: Traj, warm_start=False):
    """newton_solve!, newton.jl:169-288 (time budget omitted: max_time = inf).
    Returns NewtonStats; result in core.traj / core.nu."""
    lay, opts, obj = core.lay, core.opts, core.obj
    st = NewtonStats()
    core.reset(ref, q0, q1, warm_start)
    core.im = core._sweep(tables, window, core.traj, st)                         # :191
    core.res = residual(lay, obj, core.nu, core.im, core.traj, ref)              # :197
    r_norm = np.sum(np.abs(core.res))                                            # :198
    for _ in range(opts.max_iter):                                               # :202
        if r_norm / lay.N < opts.r_tol:                                          # :206
            break
        st.iters += 1
        if opts.solver == "lu":
            R = jacobian(lay, obj, core.im, core.beta, core.kappa)               # :210
            core.Delta = kkt_solve_lu(R, core.res)                               # :218
        else:
            core.Delta = kkt_solve_condensed(lay, obj, core.im, core.beta, core.kappa, core.res)
        alpha, it = 1.0, 0
        update_traj(lay, core.traj_cand, core.traj, core.nu_cand, core.nu, core.Delta, alpha)
        core.im = core._sweep(tables, window, core.traj_cand, st)                # :234
        res_cand = residual(lay, obj, core.nu_cand, core.im, core.traj_cand, ref)
        r_cand = np.sum(np.abs(res_cand))
        while r_cand ** 2.0 >= (1.0 - 0.001 * alpha) * r_norm ** 2.0:            # :245
            alpha *= 0.5
            it += 1
            if it > 6:
                break
            update_traj(lay, core.traj_cand, core.traj, core.nu_cand, core.nu, core.Delta, alpha)
            core.im = core._sweep(tables, window, core.traj_cand, st)
            res_cand = residual(lay, obj, core.nu_cand, core.im, core.traj_cand, ref)
            r_cand = np.sum(np.abs(res_cand))
        update_traj(lay, core.traj, core.traj, core.nu, core.nu, core.Delta, alpha)   # :273
        core.res = res_cand
        r_norm = r_cand
        st.alphas.append(alpha)
        core.beta = min(core.beta * 1.3, 1.0e2) if it > 6 else max(1.0e1, core.beta / 1.3)  # :280
    st.r_norm = float(r_norm)
    return st
