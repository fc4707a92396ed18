"""Interior-point solve of one linearized contact-LCP knot + the batched
`implicit_dynamics!` loop.  CPU restatement (numpy, fp64); test infrastructure only.

PARITY UNPINNED for the iteration control: `interior_point_solve!` is owned by
RoboDojo.jl 0.1.3 (Project.toml:29,54; call sites implicit_dynamics.jl:169,175),
whose source is not under /root/reference and cannot be run here (no Julia).
The loop below is BUILD-DEFINED, modelled on RoboDojo 0.1.3's Mehrotra
predictor-corrector, using only the callback contract the reference does ship
(SURVEY.md section 3.5):
    r!  = rlin!                       linearized_solver.jl:364-373
    rz! = rzlin!                      linearized_solver.jl:378-399, 567-571
    linear_solve!(D, rz, r; reg)      linearized_solver.jl:424-444, 489-495
    linear_solve!(dz, rz, rth; reg)   linearized_solver.jl:451-487
    general_correction_term!          linearized_solver.jl:411-418
    residual_violation / bilinear_violation   :401-409
Every control decision marked [spec] is this build's frozen specification
(DESIGN.md section "IP iteration spec"); the HIP kernel implements the same spec.
"""
from dataclasses import dataclass

import numpy as np

from . import lcp
from .dims import Dims, MODE_CONFIGURATIONFORCE


@dataclass
class IPOptions:
    """Mirror of RoboDojo InteriorPointOptions fields used on this path
    (policy.jl:54-61, implicit_dynamics.jl:25-32)."""
    r_tol: float = 1.0e-8
    kappa_tol: float = 2.0e-4
    undercut: float = 5.0
    gamma_reg: float = 0.1
    kappa_reg: float = 1.0e-3
    eps_min: float = 0.05
    ls_scale: float = 0.5
    max_iter: int = 100
    max_ls: int = 3
    diff_sol: bool = True
    stall_alpha: float = 1.0e-13   # [spec] stall exit, 0 disables (strict max_iter semantics)


def step_length(y1, y2, dy1, dy2, tau):
    """[spec] largest a in (0,1] with y - a*dy >= (1-tau)*y on the orthant pairs
    (index_ort, index.jl:332-340): a = min(1, tau*y_k/dy_k for dy_k > 0)."""
    a = 1.0
    for y, dy in ((y1, dy1), (y2, dy2)):
        m = dy > 0.0
        if np.any(m):
            a = min(a, float(np.min(tau * y[m] / dy[m])))
    return a


def centering(y1, y2, dy1, dy2, a_aff):
    """[spec] mu = y1.y2/ny ; mu_aff after the affine step ; sigma = clamp(mu_aff/mu,0,1)^3."""
    n = y1.shape[0]
    mu = float(np.dot(y1, y2)) / n
    mu_aff = float(np.dot(y1 - a_aff * dy1, y2 - a_aff * dy2)) / n
    sigma = min(max(mu_aff / mu, 0.0), 1.0) ** 3
    return mu, sigma


def interior_point_solve(tab: lcp.LinTable, z, th, opts: IPOptions, trace=None, counters=None):
    """One `interior_point_solve!(ip[t])`.  z is updated in place.

    Returns (status, iterations, dz) with dz = dz/dth = -rz^-1 rth (nz x nth) on
    success when opts.diff_sol, else None."""
    d = tab.dims
    iy1, iy2 = d.iy1, d.iy2
    rdyn, rrst, rbil = lcp.rlin(tab, z, th, 0.0)
    r_vio = lcp.residual_violation(rdyn, rrst)
    k_vio = lcp.bilinear_violation(rbil)
    iters = 0
    reg = 0.0
    for _ in range(opts.max_iter):
        if r_vio < opts.r_tol and k_vio < opts.kappa_tol:
            break
        iters += 1
        # [spec] regularisation level
        k_vio = lcp.bilinear_violation(rbil)
        reg = k_vio * opts.gamma_reg if k_vio < opts.kappa_reg else 0.0
        # Jacobian update + Schur refactor (rz!)
        lcp.rzlin(tab, z, reg=reg)
        # predictor (affine) direction
        D = lcp.linear_solve_vec(tab, rdyn, rrst, rbil, reg=reg)
        a_aff = step_length(z[iy1], z[iy2], D[iy1], D[iy2], 1.0)
        mu, sigma = centering(z[iy1], z[iy2], D[iy1], D[iy2], a_aff)
        # corrector residual: r!(r, z, th, max(sigma*mu, kappa_tol/undercut)) + correction
        kc = max(sigma * mu, opts.kappa_tol / opts.undercut)
        rdyn, rrst, rbil = lcp.rlin(tab, z, th, kc)
        rbil = rbil + D[iy1] * D[iy2]            # general_correction_term! :411-418
        D = lcp.linear_solve_vec(tab, rdyn, rrst, rbil, reg=reg)
        # [spec] fraction to boundary
        tau = max(1.0 - opts.eps_min, 1.0 - max(r_vio, k_vio) ** 2)
        alpha = step_length(z[iy1], z[iy2], D[iy1], D[iy2], tau)
        # [spec] stall exit: a step length below fp64 resolution means the iterate is jammed on
        # the boundary (every later iteration repeats the same blocked direction with a
        # geometrically shrinking alpha); converging solves never go below ~1e-3.  The solve
        # ends with status = false exactly as it would after max_iter iterations.
        if alpha < opts.stall_alpha:
            break
        # candidate point z <- z - alpha*D, residual-decrease line search
        z -= alpha * D
        k_c = r_c = 0.0
        for i in range(1, opts.max_ls + 1):
            rdyn, rrst, rbil = lcp.rlin(tab, z, th, 0.0)
            k_c = lcp.bilinear_violation(rbil)
            r_c = lcp.residual_violation(rdyn, rrst)
            if r_c <= r_vio or k_c <= k_vio:
                break
            z += alpha * opts.ls_scale ** i * D   # [spec] back off
        else:
            # [spec] max_ls evaluations without a decrease: z has moved once more and is NOT re-evaluated - the violations
            # carried into the next iteration belong to the last evaluated point.  Kept because the upstream loop this spec
            # models is written that way; counted for tests/test_oracle_line_search.py.
            if counters is not None:
                counters["ls_exhausted"] = counters.get("ls_exhausted", 0) + 1
        k_vio, r_vio = k_c, r_c
        if trace is not None:
            trace.append((z.copy(), r_vio, k_vio, alpha, reg))
    status = bool(r_vio < opts.r_tol and k_vio < opts.kappa_tol)
    dz = None
    if status and opts.diff_sol:
        # [spec] differentiate_solution!: refactor at the solution with
        # reg = max(reg, kappa_tol*gamma_reg), dz = -(rz^-1 rth)
        reg2 = max(reg, opts.kappa_tol * opts.gamma_reg)
        lcp.rzlin(tab, z, reg=reg2)
        dz = -lcp.linear_solve_mat(tab, reg=reg2)
    return status, iters, dz


def z_initialize(dims: Dims, q):
    """simulation.jl:59-63: z .= 1; z[iq2] = q."""
    z = np.ones(dims.nz)
    z[dims.ix] = q
    return z


def knot_store(dims: Dims, n_knots: int):
    """The reference keeps ONE interior-point solver per reference knot (im_traj.ip[t], implicit_dynamics.jl:71-86): its
    sensitivity block ip[t].dz is overwritten by every successful solve of knot t and survives failed ones, candidate
    evaluations, Newton solves and window shifts of the MPC loop.  This is that per-knot memory (zeros before the first
    success, like the freshly constructed solver)."""
    nd, nq, nu = dims.nd, dims.nq, dims.nu
    return dict(dq0=np.zeros((n_knots, nd, nq)), dq1=np.zeros((n_knots, nd, nq)), du1=np.zeros((n_knots, nd, nu)))


def implicit_dynamics(dims: Dims, tables, window, q, theta, opts: IPOptions,
                      gamma=None, b=None, prev=None, store=None):
    """`implicit_dynamics!` (implicit_dynamics.jl:156-192) for one rollout.

    tables : list of LinTable over reference knots (0-based knot index)
    window : int array, length H+2, 0-based knot indices (policy.jl:154-171)
    q      : (H+2, nq) trajectory configurations   theta : (H, nth)
    store  : knot_store(): per-knot sensitivity memory - updated by successful solves, read by failed ones
    prev   : (without a store) previous output dict of the same window: source of stale sensitivities, per step
    Returns dict(d (H,nd), dq0 (H,nd,nq), dq1 (H,nd,nq), du1 (H,nd,nu),
                 status (H,), iters (H,), z (H,nz))."""
    H = len(window) - 2
    d_ = dims
    nd, nq, nu = d_.nd, d_.nq, d_.nu
    out = dict(d=np.zeros((H, nd)), dq0=np.zeros((H, nd, nq)), dq1=np.zeros((H, nd, nq)),
               du1=np.zeros((H, nd, nu)), status=np.zeros(H, dtype=np.int32),
               iters=np.zeros(H, dtype=np.int32), z=np.zeros((H, d_.nz)))
    for i in range(H):
        t = int(window[i])
        z = z_initialize(d_, q[i + 2])                      # :160-164
        status, iters, dz = interior_point_solve(tables[t], z, theta[i], opts)
        out["status"][i] = status
        out["iters"][i] = iters
        out["z"][i] = z
        if dz is not None:                                   # views :84-86
            out["dq0"][i] = dz[:nd, 0:nq]
            out["dq1"][i] = dz[:nd, nq:2 * nq]
            out["du1"][i] = dz[:nd, 2 * nq:2 * nq + nu]
            if store is not None:
                for k in ("dq0", "dq1", "du1"):
                    store[k][t] = out[k][i]
        elif store is not None:
            # failed solve: ip[t].dz is untouched - the last successful solve of knot t, whenever that was
            for k in ("dq0", "dq1", "du1"):
                out[k][i] = store[k][t]
        elif prev is not None:
            # failed solve: the reference leaves ip[t].dz untouched (stale values of the
            # last successful solve of that knot); this build keeps the slot (rollout, i).
            for k in ("dq0", "dq1", "du1"):
                out[k][i] = prev[k][i]
        dvec = z[:nd].copy()                                 # :180-190
        dvec[:nq] -= q[i + 2]
        if d_.mode == MODE_CONFIGURATIONFORCE:
            dvec[nq:nq + d_.nc] -= gamma[i]
            dvec[nq + d_.nc:] -= b[i]
        out["d"][i] = dvec
    return out
