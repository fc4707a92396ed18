"""Host-side mirror of the reference's contact-implicit MPC policy over the device solver.

  ci_mpc_policy(traj, s, obj; H_mpc, N_sample, κ_mpc, mode, n_opts, ip_opts)     src/controller/policy.jl:42-96
  policy(p::CIMPC, traj, t)                                                       src/controller/policy.jl:98-146

The reference policy owns a rotating copy of the reference trajectory (`p.traj`), a window of knot indices and a
Newton workspace; every `N_sample` simulator steps it solves `newton_solve!(p.newton, p.s, p.q0, q1, p.window,
p.im_traj, p.traj, warm_start = t > 1)`, rotates the reference (`rot_n_stride!`), advances the window and returns
`newton.traj.u[1] / N_sample` until the next solve.  Here the trajectory, the window and the workspace live on the
GPU (`cimpc_set_gait`, `cimpc_newton_solve`, `cimpc_mpc_advance`); per step the host sends `q0, q1` (2 nq doubles per
robot) and reads `u1` back.  B robots / Monte-Carlo rollouts run in lock step (same cadence, independent states).
"""
from __future__ import annotations

import numpy as np

from .solver import CIMPCSolver, InteriorPointOptions, NewtonOptions


def update_altitude(model, alt, gamma_hist, q_hist, threshold=1.0):
    """`update_altitude!(alt, ϕ, s, traj, t, nc, N_sample; threshold)` (mpc_utils.jl:109-136) on the window of simulator
    steps the caller hands over (the reference looks at steps max(0, t-1-N_sample)+1 .. t-1): per contact point, the step
    with the largest normal impulse decides - if it exceeds `threshold`, the altitude becomes ϕ_i at the configuration
    that step ended in.  gamma_hist: (n, nc); q_hist: (n, nq) with q_hist[j] = traj.q[j+2]; alt: (nc,) updated in place."""
    import torch
    gamma_hist = np.asarray(gamma_hist, dtype=np.float64).reshape(-1, model.nc)
    q_hist = np.asarray(q_hist, dtype=np.float64).reshape(gamma_hist.shape[0], model.nq)
    for i in range(model.nc):
        g_max, idx = 0.0, -1
        for j in range(gamma_hist.shape[0]):
            if gamma_hist[j, i] > g_max:
                g_max, idx = gamma_hist[j, i], j
        if g_max > threshold:
            alt[i] = float(model.phi(torch.as_tensor(q_hist[idx]))[i])
    return alt


class CIMPCPolicy:
    def __init__(self, problem, obj_q, obj_u, H_mpc=None, N_sample=1, kappa_mpc=None, B=1, mode=0,
                 n_opts: NewtonOptions | None = None, ip_opts: InteriorPointOptions | None = None, device=0,
                 phase=None, obj_gamma=None, obj_b=None, obj_v=None, v_target=None):
        """problem: a `lcp_models.ReferenceProblem` (reference trajectory + per-knot linearization at κ_mpc);
        obj_q, obj_u: (H_mpc, nq, nq), (H_mpc, nu, nu) TrackingObjective weights; obj_gamma / obj_b: contact-force
        weights of `:configurationforce` mode (mode = 1, the default of ci_mpc_policy); obj_v (+ v_target): the
        velocity weights of a TrackingVelocityObjective."""
        m = problem.model
        self.problem = problem
        self.H = H_mpc or problem.H
        self.N_sample = int(N_sample)
        self.B = B
        kappa = problem.kappa if kappa_mpc is None else kappa_mpc
        if abs(kappa - problem.kappa) > 0:
            raise ValueError("the linearization table was built at κ = %g, policy asks for κ_mpc = %g" % (problem.kappa, kappa))
        self.solver = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, problem.H, self.H, B=B, mode=mode,
                                  ip_opts=ip_opts or InteriorPointOptions(kappa_tol=kappa),       # policy.jl:54-61
                                  newton_opts=n_opts or NewtonOptions(kappa=kappa, r_tol=3e-4, max_iter=5), device=device)
        for t in range(problem.H):
            self.solver.set_linearization(t + 1, problem.z[t], problem.theta[t], problem.r0[t], problem.rz0[t], problem.rth0[t])
        self.solver.set_objective(obj_q, obj_u, obj_gamma, obj_b, V=obj_v, v_target=v_target)
        from .lcp_models import get_stride
        self.stride = get_stride(m, problem.q)
        self.phase0 = None if phase is None else np.asarray(phase, dtype=np.int32).reshape(B)
        self.newton_iters = []
        self.reset()

    def reset(self):
        """policy(p, traj, t = 1): set_trajectory!(p.traj, p.ref_traj), reset_window!, p.q0 = ref_traj.q[1], cnt = N_sample."""
        P = self.problem
        self.solver.set_gait(P.q, P.u, P.theta, self.stride, w=P.w, gamma=P.gamma, b=P.b, phase=self.phase0)
        ph = np.zeros(self.B, dtype=np.int64) if self.phase0 is None else self.phase0.astype(np.int64)
        self.q0 = np.stack([P.q[0] if p == 0 else P.q[(p - 1) % P.H + 1] + ((p - 1) // P.H) * self.stride for p in ph])
        self.cnt = self.N_sample
        self.solves = 0
        self.u = np.zeros((self.B, P.model.nu))

    def __call__(self, q1):
        """One simulator step: q1 = current configurations (B, nq) (`traj.q[t+1]`).  Returns the controls (B, nu)."""
        q1 = np.asarray(q1, dtype=np.float64).reshape(self.B, -1)
        if self.cnt == self.N_sample:
            u1, it, rn = self.solver.newton_solve(self.q0, q1, warm_start=self.solves > 0)
            self.newton_iters.append(it.copy())
            self.solver.mpc_advance(self.stride)            # rot_n_stride! + update_window!
            self.q0 = q1.copy()
            self.cnt = 0
            self.solves += 1
            self.u = u1 / self.N_sample                      # policy.jl:142-144 (:direct)
        self.cnt += 1
        return self.u

    def set_altitude(self, alt):
        """`set_altitude!(p.im_traj, p.altitude)` (policy.jl:116, implicit_dynamics.jl:141-154): (B, nc) or (nc,)."""
        alt = np.broadcast_to(np.asarray(alt, dtype=np.float64), (self.B, self.problem.model.nc))
        self.solver.set_altitude(np.ascontiguousarray(alt))

    def close(self):
        self.solver.close()
