"""Shared helpers of the parity tests: build the SAME seeded synthetic problem for the CPU
oracle (oracle/) and for the HIP library (through the C ABI)."""
import numpy as np

from oracle import ip as oip
from oracle import lcp, newton as onewton, synth
from oracle.dims import (Dims, QUADRUPED, HOPPER_2D, PUSHBOT, CENTROIDAL, FLAMINGO, HOPPER_3D, WALLEDCARTPOLE, PARTICLE,  # noqa: F401
                         PARTICLE_2D, CENTROIDAL_WALL)

MODELS = dict(quadruped=QUADRUPED, hopper=HOPPER_2D, pushbot=PUSHBOT, centroidal=CENTROIDAL, flamingo=FLAMINGO, hopper3d=HOPPER_3D,
              walledcartpole=WALLEDCARTPOLE, particle=PARTICLE, particle2d=PARTICLE_2D, centroidal_wall=CENTROIDAL_WALL,
              anydims=dict(nq=5, nu=3, nw=2, nc=3, nb=6))       # a dimension set no kernel was compiled for


def make_case(model="quadruped", mode=0, H_ref=12, H=6, B=3, seed=0, perturb=2e-2, kappa=2e-4):
    d = Dims(**MODELS[model], mode=mode)
    prob = synth.make_problem(d, H_ref, seed=seed, kappa=kappa)
    tabs = [lcp.LinTable(d, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
            for t in range(H_ref)]
    rollouts = []
    rng = np.random.default_rng(seed + 1000)
    for b in range(B):
        phase = int(rng.integers(0, H_ref))
        rollouts.append(synth.make_rollout(d, prob, H, phase=phase, seed=seed * 977 + b, perturb=perturb))
    return d, prob, tabs, rollouts


def make_solver(d, prob, rollouts, H, ip_opts=None, newton_opts=None, obj=None):
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
    H_ref = prob["z0"].shape[0]
    B = len(rollouts)
    s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=B, mode=d.mode,
                    ip_opts=ip_opts or InteriorPointOptions(kappa_tol=prob["kappa"]),
                    newton_opts=newton_opts or NewtonOptions(kappa=prob["kappa"]))
    for t in range(H_ref):
        s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
    s.set_window(np.stack([w for (w, _, _, _) in rollouts]) + 1)
    s.set_reference(np.stack([r.q for (_, r, _, _) in rollouts]), np.stack([r.u for (_, r, _, _) in rollouts]),
                    np.stack([r.w for (_, r, _, _) in rollouts]), np.stack([r.gamma for (_, r, _, _) in rollouts]),
                    np.stack([r.b for (_, r, _, _) in rollouts]), np.stack([r.theta for (_, r, _, _) in rollouts]))
    if obj is not None:
        s.set_objective(obj.q, obj.u, obj.gamma, obj.b, V=obj.v, q_target=obj.q_target if obj.v is not None else None,
                        v_target=obj.v_target if obj.v is not None else None)
    return s


def oracle_sweep(d, tabs, rollouts, opts, traj_of=None):
    """implicit_dynamics! of the oracle for every rollout; evaluates the rollout's reference
    trajectory with (q0, q1) substituted (the state newton_solve! sweeps first)."""
    outs = []
    for (window, ref, q0, q1) in rollouts:
        tr = ref.copy()
        tr.q[0], tr.q[1] = q0, q1
        tr.update_theta(d, 0)
        tr.update_theta(d, 1)
        outs.append((tr, oip.implicit_dynamics(d, tabs, window, tr.q, tr.theta, opts, gamma=tr.gamma, b=tr.b)))
    return outs


def oracle_scatter(d, tabs, rollout, opts, K=4, seed=0):
    """How far the ORACLE's own converged answers move when theta is perturbed in the last place (theta * (1 +- 2^-52),
    K draws): per solve the largest deviation of z, d and the sensitivities among the draws that keep (status, iters).
    A converged interior-point iterate is unique only up to the termination tolerances; ill-conditioned solves amplify
    round-off well above the nominal 1e-7.  Device-vs-oracle tolerances are max(nominal, 50 x this scatter), so a tight
    bound applies wherever the oracle itself is stable.  Returns (base_out, dict of (H,) arrays, sensitive (H,) bool)."""
    (tr, o), = oracle_sweep(d, tabs, [rollout], opts)
    H = o["iters"].shape[0]
    sc = {k: np.zeros(H) for k in ("z", "d", "dq0", "dq1", "du1")}
    sens = np.zeros(H, dtype=bool)
    rng = np.random.default_rng(seed)
    for _ in range(K):
        th = tr.theta * (1.0 + (rng.integers(0, 2, tr.theta.shape) * 2 - 1) * 2.0 ** -52)
        p = oip.implicit_dynamics(d, tabs, rollout[0], tr.q, th, opts, gamma=tr.gamma, b=tr.b)
        same = (p["status"] == o["status"]) & (p["iters"] == o["iters"])
        sens |= ~same
        for k in sc:
            dev = np.abs(p[k] - o[k]).reshape(H, -1).max(axis=1)
            sc[k] = np.maximum(sc[k], np.where(same, dev, 0.0))
    return (tr, o), sc, sens
