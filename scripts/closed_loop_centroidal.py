#!/usr/bin/env python3
"""examples/centroidal_quadruped/continuous_trot.jl with BOTH sides on the device: the MPC policy (H_mpc = 50,
TrackingVelocityObjective of :43-47, kappa 1e-3, IP r_tol 1e-4, Newton r_tol 3e-5 / max_iter 5, N_sample = 5) and the
3-D centroidal plant, for B robots that carry different PAYLOADS (a constant body force the controller does not know
about - BASELINE configs[4]) plus the example's two open-loop pushes (:76-77).  Prints per robot: every plant step
converged?, body height / orientation drift against the reference, Newton iterations per solve.
usage: python scripts/closed_loop_centroidal.py [--steps 300] [--payload 0 -5 -15 -30]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from contactimplicitmpc.jl_amd import InteriorPointOptions, NewtonOptions, gait_io, lcp_models, plant  # noqa: E402
from contactimplicitmpc.jl_amd.policy import CIMPCPolicy  # noqa: E402


def run(steps=300, payload=(0.0, -5.0, -15.0, -30.0), H_mpc=50, N_sample=5, pushes=True, verbose=True):
    kappa = 1e-3
    m = lcp_models.CentroidalQuadruped()
    gait = gait_io.load_gait(os.path.join(ROOT, "tests", "golden", "gaits", "centroidal_inplace_trot_v7.jld2"))
    P = lcp_models.reference_problem(m, gait, kappa)
    B = len(payload)
    tile = lambda M: np.tile(np.asarray(M, dtype=float)[None], (H_mpc, 1, 1))
    obj_q = tile(lcp_models.relative_state_cost(1.0 * np.array([0.0, 1, 1]), 0.3 * np.ones(3), 1.0 * np.array([0.2, 0.2, 1])))
    obj_u = tile(np.diag(3e-3 * np.ones(12)))
    obj_v = tile(np.diag(1e-3 * np.concatenate([np.ones(3), 1e3 * np.ones(3), np.ones(12)])))
    pol = CIMPCPolicy(P, obj_q, obj_u, H_mpc=H_mpc, N_sample=N_sample, B=B, mode=0, obj_v=obj_v,
                      n_opts=NewtonOptions(kappa=kappa, r_tol=3e-5, max_iter=5),
                      ip_opts=InteriorPointOptions(kappa_tol=kappa, r_tol=1e-4, undercut=5.0))
    h_sim = P.h / N_sample
    w_const = np.zeros((B, 3)); w_const[:, 2] = np.asarray(payload) * h_sim          # impulse per simulator step
    push = plant.OpenLoopDisturbance([np.array([0.0, 5.0 if i == 99 else 0.0, 1.0 if i == 9 else 0.0]) for i in range(steps // N_sample + 2)],
                                     N_sample) if pushes else None
    dist = (lambda t: w_const + (push(t) if push is not None else 0.0))
    q1 = np.tile(P.q[1], (B, 1)); v1 = np.tile((P.q[1] - P.q[0]) / P.h, (B, 1))
    t0 = time.perf_counter()
    ok, q, u, g, b = plant.simulate("centroidal_quadruped", pol, q1, v1, steps, h_sim, mu=m.mu_world, disturbances=dist)
    dt = time.perf_counter() - t0
    its = np.mean(np.stack(pol.newton_iters), axis=0)
    solves = pol.solves
    pol.close()
    ref_h = P.q[:, 2].mean()
    out = []
    for r in range(B):
        dz = q[:, r, 2] - ref_h
        out.append(dict(payload_N=float(payload[r]), height_drift_max=float(np.abs(dz).max()), height_end=float(dz[-1]),
                        orientation_max=float(np.abs(q[:, r, 3:6]).max()), newton_iters=float(its[r])))
        if verbose:
            print("robot %d payload %+6.1f N: body height drift max %.4f m (end %+.4f), |orientation| max %.4f rad, Newton iterations per solve %.2f"
                  % (r, payload[r], out[-1]["height_drift_max"], out[-1]["height_end"], out[-1]["orientation_max"], its[r]))
    if verbose:
        print("all plant steps converged: %s; %d plant steps, %d MPC solves of %d robots in %.2f s" % (ok, steps, solves, B, dt))
    return ok, out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--payload", type=float, nargs="*", default=[0.0, -5.0, -15.0, -30.0])
    ap.add_argument("--horizon", type=int, default=50)
    a = ap.parse_args()
    run(a.steps, tuple(a.payload), H_mpc=a.horizon)
