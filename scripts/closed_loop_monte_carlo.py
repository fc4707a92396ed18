#!/usr/bin/env python3
"""Closed-loop Monte-Carlo on the device (examples/quadruped/monte_carlo.jl:76-92 in spirit): B quadrupeds, initial
configurations offset by U(-perturb, perturb) in the joint coordinates, the reference's MPC configuration of
test/controller/mpc_quadruped.jl (H_mpc 10, N_sample 5), policy AND plant on the MI355X (`CIMPCPolicy`, `cimpc_plant_step`).
Prints wall time per side, MPC steps/s in closed loop and the tracking errors over the batch."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import plant as pl, synth  # noqa: E402  (tracking_error and the objective constants only)
from real_problems import real_problem  # noqa: E402
from contactimplicitmpc.jl_amd import InteriorPointOptions, NewtonOptions, plant  # noqa: E402
from contactimplicitmpc.jl_amd.policy import CIMPCPolicy  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--robots", type=int, default=256)
ap.add_argument("--steps", type=int, default=150)
ap.add_argument("--perturb", type=float, default=0.03)
a = ap.parse_args()
KAPPA, H_MPC, N_SAMPLE = 2e-4, 10, 5
d, P, prob, tabs = real_problem("quadruped", KAPPA, True)
obj = synth.make_objective(d, H_MPC, kind="quadruped")
B = a.robots
pol = CIMPCPolicy(P, obj.q, obj.u, H_mpc=H_MPC, N_sample=N_SAMPLE, B=B, n_opts=NewtonOptions(kappa=KAPPA, r_tol=3e-4, max_iter=5),
                  ip_opts=InteriorPointOptions(kappa_tol=KAPPA, r_tol=1e-8))
rng = np.random.default_rng(100)
q1 = np.tile(P.q[1], (B, 1)); v1 = np.tile((P.q[1] - P.q[0]) / P.h, (B, 1))
q1[:, 3:] += rng.uniform(-a.perturb, a.perturb, (B, d.nq - 3)); q1[:, 1] += 0.02          # joints off, body 2 cm higher
t_pol = [0.0]
def policy(q):
    t0 = time.perf_counter(); u = pol(q); t_pol[0] += time.perf_counter() - t0
    return u
t0 = time.perf_counter()
ok, q, u, g, b = plant.simulate("quadruped", policy, q1, v1, a.steps, P.h / N_SAMPLE, mu=1.0)
wall = time.perf_counter() - t0
e = np.array([pl.tracking_error(P.q, P.u, P.gamma, P.b, q[:, r], u[:, r], g[:, r], b[:, r], N_SAMPLE) for r in range(B)])
print("robots %d, plant steps %d, MPC solves %d: wall %.2f s (policy %.2f s, plant + host glue %.2f s), all plant steps converged: %s"
      % (B, a.steps, pol.solves, wall, t_pol[0], wall - t_pol[0], ok))
print("closed-loop MPC steps/s %.0f, plant steps/s %.0f, Newton iterations per solve %.2f"
      % (B * pol.solves / wall, B * a.steps / wall, np.mean(pol.newton_iters)))
print("tracking error over the batch (mean / max):  q %.4f / %.4f   u %.4f / %.4f   gamma %.3f / %.3f   b %.4f / %.4f"
      % (e[:, 0].mean(), e[:, 0].max(), e[:, 1].mean(), e[:, 1].max(), e[:, 2].mean(), e[:, 2].max(), e[:, 3].mean(), e[:, 3].max()))
print("robots inside the reference's 1.5 x nominal bound (0.0201, 0.0437, 0.374, 0.0789): %d of %d"
      % (int(np.all(e < 1.5 * np.array([0.0201, 0.0437, 0.374, 0.0789]), axis=1).sum()), B))
pol.close()
