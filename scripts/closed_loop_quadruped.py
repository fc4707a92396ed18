#!/usr/bin/env python3
"""test/controller/mpc_quadruped.jl:1-64 at full length (H_sim = 1000 plant steps, 200 MPC solves), CPU plant
(oracle/plant.py).  Controller: --controller oracle (CPU only) or device (the product: CIMPCPolicy over the C ABI).
Prints tracking_error next to the reference's recorded nominal values."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ip as oip, newton as onewton, plant as pl, synth  # noqa: E402
from real_problems import real_problem  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--controller", default="oracle", choices=["oracle", "device"])
ap.add_argument("--steps", type=int, default=1000)
a = ap.parse_args()
H_MPC, N_SAMPLE, KAPPA = 10, 5, 2e-4
d, P, prob, tabs = real_problem("quadruped", KAPPA, True)
obj = synth.make_objective(d, H_MPC, kind="quadruped")
q1, v1 = P.q[1].copy(), (P.q[1] - P.q[0]) / P.h
plant = pl.QuadrupedPlant()
t0 = time.time()
if a.controller == "oracle":
    ref = onewton.Traj(q=P.q.copy(), u=P.u.copy(), w=P.w.copy(), gamma=P.gamma.copy(), b=P.b.copy(), theta=P.theta.copy())
    pol = pl.OraclePolicy(d, tabs, ref, prob["stride"], obj, H_MPC, N_SAMPLE, KAPPA,
                          onewton.NewtonOptions(r_tol=3e-4, max_iter=5, solver="lu"), oip.IPOptions(kappa_tol=KAPPA, r_tol=1e-8))
    ok, q, u, g, b = pl.simulate(plant, pol, q1, v1, a.steps, P.h / N_SAMPLE)
    its = np.mean(pol.iters)
else:
    from contactimplicitmpc.jl_amd import InteriorPointOptions, NewtonOptions
    from contactimplicitmpc.jl_amd.policy import CIMPCPolicy
    pol = CIMPCPolicy(P, obj.q, obj.u, H_mpc=H_MPC, N_sample=N_SAMPLE, B=1, n_opts=NewtonOptions(kappa=KAPPA, r_tol=3e-4, max_iter=5),
                      ip_opts=InteriorPointOptions(kappa_tol=KAPPA, r_tol=1e-8))
    t_pol = [0.0]
    def call(qq, t):
        t1 = time.perf_counter(); u = pol(qq[t + 1][None])[0]; t_pol[0] += time.perf_counter() - t1
        return u
    ok, q, u, g, b = pl.simulate(plant, call, q1, v1, a.steps, P.h / N_SAMPLE)
    its = np.mean(pol.newton_iters)
    print("policy time: %.3f s for %d MPC solves (%.2f ms each); simulated time %.2f s -> %.1fx real time"
          % (t_pol[0], pol.solves, 1e3 * t_pol[0] / pol.solves, a.steps * P.h / N_SAMPLE, a.steps * P.h / N_SAMPLE / t_pol[0]))
    pol.close()
e = pl.tracking_error(P.q, P.u, P.gamma, P.b, q, u, g, b, N_SAMPLE)
print("controller %s, %d plant steps, status %s, %.1f s, Newton iterations per solve %.2f" % (a.controller, a.steps, ok, time.time() - t0, its))
print("tracking_error   q %.5f  u %.5f  gamma %.4f  b %.5f" % e)
print("reference nominal q 0.0201   u 0.0437   gamma 0.374   b 0.0789   (test/controller/mpc_quadruped.jl:59-62, bound 1.5x)")
