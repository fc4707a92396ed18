#!/usr/bin/env python3
"""The reference's closed-loop tests at full length (1000 plant steps, 200 MPC solves), CPU plant (oracle/plant.py):
  --model quadruped   test/controller/mpc_quadruped.jl:1-64   (H_mpc 10, :configuration, TrackingObjective)
  --model flamingo    test/controller/mpc_flamingo.jl:1-80    (H_mpc 15, :configurationforce, TrackingVelocityObjective)
Controller: --controller oracle (CPU only) or device (the product: CIMPCPolicy over the C ABI).
Prints tracking_error next to the nominal values the reference recorded."""
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
ap.add_argument("--model", default="quadruped", choices=["quadruped", "flamingo"])
ap.add_argument("--controller", default="oracle", choices=["oracle", "device"])
ap.add_argument("--steps", type=int, default=1000)
a = ap.parse_args()
N_SAMPLE, KAPPA = 5, 2e-4
t = lambda v, H: np.tile(np.diag(np.asarray(v, dtype=float))[None], (H, 1, 1))
if a.model == "quadruped":
    H_MPC, mode, nominal, where = 10, 0, (0.0201, 0.0437, 0.374, 0.0789), "test/controller/mpc_quadruped.jl:59-62"
    d, P, prob, tabs = real_problem("quadruped", KAPPA, True)
    obj = synth.make_objective(d, H_MPC, kind="quadruped")
    plant, sim_opts = pl.QuadrupedPlant(), pl.SIM_OPTS
else:
    H_MPC, mode, nominal, where = 15, 1, (0.0154, 0.0829, 0.444, 0.0169), "test/controller/mpc_flamingo.jl:71-74"
    d, P, prob, tabs = real_problem("flamingo", KAPPA, False, 1)
    obj = onewton.Objective(q=t(1e-1 * np.array([3e2, 1e-6, 3e2, 1, 1, 1, 1, 0.1, 0.1]), H_MPC), u=t(3e-1 * np.array([0.1, 0.1, 0.3, 0.3, 2, 2]), H_MPC),
                            gamma=t(1e-100 * np.ones(4), H_MPC), b=t(1e-100 * np.ones(8), H_MPC), v=t(1e-3 * np.array([1, 1, 1e4, 1, 1, 1, 1, 1e4, 1e4]), H_MPC))
    plant = pl.FlamingoPlant()
    sim_opts = oip.IPOptions(r_tol=1e-8, kappa_tol=1e-8, undercut=np.inf, gamma_reg=0.0, eps_min=0.05, max_iter=100, max_ls=25)
q1, v1 = P.q[1].copy(), (P.q[1] - P.q[0]) / P.h
t0 = time.time()
if a.controller == "oracle":
    ref = onewton.Traj(q=P.q.copy(), u=P.u.copy(), w=P.w.copy(), gamma=P.gamma.copy(), b=P.b.copy(), theta=P.theta.copy())
    pol = pl.OraclePolicy(d, tabs, ref, prob["stride"], obj, H_MPC, N_SAMPLE, KAPPA,
                          onewton.NewtonOptions(r_tol=3e-4, max_iter=5, solver="lu"), oip.IPOptions(kappa_tol=KAPPA, r_tol=1e-8))
    ok, q, u, g, b = pl.simulate(plant, pol, q1, v1, a.steps, P.h / N_SAMPLE, opts=sim_opts)
    its = np.mean(pol.iters)
else:
    from contactimplicitmpc.jl_amd import InteriorPointOptions, NewtonOptions
    from contactimplicitmpc.jl_amd.policy import CIMPCPolicy
    pol = CIMPCPolicy(P, obj.q, obj.u, H_mpc=H_MPC, N_sample=N_SAMPLE, B=1, mode=mode, obj_gamma=obj.gamma if mode else None,
                      obj_b=obj.b if mode else None, obj_v=obj.v, n_opts=NewtonOptions(kappa=KAPPA, r_tol=3e-4, max_iter=5),
                      ip_opts=InteriorPointOptions(kappa_tol=KAPPA, r_tol=1e-8))
    t_pol = [0.0]
    def call(qq, k):
        t1 = time.perf_counter(); uu = pol(qq[k + 1][None])[0]; t_pol[0] += time.perf_counter() - t1
        return uu
    ok, q, u, g, b = pl.simulate(plant, call, q1, v1, a.steps, P.h / N_SAMPLE, opts=sim_opts)
    its = np.mean(pol.newton_iters)
    print("policy time: %.3f s for %d MPC solves (%.2f ms each); simulated time %.2f s -> %.1fx real time"
          % (t_pol[0], pol.solves, 1e3 * t_pol[0] / pol.solves, a.steps * P.h / N_SAMPLE, a.steps * P.h / N_SAMPLE / t_pol[0]))
    pol.close()
e = pl.tracking_error(P.q, P.u, P.gamma, P.b, q, u, g, b, N_SAMPLE)
print("%s, controller %s, %d plant steps, status %s, %.1f s, Newton iterations per solve %.2f" % (a.model, a.controller, a.steps, ok, time.time() - t0, its))
print("tracking_error    q %.5f  u %.5f  gamma %.4f  b %.5f" % e)
print("reference nominal q %.4f   u %.4f   gamma %.3f   b %.4f   (%s, bound 1.5x)" % (nominal + (where,)))
