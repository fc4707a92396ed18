"""centroidal_quadruped_wall (ny = 48) against centroidal_quadruped (ny = 24) on the same synthetic construction: H = 50, 64 rollouts,
one cold newton_solve! each (ms per batch step) and one implicit_dynamics! sweep (ms).  python scripts/dbg/wall_vs_centroidal.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
from common import make_case, make_solver
from oracle import synth
from contactimplicitmpc.jl_amd import NewtonOptions, InteriorPointOptions
H, H_ref, B = 50, 60, 64
for model in ("centroidal", "centroidal_wall"):
    d, prob, tabs, rollouts = make_case(model, 0, H_ref=H_ref, H=H, B=B, seed=3, perturb=5e-3, kappa=1e-3)
    obj = synth.make_objective(d, H, dense_q=True)
    s = make_solver(d, prob, rollouts, H, obj=obj, ip_opts=InteriorPointOptions(kappa_tol=1e-3, r_tol=1e-4),
                    newton_opts=NewtonOptions(kappa=1e-3, r_tol=3e-5, max_iter=5))
    q = np.stack([ro.q for (_, ro, _, _) in rollouts]); th = np.stack([ro.theta for (_, ro, _, _) in rollouts])
    s.implicit_dynamics(q, th)
    t0 = time.perf_counter()
    for _ in range(3):
        out = s.implicit_dynamics(q, th)
    t_sw = (time.perf_counter() - t0) / 3
    q0 = np.stack([r[2] for r in rollouts]); q1 = np.stack([r[3] for r in rollouts])
    s.newton_solve(q0, q1)
    t0 = time.perf_counter()
    for _ in range(3):
        u1, it, rn = s.newton_solve(q0, q1)
    t_n = (time.perf_counter() - t0) / 3
    st = s.stats()
    print("%-16s sweep (B3 seam, host copies included) %.2f ms  iters/solve %.1f ok %.2f | newton_solve %.2f ms  newton iters %.2f  ip solves %d  ip iters/solve %.1f"
          % (model, 1e3 * t_sw, out["iters"].mean(), out["status"].mean(), 1e3 * t_n, it.mean(), st["ip_solves"], st["ip_iters"] / max(1, st["ip_solves"])))
    s.close()
