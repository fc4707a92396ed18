import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from common import make_case, make_solver
from oracle import synth
from contactimplicitmpc.jl_amd import NewtonOptions
def log(*a):
    print(*a, flush=True)
for model, mode in (("anydims", 0), ("anydims", 1), ("centroidal_wall", 0)):
    d, prob, tabs, rollouts = make_case(model, mode, H_ref=6, H=4, B=2, seed=3)
    s = make_solver(d, prob, rollouts, 4)
    q = np.stack([r.q for (_, r, _, _) in rollouts]); th = np.stack([r.theta for (_, r, _, _) in rollouts])
    g = np.stack([r.gamma for (_, r, _, _) in rollouts]); b = np.stack([r.b for (_, r, _, _) in rollouts])
    log("implicit_dynamics", model, mode, "...")
    t0 = time.time()
    out = s.implicit_dynamics(q, th, g, b)
    log("  done", round(time.time() - t0, 3), out["iters"].tolist(), out["status"].tolist())
    s.close()
for model, H, Hr in (("anydims", 8, 10), ("centroidal_wall", 5, 6)):
    d, prob, tabs, rollouts = make_case(model, 0, H_ref=Hr, H=H, B=2, seed=23, perturb=5e-3)
    obj = synth.make_objective(d, H, kind=model)
    for backend in (1, 0):
        s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-5, max_iter=4, kkt_backend=backend))
        log("newton_solve", model, "backend", backend, "...")
        t0 = time.time()
        u1, it, rn = s.newton_solve(np.stack([r[2] for r in rollouts]), np.stack([r[3] for r in rollouts]))
        log("  done", round(time.time() - t0, 3), it.tolist(), s.stats())
        s.close()
