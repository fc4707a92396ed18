"""Device sensitivities against the oracle, tight: max abs error of dq0 / dq1 / du1 over the converged solves of a small batch,
and the residual norm after ONE Newton iteration (device vs oracle).  python scripts/dbg/sens_err.py [model ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ip as oip, newton as onewton, synth  # noqa: E402
from common import make_case, make_solver, oracle_sweep  # noqa: E402
from contactimplicitmpc.jl_amd import NewtonOptions  # noqa: E402

for model in (sys.argv[1:] or ["hopper3d", "quadruped", "centroidal"]):
    d, prob, tabs, rollouts = make_case(model, 0, H_ref=8, H=6, B=4, seed=3)
    opts = oip.IPOptions(kappa_tol=prob["kappa"])
    s = make_solver(d, prob, rollouts, 6)
    ref = oracle_sweep(d, tabs, rollouts, opts)
    q = np.stack([tr.q for tr, _ in ref]); th = np.stack([tr.theta for tr, _ in ref])
    g = np.stack([tr.gamma for tr, _ in ref]); bb = np.stack([tr.b for tr, _ in ref])
    out = s.implicit_dynamics(q, th, g, bb, want_z=True)
    for k in ("dq0", "dq1", "du1"):
        e = 0.0; sc = 0.0; worst = None
        for b, (tr, o) in enumerate(ref):
            ok = (out["status"][b] == o["status"]) & (out["iters"][b] == o["iters"]) & (o["status"] == 1)
            err = np.abs(out[k][b][ok] - o[k][ok])
            if err.size and err.max() > e:
                e = err.max(); worst = (b, np.unravel_index(err.argmax(), err.shape))
            sc = max(sc, np.abs(o[k][ok]).max())
        print(model, k, "max abs err %.3e  scale %.3e" % (e, sc), worst, flush=True)
    H, H_ref = (10, 12)
    d, prob, tabs, rollouts = make_case(model, 0, H_ref=H_ref, H=H, B=4, seed=23, perturb=5e-3)
    obj = synth.make_objective(d, H, kind=model, dense_q=(model == "centroidal"))
    s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-5, max_iter=1))
    u1, it, rn = s.newton_solve(np.stack([r[2] for r in rollouts]), np.stack([r[3] for r in rollouts]))
    cnt = s.rollout_counters()
    for b, (window, rf, q0, q1) in enumerate(rollouts):
        core = onewton.Newton(d, H, obj, onewton.NewtonOptions(r_tol=1e-5, max_iter=1, solver="lu"), oip.IPOptions(kappa_tol=prob["kappa"]), prob["kappa"], rf)
        st = onewton.newton_solve(core, q0, q1, window, tabs, rf)
        print(model, "rollout", b, "r_norm after one iteration: device %.9e oracle %.9e  rel %.2e   ip_iters %d / %d  sweeps %d / %d" % (rn[b], st.r_norm / core.lay.N, rn[b] / (st.r_norm / core.lay.N) - 1, cnt["ip_iters"][b], st.ip_iters, cnt["sweeps"][b], st.sweeps), flush=True)
