"""Per (step, rollout) numbers of tests/test_gpu_round4.py::test_relinearisation_between_warm_started_solves (diagnostics)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import ip as oip, lcp, newton as onewton, synth
from common import make_case, make_solver
from test_gpu_round4 import _perturbed_knot
from contactimplicitmpc.jl_amd import InteriorPointOptions, NewtonOptions

relin = len(sys.argv) < 2 or sys.argv[1] != "norelin"
H, H_ref, B = 8, 12, 4
d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=H_ref, H=H, B=B, seed=31, perturb=5e-3)
obj = synth.make_objective(d, H)
s = make_solver(d, prob, rollouts, H, obj=obj, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]),
                newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-5, max_iter=4))
mk = lambda sv: [onewton.Newton(d, H, obj, onewton.NewtonOptions(r_tol=1e-5, max_iter=4, solver=sv), oip.IPOptions(kappa_tol=prob["kappa"]), prob["kappa"], ref) for (_, ref, _, _) in rollouts]
c1, c2 = mk("lu"), mk("condensed")
tabs = list(tabs); rng = np.random.default_rng(7)
q0 = np.stack([r[2] for r in rollouts]); q1 = np.stack([r[3] for r in rollouts])
for step, knots in enumerate(([], [1, 4, 5, 10], [0, 4, 7])):
    for t in (knots if relin else []):
        new = _perturbed_knot(prob, t, rng)
        s.set_linearization(t + 1, *new)
        tabs[t] = lcp.LinTable(d, *new)
    u1, it, rn = s.newton_solve(q0, q1, warm_start=step > 0)
    tr = s.trajectory(); cnt = s.rollout_counters()
    for b, (window, ref, a, b_) in enumerate(rollouts):
        st = onewton.newton_solve(c1[b], a, b_, window, tabs, ref, warm_start=step > 0)
        st2 = onewton.newton_solve(c2[b], a, b_, window, tabs, ref, warm_start=step > 0)
        print("step", step, "b", b, "dev (it, sweeps, ip_iters, fail)", int(it[b]), int(cnt["sweeps"][b]), int(cnt["ip_iters"][b]), int(cnt["ip_failures"][b]),
              "| oracle lu", st.iters, st.sweeps, st.ip_iters, st.ip_fail, "| condensed", st2.iters, st2.sweeps, st2.ip_iters, st2.ip_fail,
              "| du %.2e dq %.2e | lu-vs-condensed du %.2e dq %.2e | r_norm dev %.3e oracle %.3e" % (
                  np.abs(u1[b] - c1[b].traj.u[0]).max(), np.abs(tr["q"][b] - c1[b].traj.q).max(),
                  np.abs(c2[b].traj.u[0] - c1[b].traj.u[0]).max(), np.abs(c2[b].traj.q - c1[b].traj.q).max(), rn[b], st.r_norm / c1[b].lay.N))
