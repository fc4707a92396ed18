import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from common import make_solver
from oracle import ip as oip, newton as onewton, synth
from real_problems import real_problem, real_rollout
from contactimplicitmpc.jl_amd import InteriorPointOptions, NewtonOptions
kappa, H, B, perturb = 2e-4, 40, 8, 0.05
d, P, prob, tabs = real_problem("quadruped", kappa)
rng = np.random.default_rng(4)
phases = [int(rng.integers(0, P.H)) for b in range(B)]
rollouts = [real_rollout(d, prob, H, phases[b], seed=10 + b, perturb=perturb) for b in range(B)]
obj = synth.make_objective(d, H, kind="quadruped")
s = make_solver(d, prob, rollouts, H, obj=obj, ip_opts=InteriorPointOptions(kappa_tol=kappa, r_tol=1e-8), newton_opts=NewtonOptions(kappa=kappa, r_tol=3e-4, max_iter=5))
u1, it, rn = s.newton_solve(np.stack([r[2] for r in rollouts]), np.stack([r[3] for r in rollouts]))
traj = s.trajectory(); cnt = s.rollout_counters()
for b, (window, ref, q0, q1) in enumerate(rollouts):
    core = onewton.Newton(d, H, obj, onewton.NewtonOptions(r_tol=3e-4, max_iter=5, solver="condensed"), oip.IPOptions(kappa_tol=kappa, r_tol=1e-8), kappa, ref)
    st = onewton.newton_solve(core, q0, q1, window, tabs, ref)
    print(b, "phase", phases[b], "it", it[b], st.iters, "sweeps", cnt["sweeps"][b], st.sweeps, "ip", cnt["ip_iters"][b], st.ip_iters, "fail", cnt["ip_failures"][b], st.ip_fail,
          "alphas", st.alphas, "rn", rn[b], st.r_norm / core.lay.N, "dq %.2e" % np.abs(traj["q"][b] - core.traj.q).max())
