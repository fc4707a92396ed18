import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from common import make_case, make_solver
from oracle import synth, ip as oip, newton as onewton
from contactimplicitmpc.jl_amd import NewtonOptions
H, H_ref, B = 6, 8, 3
d, prob, tabs, rollouts = make_case("hopper", 0, H_ref=H_ref, H=H, B=B, seed=35, perturb=5e-3)
for vtscale in (0.0, 0.01):
    obj = synth.make_objective(d, H, kind="hopper", velocity=True)
    obj.v = obj.v * 1e3; obj.v_target = vtscale * np.random.default_rng(35).standard_normal((H, d.nq)); obj.q_target = None; obj.__post_init__()
    for mi in (0, 1, 5):
        s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-5, max_iter=mi))
        u1, it, rn = s.newton_solve(np.stack([r[2] for r in rollouts]), np.stack([r[3] for r in rollouts]))
        tr = s.trajectory()
        for b, (window, ref, q0, q1) in enumerate(rollouts):
            core = onewton.Newton(d, H, obj, onewton.NewtonOptions(r_tol=1e-5, max_iter=mi, solver="lu"), oip.IPOptions(kappa_tol=prob["kappa"]), prob["kappa"], ref)
            st = onewton.newton_solve(core, q0, q1, window, tabs, ref)
            print("vt", vtscale, "max_iter", mi, "b", b, "iters", it[b], st.iters, "rn %.6e %.6e" % (rn[b], st.r_norm / core.lay.N), "dq %.2e" % np.abs(tr["q"][b] - core.traj.q).max())
