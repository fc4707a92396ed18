import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
np.seterr(all="ignore")
from oracle import ip as oip, newton as onewton, synth
from common import make_case, make_solver
from contactimplicitmpc.jl_amd import NewtonOptions
H, H_ref, B = 10, 16, 6
d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=H_ref, H=H, B=B, seed=11, perturb=3e-2)
obj = synth.make_objective(d, H)
for max_iter in (0, 1, 2, 4):
    s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-6, max_iter=max_iter))
    u1, it, rn = s.newton_solve(np.stack([r[2] for r in rollouts]), np.stack([r[3] for r in rollouts]))
    traj = s.trajectory(); stt = s.stats()
    print("max_iter", max_iter, "gpu stats", stt)
    for b, (window, ref, q0, q1) in enumerate(rollouts):
        core = onewton.Newton(d, H, obj, onewton.NewtonOptions(r_tol=1e-6, max_iter=max_iter, solver="lu"),
                              oip.IPOptions(kappa_tol=prob["kappa"]), prob["kappa"], ref)
        st = onewton.newton_solve(core, q0, q1, window, tabs, ref)
        print("  b", b, "it", it[b], st.iters, "sweeps", st.sweeps, "ipit", st.ip_iters, "alphas", st.alphas,
              "rn %.6e %.6e" % (rn[b], st.r_norm / core.lay.N),
              "dq %.2e du %.2e dnu %.2e" % (np.abs(traj["q"][b] - core.traj.q).max(), np.abs(traj["u"][b] - core.traj.u).max(),
                                            np.abs(traj["nu"][b] - core.nu).max()))
