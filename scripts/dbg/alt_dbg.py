import sys, os; sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import numpy as np
from oracle import ip as oip
from oracle import newton as onewton, synth
from common import make_case, make_solver
from contactimplicitmpc.jl_amd import NewtonOptions
B,H,H_ref=4,8,12
d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=H_ref, H=H, B=B, seed=17, perturb=5e-3)
obj = synth.make_objective(d, H, kind="quadruped")
alt = np.random.default_rng(5).uniform(0.0, 0.02, (B, d.nc))
s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-5, max_iter=4))
s.set_altitude(alt)
u1, it, rn = s.newton_solve(np.stack([r[2] for r in rollouts]), np.stack([r[3] for r in rollouts]))
traj = s.trajectory(); cnt = s.rollout_counters()
for b,(window,ref,q0,q1) in enumerate(rollouts):
    for t in tabs: t.alt=alt[b].copy()
    core = onewton.Newton(d, H, obj, onewton.NewtonOptions(r_tol=1e-5, max_iter=4, solver="lu"), oip.IPOptions(kappa_tol=prob["kappa"]), prob["kappa"], ref)
    st = onewton.newton_solve(core, q0, q1, window, tabs, ref)
    print(b, 'dev', int(it[b]), int(cnt["sweeps"][b]), int(cnt["ip_iters"][b]), 'rn %.3e'%rn[b], '| oracle', st.iters, st.sweeps, st.ip_iters, 'rn %.3e'%st.r_norm if hasattr(st,'r_norm') else '', '| du1 %.2e dq %.2e'%(np.abs(u1[b]-core.traj.u[0]).max(), np.abs(traj["q"][b]-core.traj.q).max()))
