import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from common import make_case, make_solver
from oracle import synth, ip as oip, newton as onewton
from contactimplicitmpc.jl_amd import NewtonOptions
H, H_ref, B = 6, 8, 3
d, prob, tabs, rollouts = make_case("hopper", 0, H_ref=H_ref, H=H, B=B, seed=35, perturb=5e-3)
obj = synth.make_objective(d, H, kind="hopper", velocity=True)
obj.v = obj.v * 1e3
s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-5, max_iter=1))
qs, ths = [], []
outs = []
for b, (window, ref, q0, q1) in enumerate(rollouts):
    core = onewton.Newton(d, H, obj, onewton.NewtonOptions(r_tol=1e-5, max_iter=1, solver="lu"), oip.IPOptions(kappa_tol=prob["kappa"]), prob["kappa"], ref)
    st = onewton.newton_solve(core, q0, q1, window, tabs, ref)
    qs.append(core.traj.q.copy()); ths.append(core.traj.theta.copy())
    outs.append(oip.implicit_dynamics(d, tabs, window, core.traj.q, core.traj.theta, oip.IPOptions(kappa_tol=prob["kappa"])))
out = s.implicit_dynamics(np.stack(qs), np.stack(ths), want_z=True)
for b in range(B):
    o = outs[b]
    print("b", b, "iters gpu", out["iters"][b], "oracle", o["iters"], "status", out["status"][b], o["status"], "dd %.2e" % np.abs(out["d"][b] - o["d"]).max(),
          "dz %.2e" % np.abs(out["z"][b] - o["z"]).max())
    for i in range(H):
        if np.abs(out["d"][b][i] - o["d"][i]).max() > 1e-7:
            print("   i", i, "d gpu", out["d"][b][i], "oracle", o["d"][i]); print("   z gpu", out["z"][b][i]); print("   z ora", o["z"][i])
