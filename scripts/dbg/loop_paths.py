"""Per-step Newton iteration counts and residual norms of bench.py's warm-started B = 1 hopper loop: the device (whatever library
CIMPC_LIB names) or the oracle on the CPU.   python scripts/dbg/loop_paths.py device|oracle [steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
who = sys.argv[1] if len(sys.argv) > 1 else "device"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 43
from contactimplicitmpc.jl_amd import synthetic as synth  # noqa: E402
from contactimplicitmpc.jl_amd.trajectory import Dims  # noqa: E402

d = Dims(nq=4, nu=2, nw=2, nc=1, nb=2, mode=0)
H, H_ref = 20, 30
prob = synth.make_problem(d, H_ref, seed=1)
obj = synth.make_objective(d, H, kind="hopper")
window, ref, q0, q1 = synth.make_rollout(d, prob, H, phase=0, seed=7, perturb=0.01)
stride = np.zeros(d.nq)
stride[0] = prob["q_ref"][-2][0] - prob["q_ref"][0][0]
out = []
if who == "device":
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
    s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=1, mode=0, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]),
                    newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-4, max_iter=5))
    for t in range(H_ref):
        s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
    s.set_objective(obj.q, obj.u)
    s.set_gait(prob["q_ref"], prob["u_ref"], prob["th0"], stride, w=prob["w_ref"], gamma=prob["gamma_ref"], b=prob["b_ref"])
    a, b = q0[None].copy(), q1[None].copy()
    for k in range(steps):
        u1, it, rn = s.newton_solve(a, b, warm_start=k > 0)
        cnt = s.rollout_counters()
        nxt = s.trajectory(which=("q",))["q"][:, 2].copy()
        out.append((int(it[0]), float(rn[0]), int(cnt["ip_iters"][0]), int(cnt["sweeps"][0]), nxt[0].copy()))
        s.mpc_advance(stride)
        a, b = b, nxt
else:
    from oracle import ip as oip, lcp, mpc as ompc, newton as onewton
    tabs = [lcp.LinTable(d, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t]) for t in range(H_ref)]
    # the device's set_gait starts the window at the gait's first knot with the gait as the reference: the same state here
    core = onewton.Newton(d, H, obj, onewton.NewtonOptions(r_tol=3e-4, max_iter=5, solver="condensed"), oip.IPOptions(kappa_tol=prob["kappa"]), prob["kappa"], ref)
    rf, win = ref.copy(), np.array(window)
    a, b = q0.copy(), q1.copy()
    for k in range(steps):
        st = onewton.newton_solve(core, a, b, win, tabs, rf, warm_start=k > 0)
        nxt = core.traj.q[2].copy()
        out.append((st.iters, st.r_norm / core.lay.N, st.ip_iters, st.sweeps, nxt))
        ompc.rot_n_stride(d, rf, stride)
        win = ompc.update_window(win, H_ref)
        a, b = b, nxt
for k, (it, rn, ipi, sw, q) in enumerate(out):
    print("%s step %2d  newton %d  r_norm %.6e  ip_iters %5d sweeps %3d  q3 %s" % (who, k, it, rn, ipi, sw, np.array2string(q, precision=12)))
print(who, "mean newton iterations over steps 3..: %.3f" % np.mean([o[0] for o in out[3:]]))
