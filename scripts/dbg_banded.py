import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
g = np.load(os.path.join(ROOT, "tests/golden/hopper_velocity_newton.npz"), allow_pickle=True)
from oracle.dims import Dims, HOPPER_2D
d = Dims(**HOPPER_2D)
B, H, H_ref = int(g["B"]), int(g["H"]), int(g["H_ref"])
res = {}
for backend in (1, 0):
    s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=float(g["kappa"])),
                    newton_opts=NewtonOptions(kappa=float(g["kappa"]), r_tol=float(g["newton_r_tol"]), max_iter=int(g["newton_max_iter"]), kkt_backend=backend))
    for t in range(H_ref):
        s.set_linearization(t + 1, g["z0"][t], g["th0"][t], g["r0"][t], g["rz0"][t], g["rth0"][t])
    s.set_window(g["window"] + 1)
    s.set_reference(g["q_ref"], g["u_ref"], g["w_ref"], g["gamma_ref"], g["b_ref"], g["theta_ref"])
    opt = lambda k: g[k] if k in g.files else None
    s.set_objective(g["obj_q"], g["obj_u"], opt("obj_gamma"), opt("obj_b"), V=opt("obj_v"), q_target=opt("obj_q_target"), v_target=opt("obj_v_target"))
    out = s.implicit_dynamics(g["sweep_q"], g["sweep_theta"], g["sweep_gamma"], g["sweep_b"])
    r = np.random.default_rng(0).standard_normal((B, s.N if hasattr(s, "N") else H * (d.nr + d.nd)))
    dl = s.kkt_solve(r, 1e-5)
    u1, it, rn = s.newton_solve(g["q0"], g["q1"])
    res[backend] = (dl, it.copy(), rn.copy(), s.trajectory()["q"].copy())
    print("backend", backend, "iters", it, "rn", rn, "golden", g["newton_iters"], g["newton_rnorm"])
    s.close()
print("kkt delta rel diff", np.abs(res[0][0] - res[1][0]).max() / np.abs(res[1][0]).max(), "traj diff", np.abs(res[0][3] - res[1][3]).max())
print("obj_v diag", np.diag(g["obj_v"][0]) if "obj_v" in g.files else None, "obj_q diag", np.diag(g["obj_q"][0]))
