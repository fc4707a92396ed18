"""Is a cold-start headline solve reproducible bit for bit from call to call (same handle) and handle to handle?  python scripts/dbg/repro_check.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
import bench
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
H, H_ref, B = 40, 60, 512
d, prob, obj, ro = bench.build_inputs(B, H, H_ref, seed=1234, perturb=0.05)
q0 = np.stack([r[2] for r in ro]); q1 = np.stack([r[3] for r in ro])
outs = []
for handle in range(2):
    s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]),
                    newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-4, max_iter=5))
    for t in range(H_ref):
        s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
    s.set_objective(obj.q, obj.u)
    s.set_window(np.stack([w for (w, _, _, _) in ro]) + 1)
    s.set_reference(*[np.stack([getattr(r, k) for (_, r, _, _) in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")])
    for rep in range(4):
        u1, it, rn = s.newton_solve(q0, q1)
        outs.append((u1.copy(), it.copy(), rn.copy(), s.stats()["rounds"]))
    s.close()
ref = outs[0]
for k, o in enumerate(outs):
    diff = np.flatnonzero((o[0] != ref[0]).any(axis=1))
    print("solve", k, "rounds", o[3], "converged", int((o[2] < 3e-4).sum()), "rollouts differing from solve 0:", len(diff), diff[:8].tolist(),
          "newton iters differ:", int((o[1] != ref[1]).sum()))
