"""Per-rollout chain statistics of one batched newton_solve (which rollouts set the number of rounds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
np.seterr(all="ignore")
import bench
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
H, H_ref, B = 40, 60, int(os.environ.get("B", "512"))
d, prob, obj, ro = bench.build_inputs(B, H, H_ref, seed=1234, perturb=0.05)
s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]),
                newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-4, max_iter=5))
for t in range(H_ref):
    s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
s.set_objective(obj.q, obj.u)
s.set_window(np.stack([w for (w, _, _, _) in ro]) + 1)
s.set_reference(np.stack([r.q for (_, r, _, _) in ro]), np.stack([r.u for (_, r, _, _) in ro]), np.stack([r.w for (_, r, _, _) in ro]),
                np.stack([r.gamma for (_, r, _, _) in ro]), np.stack([r.b for (_, r, _, _) in ro]), np.stack([r.theta for (_, r, _, _) in ro]))
u1, it, rn = s.newton_solve(np.stack([r[2] for r in ro]), np.stack([r[3] for r in ro]))
c = s.rollout_counters()
print("stats", s.stats())
print("newton iters hist", np.bincount(it))
print("sweeps/rollout hist", np.bincount(c["sweeps"]))
print("ip fails/rollout: total", c["ip_failures"].sum(), "rollouts with fails", (c["ip_failures"] > 0).sum(), "max", c["ip_failures"].max())
order = np.argsort(-c["sweeps"])[:10]
for b in order:
    print("rollout", b, "newton", it[b], "sweeps", c["sweeps"][b], "ip_iters", c["ip_iters"][b], "fails", c["ip_failures"][b], "rnorm %.2e" % rn[b])
