"""Diagnostic: who is in the tail of the headline batch?  Per-rollout Newton iterations / reference-equivalent evaluations /
interior-point failures of the bench workload (B = 512), and the interior-point iteration histogram of one cold sweep."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
from contactimplicitmpc.jl_amd.trajectory import Dims

H, H_ref, B = 40, 60, 512
d = Dims(**bench.QUADRUPED)
_, prob, obj = bench.build_problem(H, H_ref)
ro = bench.build_rollouts(d, prob, B, H, H_ref, 1234, 0.05, first=0)
s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]),
                newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-4, max_iter=5))
for t in range(H_ref):
    s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
s.set_objective(obj.q, obj.u)
s.set_window(np.stack([w for (w, _, _, _) in ro]) + 1)
s.set_reference(*[np.stack([getattr(r, k) for (_, r, _, _) in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")])
u1, it, rn = s.newton_solve(np.stack([r[2] for r in ro]), np.stack([r[3] for r in ro]))
c = s.rollout_counters()
print("stats", s.stats())
print("newton iters histogram", np.bincount(it, minlength=6).tolist())
print("reference-equivalent evaluations per rollout: mean %.2f max %d" % (c["sweeps"].mean(), c["sweeps"].max()))
print("ip failures per rollout: rollouts with any %d, total %d, max %d" % ((c["ip_failures"] > 0).sum(), c["ip_failures"].sum(), c["ip_failures"].max()))
slow = np.argsort(-c["sweeps"])[:20]
print("20 heaviest rollouts: sweeps", c["sweeps"][slow].tolist(), "newton", it[slow].tolist(), "ip_fail", c["ip_failures"][slow].tolist())
log = s.newton_log()
print("line-search iterate of accepted steps (all rollouts):", np.bincount(log[..., 3][log[..., 0] > 0].astype(int), minlength=8).tolist())
