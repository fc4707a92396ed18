"""Times the banded LDL^T KKT backend on the real centroidal problem (H = 50, the example objective)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions, gait_io, lcp_models
m = lcp_models.CentroidalQuadruped()
P = lcp_models.reference_problem(m, gait_io.load_gait(os.path.join(ROOT, "tests", "golden", "gaits", "centroidal_inplace_trot_v7.jld2")), 1e-3)
H = 50
R = np.tile((3e-3 * np.eye(m.nu))[None], (H, 1, 1))
Qx = np.tile(lcp_models.relative_state_cost([0.0, 1, 1], 3e-1 * np.ones(3), [0.2, 0.2, 1.0])[None], (H, 1, 1))
Vx = np.tile(np.diag(1e-3 * np.concatenate([np.ones(3), 1e3 * np.ones(3), np.ones(12)]))[None], (H, 1, 1))
B = 1
ro = [lcp_models.make_rollout(P, H, 5, seed=100, perturb=0.01)]
s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=1e-3, r_tol=1e-4), newton_opts=NewtonOptions(kappa=1e-3, r_tol=3e-5, max_iter=5))
for t in range(P.H):
    s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
s.set_objective(Qx, R, V=Vx, v_target=np.zeros((H, m.nq)))
s.set_window(np.stack([r["window"] for r in ro]) + 1)
s.set_reference(*(np.stack([r[k] for r in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")))
s.implicit_dynamics(np.stack([r["q"] for r in ro]), np.stack([r["theta"] for r in ro]))
r = np.random.default_rng(0).standard_normal((B, H * 48))
s.kkt_solve(r, 1e-5)
s.profile_enable(True); s.profile_reset()
t0 = time.perf_counter()
for _ in range(5):
    s.kkt_solve(r, 1e-5)
dt = (time.perf_counter() - t0) / 5
print(os.environ.get("CIMPC_LIB", "default"), "kkt_solve wall %.2f ms, kernel %.2f ms" % (1e3 * dt, s.profile_read()["kkt_ms"] / 5))
