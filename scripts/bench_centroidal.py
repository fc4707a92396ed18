"""One-off measurement for BASELINE configs[4] dimensions (centroidal_quadruped: nq 18, nu 12, nw 3, nc 4, nb 16,
H = 60) in fp64, TrackingObjective (the velocity objective of the example needs the dense-LU fallback, N = 2880:
not a practical configuration yet - DESIGN.md).  Lock-step rounds, 32-lane interior-point groups, scalar KKT."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from contactimplicitmpc.jl_amd import synthetic as synth
from contactimplicitmpc.jl_amd.trajectory import Dims
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
d = Dims(nq=18, nu=12, nw=3, nc=4, nb=16)
H, H_ref = 60, 71
prob = synth.make_problem(d, H_ref, seed=1)
obj = synth.make_objective(d, H, dense_q=True)
for B in (1, 64, 512):
    ro = [synth.make_rollout(d, prob, H, phase=int(np.random.default_rng(g).integers(0, H_ref)), seed=100 + g, perturb=0.02) for g in range(B)]
    s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"], r_tol=1e-4),
                    newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-5, max_iter=5))
    for t in range(H_ref):
        s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
    s.set_objective(obj.q, obj.u)
    s.set_window(np.stack([w for (w, _, _, _) in ro]) + 1)
    s.set_reference(np.stack([r.q for (_, r, _, _) in ro]), np.stack([r.u for (_, r, _, _) in ro]), np.stack([r.w for (_, r, _, _) in ro]),
                    np.stack([r.gamma for (_, r, _, _) in ro]), np.stack([r.b for (_, r, _, _) in ro]), np.stack([r.theta for (_, r, _, _) in ro]))
    q0 = np.stack([r[2] for r in ro]); q1 = np.stack([r[3] for r in ro])
    s.newton_solve(q0, q1)
    s.profile_enable(True); s.profile_reset()
    n = 3 if B > 1 else 10
    t0 = time.perf_counter()
    for _ in range(n):
        u1, it, rn = s.newton_solve(q0, q1)
    dt = (time.perf_counter() - t0) / n
    st = s.stats(); pr = s.profile_read()
    print({k: round(v / n, 3) for k, v in pr.items() if k.endswith('_ms')}, {k: v // n for k, v in pr.items() if k.endswith('launches')})
    print("centroidal H=60 B=%d: %.2f ms per batch step, %.0f MPC steps/s, newton iters/step %.2f, ip iters/solve %.2f, rounds %d, ip failures %d"
          % (B, 1e3 * dt, B / dt, it.mean(), st["ip_iters"] / max(st["ip_solves"], 1), st["rounds"], st["ip_failures"]))
    s.close()

# ---- the same dimensions on the REAL problem: examples/centroidal_quadruped/reference/inplace_trot_v7.jld2 through the
# model restatement (lcp_models.CentroidalQuadruped), continuous_trot.jl settings (H_mpc = 50, kappa = 1e-3, IP r_tol 1e-4,
# Newton r_tol 3e-5 / max_iter 5), tracking part of its objective with body-x weighted like y, z (the example's velocity
# term, which makes its x-singular Q definite, needs the dense-LU backend)
from contactimplicitmpc.jl_amd import gait_io, lcp_models
m = lcp_models.CentroidalQuadruped()
P = lcp_models.reference_problem(m, gait_io.load_gait(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "gaits",
                                                                    "centroidal_inplace_trot_v7.jld2")), 1e-3)
H = 50
Q = np.tile(lcp_models.relative_state_cost([1.0, 1, 1], 3e-1 * np.ones(3), [0.2, 0.2, 1.0])[None], (H, 1, 1))
R = np.tile((3e-3 * np.eye(m.nu))[None], (H, 1, 1))
for B in (1, 64, 512):
    ro = [lcp_models.make_rollout(P, H, int(np.random.default_rng(g).integers(0, P.H)), seed=100 + g, perturb=0.02) for g in range(B)]
    s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=1e-3, r_tol=1e-4),
                    newton_opts=NewtonOptions(kappa=1e-3, r_tol=3e-5, max_iter=5))
    for t in range(P.H):
        s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
    s.set_objective(Q, R)
    s.set_window(np.stack([r["window"] for r in ro]) + 1)
    s.set_reference(*(np.stack([r[k] for r in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")))
    q0 = np.stack([r["q0"] for r in ro]); q1 = np.stack([r["q1"] for r in ro])
    s.newton_solve(q0, q1)
    n = 3 if B > 1 else 10
    t0 = time.perf_counter()
    for _ in range(n):
        u1, it, rn = s.newton_solve(q0, q1)
    dt = (time.perf_counter() - t0) / n
    st = s.stats()
    print("REAL centroidal inplace_trot_v7 H=50 B=%d: %.2f ms per batch step, %.0f MPC steps/s, newton iters/step %.2f, ip iters/solve %.2f, ip failures %d"
          % (B, 1e3 * dt, B / dt, it.mean(), st["ip_iters"] / max(st["ip_solves"], 1), st["ip_failures"]))
    s.close()

# ---- ... and with the example's OWN objective (continuous_trot.jl:43-47: TrackingVelocityObjective, Q singular along a
# common x shift): block-tridiagonal P -> banded LDL^T KKT backend (kkt_dense.hip: kkt_banded_kernel), lock-step rounds
Qx = np.tile(lcp_models.relative_state_cost([0.0, 1, 1], 3e-1 * np.ones(3), [0.2, 0.2, 1.0])[None], (H, 1, 1))
Vx = np.tile(np.diag(1e-3 * np.concatenate([np.ones(3), 1e3 * np.ones(3), np.ones(12)]))[None], (H, 1, 1))
for B in (1, 64, 512):
    ro = [lcp_models.make_rollout(P, H, int(np.random.default_rng(g).integers(0, P.H)), seed=100 + g, perturb=0.01) for g in range(B)]
    s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=1e-3, r_tol=1e-4),
                    newton_opts=NewtonOptions(kappa=1e-3, r_tol=3e-5, max_iter=5))
    for t in range(P.H):
        s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
    s.set_objective(Qx, R, V=Vx, v_target=np.zeros((H, m.nq)))
    s.set_window(np.stack([r["window"] for r in ro]) + 1)
    s.set_reference(*(np.stack([r[k] for r in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")))
    q0 = np.stack([r["q0"] for r in ro]); q1 = np.stack([r["q1"] for r in ro])
    s.newton_solve(q0, q1)
    s.profile_enable(True); s.profile_reset()
    n = 3 if B > 1 else 10
    t0 = time.perf_counter()
    for _ in range(n):
        u1, it, rn = s.newton_solve(q0, q1)
    dt = (time.perf_counter() - t0) / n
    st = s.stats(); pr = s.profile_read()
    print("REAL centroidal, example objective (velocity, banded LDL^T) H=50 B=%d: %.2f ms per batch step, %.0f MPC steps/s, newton iters/step %.2f, "
          "kkt %.2f ms in %d launches, ip failures %d" % (B, 1e3 * dt, B / dt, it.mean(), pr["kkt_ms"] / n, pr["kkt_launches"] // n, st["ip_failures"]))
    s.close()
