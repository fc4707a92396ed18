"""GPU parity on the REAL reference problems: the reference's gait files through the model
restatements (contactimplicitmpc/jl_amd/lcp_models.py) into the C-ABI path, against the oracle on the same inputs."""
import numpy as np
import pytest

from common import make_solver, oracle_sweep
from oracle import ip as oip
from oracle import newton as onewton, synth
from real_problems import real_problem, real_rollout

pytestmark = pytest.mark.gpu


def test_reference_implicit_dynamics_test_on_device(gpu_required):
    """test/controller/implicit_dynamics.jl:7-24 through the device: every knot of quadruped gait2 (60 knots as
    two windows of 30 + wrap-around ones), |dq2|_inf < 1e-2, and agreement with the oracle."""
    d, P, prob, tabs = real_problem("quadruped", 1e-4)
    H = 30
    rollouts = [real_rollout(d, prob, H, phase, seed=0) for phase in (0, 30, 45, 59)]
    from contactimplicitmpc.jl_amd import InteriorPointOptions
    s = make_solver(d, prob, rollouts, H, ip_opts=InteriorPointOptions(kappa_tol=2e-4, r_tol=1e-8))
    ref = oracle_sweep(d, tabs, rollouts, oip.IPOptions(kappa_tol=2e-4, r_tol=1e-8))
    q = np.stack([tr.q for tr, _ in ref]); th = np.stack([tr.theta for tr, _ in ref])
    out = s.implicit_dynamics(q, th, want_z=True)
    assert out["status"].all()
    assert np.abs(out["d"]).max() < 1e-2
    for b, (tr, o) in enumerate(ref):
        same = out["iters"][b] == o["iters"]
        assert same.mean() >= 0.9
        np.testing.assert_allclose(out["d"][b][same], o["d"][same], rtol=0, atol=1e-8)
        np.testing.assert_allclose(out["z"][b][same], o["z"][same], rtol=0, atol=1e-7)
        for k in ("dq0", "dq1", "du1"):
            np.testing.assert_allclose(out[k][b][same], o[k][same], rtol=0, atol=1e-7 * max(1.0, np.abs(o[k]).max()))


@pytest.mark.parametrize("which,kappa,H,B,perturb,ip_rtol,n_rtol", [
    ("quadruped", 2e-4, 40, 8, 0.05, 1e-8, 3e-4),       # BASELINE configs[2/3] on the true problem (mpc_quadruped.jl:19-47)
    ("hopper", 2e-4, 20, 6, 0.05, 1e-8, 3e-4),          # BASELINE configs[1] on the REAL hopper problem: gait_forward.jld2 (a :joint_traj
                                                        # file, linearized at its own z / θ), objective and options of examples/hopper/flat.jl:30-47
    ("centroidal", 1e-3, 50, 4, 0.02, 1e-4, 3e-5),      # continuous_trot.jl:31-73 (H_mpc = 50; tracking part of the objective)
    ("centroidal_velocity", 1e-3, 50, 3, 0.01, 1e-4, 3e-5),   # ... with the example's OWN TrackingVelocityObjective (:43-47)
])
def test_newton_solve_on_real_problems(gpu_required, which, kappa, H, B, perturb, ip_rtol, n_rtol):
    from contactimplicitmpc.jl_amd import InteriorPointOptions, NewtonOptions, lcp_models
    velocity = which.endswith("_velocity")
    which = which.split("_")[0]
    d, P, prob, tabs = real_problem(which, kappa)
    H_ref = P.H
    rng = np.random.default_rng(4)
    rollouts = [real_rollout(d, prob, H, int(rng.integers(0, H_ref)), seed=10 + b, perturb=perturb) for b in range(B)]
    if which == "quadruped":
        obj = synth.make_objective(d, H, kind="quadruped")
    elif which == "hopper":
        obj = synth.make_objective(d, H, kind="hopper")      # examples/hopper/flat.jl:30-34
    elif velocity:
        # continuous_trot.jl:43-47: Q singular along a common x shift, made definite by the velocity term - the banded
        # LDL^T backend (block-tridiagonal P); the oracle solves the dense KKT system by LU
        obj = synth.make_objective(d, H, kind="quadruped", velocity=True)
        obj.q = np.tile(lcp_models.relative_state_cost([0.0, 1, 1], 3e-1 * np.ones(3), [0.2, 0.2, 1.0])[None], (H, 1, 1))
        obj.u = np.tile((3e-3 * np.eye(d.nu))[None], (H, 1, 1))
        obj.v = np.tile(np.diag(1e-3 * np.concatenate([np.ones(3), 1e3 * np.ones(3), np.ones(12)]))[None], (H, 1, 1))
        obj.v_target = np.zeros((H, d.nq)); obj.q_target = None
        obj.__post_init__()
    else:
        obj = synth.make_objective(d, H, kind="quadruped")
        # the example's body-x weight is 0 (Q singular along a common x shift; its velocity term restores definiteness,
        # continuous_trot.jl:43-47) - the tracking-only objective of this test weights body x like y and z
        obj.q = np.tile(lcp_models.relative_state_cost([1.0, 1, 1], 3e-1 * np.ones(3), [0.2, 0.2, 1.0])[None], (H, 1, 1))
        obj.u = np.tile((3e-3 * np.eye(d.nu))[None], (H, 1, 1))
        obj.__post_init__()
    s = make_solver(d, prob, rollouts, H, obj=obj, ip_opts=InteriorPointOptions(kappa_tol=kappa, r_tol=ip_rtol),
                    newton_opts=NewtonOptions(kappa=kappa, r_tol=n_rtol, max_iter=5))
    u1, it, rn = s.newton_solve(np.stack([r[2] for r in rollouts]), np.stack([r[3] for r in rollouts]))
    traj = s.trajectory()
    cnt = s.rollout_counters()
    # On the real problems r_tol = 1e-8 of the interior point sits AT the round-off floor of the residual (mass-matrix
    # entries / h^2 ~ 1e4): single-iteration flips between two correct implementations are frequent (the numpy and
    # the C oracle disagree on 2 of these 8 rollouts, the numpy oracle with itself across two CPUs), each moves d by
    # ~1e-6 and the short Newton run (stops at r_tol = 3e-4, weakly regularised duals) amplifies that to 1e-4..1e-3
    # in q.  Required: identical Newton iteration counts, at least half of the rollouts on the oracle's exact path
    # (1e-6), all of them within the amplification band.
    tight = 0
    for b, (window, ref, q0, q1) in enumerate(rollouts):
        core = onewton.Newton(d, H, obj, onewton.NewtonOptions(r_tol=n_rtol, max_iter=5, solver="lu" if velocity else "condensed"),
                              oip.IPOptions(kappa_tol=kappa, r_tol=ip_rtol), kappa, ref)
        st = onewton.newton_solve(core, q0, q1, window, tabs, ref)
        assert it[b] == st.iters, (b, it[b], st.iters)
        assert abs(int(cnt["ip_iters"][b]) - st.ip_iters) <= 0.01 * st.ip_iters and cnt["sweeps"][b] == st.sweeps
        dq = np.abs(traj["q"][b] - core.traj.q).max()
        du = np.abs(traj["u"][b] - core.traj.u).max() / max(1.0, np.abs(core.traj.u).max())
        assert dq < 1e-2 and du < 5e-2, (b, dq, du)
        # residual norm at exit (1.5e-4: mostly the d entries, which a flipped interior-point iteration moves by ~1e-6 each): 2 % where
        # the trajectories agree to 1e-6, inside the amplification band elsewhere
        np.testing.assert_allclose(rn[b], st.r_norm / core.lay.N, rtol=2e-2 if dq < 1e-6 else 0.15, atol=1e-12)
        if dq < 1e-6:
            tight += 1
            np.testing.assert_allclose(traj["nu"][b], core.nu, rtol=0, atol=1e-4 * max(1.0, np.abs(core.nu).max()))
    assert tight >= B // 2, (tight, B)
    assert it.max() >= 1
