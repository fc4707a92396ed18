"""Closed loop: the reference's own end-to-end test, test/controller/mpc_quadruped.jl:1-64, re-stated.

  quadruped gait2 (update_friction_coefficient!), H_mpc = 10, N_sample = 5, κ_mpc = 2e-4, TrackingObjective :23-27,
  NewtonOptions(r_tol 3e-4, max_iter 5), IP (undercut 5, γ_reg 0.1, κ_tol = κ_mpc, r_tol 1e-8); simulate 1000 plant
  steps at h/5 and compare `tracking_error` with the RECORDED nominal values q 0.0201, u 0.0437, γ 0.374, b 0.0789
  (the reference asserts < 1.5 x nominal).

The plant (RoboDojo's simulate!, external) is oracle/plant.py: the nonlinear time-stepping complementarity problem
solved to 1e-8 per step.  Measured with the oracle as the controller, H_sim = 1000 (scripts/closed_loop.py):
q 0.02048, u 0.04331, γ 0.3768, b 0.07917 - within 2 %, 1 %, 0.7 %, 0.3 % of the reference's recorded values.
The suite runs 300 plant steps (one and a half gait cycles): CPU with the oracle, GPU with the product policy
(contactimplicitmpc/jl_amd/policy.py over the C ABI).
"""
import numpy as np
import pytest

from oracle import ip as oip
from oracle import newton as onewton, plant as pl, synth
from real_problems import real_problem

NOMINAL = (0.0201, 0.0437, 0.374, 0.0789)         # test/controller/mpc_quadruped.jl:59-62
H_MPC, N_SAMPLE, KAPPA = 10, 5, 2e-4


def _oracle_loop(H_sim):
    d, P, prob, tabs = real_problem("quadruped", KAPPA, True)
    obj = synth.make_objective(d, H_MPC, kind="quadruped")
    ref = onewton.Traj(q=P.q.copy(), u=P.u.copy(), w=P.w.copy(), gamma=P.gamma.copy(), b=P.b.copy(), theta=P.theta.copy())
    pol = pl.OraclePolicy(d, tabs, ref, prob["stride"], obj, H_MPC, N_SAMPLE, KAPPA,
                          onewton.NewtonOptions(r_tol=3e-4, max_iter=5, solver="lu"), oip.IPOptions(kappa_tol=KAPPA, r_tol=1e-8))
    q1, v1 = P.q[1].copy(), (P.q[1] - P.q[0]) / P.h                     # initial_conditions, trajectory.jl:222-227
    ok, q, u, g, b = pl.simulate(pl.QuadrupedPlant(), pol, q1, v1, H_sim, P.h / N_SAMPLE)
    return P, pol, ok, q, u, g, b


def test_numpy_plant_equals_the_torch_model():
    """Two independent derivations of the quadruped dynamics: closed-form sums over the bodies (oracle/plant.py) and
    the autodiff Lagrangian (lcp_models.py)."""
    from contactimplicitmpc.jl_amd import lcp_models
    P, m = pl.QuadrupedPlant(), lcp_models.Quadruped()
    rng = np.random.default_rng(0)
    for _ in range(3):
        z, th = rng.uniform(0.1, 1.0, 43), rng.uniform(0.1, 1.0, 34)
        r_t, rz_t, _ = m.linearize(z, th, 1e-3)
        np.testing.assert_allclose(P.residual(z, th, 1e-3), r_t, rtol=0, atol=1e-12)
        np.testing.assert_allclose(P.jacobian_z(z, th), rz_t, rtol=0, atol=1e-11 * np.abs(rz_t).max())


def test_plant_step_reproduces_the_gait():
    """Plant step from two consecutive gait configurations with the gait's control: lands on the next one (the gait
    file satisfies the dynamics to 1e-5; contact forces differ by the κ the file was optimised at)."""
    d, P, prob, tabs = real_problem("quadruped", KAPPA, True)
    plant = pl.QuadrupedPlant()
    for t in (0, 17, 40):
        status, it, q2, gam, b = pl.plant_step(plant, P.q[t], P.q[t + 1], P.u[t], np.zeros(2), plant.mu_world, P.h, pl.SIM_OPTS)
        assert status and it <= 30
        assert np.abs(q2 - P.q[t + 2]).max() < 5e-3


def test_closed_loop_tracking_error_with_the_oracle_controller():
    P, pol, ok, q, u, g, b = _oracle_loop(300)
    assert ok                                                            # @test status
    qe, ue, ge, be = pl.tracking_error(P.q, P.u, P.gamma, P.b, q, u, g, b, N_SAMPLE)
    assert qe < NOMINAL[0] * 1.5 and ue < NOMINAL[1] * 1.5 and ge < NOMINAL[2] * 1.5 and be < NOMINAL[3] * 1.5   # :59-62
    # and close to the recorded nominal values themselves (u, γ, b are stationary; q drifts, so shorter runs sit lower)
    assert abs(ue / NOMINAL[1] - 1) < 0.05 and abs(ge / NOMINAL[2] - 1) < 0.05 and abs(be / NOMINAL[3] - 1) < 0.05
    assert 0.5 * NOMINAL[0] < qe
    assert 1.0 <= np.mean(pol.iters) <= 5.0


@pytest.mark.gpu
def test_gait_advance_equals_rot_n_stride_on_the_full_trajectory(gpu_required):
    """cimpc_set_gait + cimpc_mpc_advance against rot_n_stride! / update_window! applied to the FULL reference
    trajectory (policy.jl:136-139), H_mpc = 10 < H_ref = 60, rollouts starting at different phases, two laps."""
    from contactimplicitmpc.jl_amd import CIMPCSolver
    from oracle import mpc as ompc
    d, P, prob, tabs = real_problem("quadruped", KAPPA, True)
    B, H = 3, H_MPC
    phase = np.array([0, 7, 55], dtype=np.int32)
    s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, P.H, H, B=B, mode=0)
    s.set_gait(P.q, P.u, P.theta, prob["stride"], w=P.w, gamma=P.gamma, b=P.b, phase=phase)
    refs = []
    for b in range(B):
        tr = onewton.Traj(q=P.q.copy(), u=P.u.copy(), w=P.w.copy(), gamma=P.gamma.copy(), b=P.b.copy(), theta=P.theta.copy())
        win = np.arange(H + 2)
        for _ in range(int(phase[b])):
            ompc.rot_n_stride(d, tr, prob["stride"]); win = ompc.update_window(win, P.H)
        refs.append((tr, win))
    for step in range(130):
        got = s.reference()
        for b, (tr, win) in enumerate(refs):
            np.testing.assert_array_equal(got["window"][b] - 1, win)          # the ABI speaks 1-based knots
            np.testing.assert_allclose(got["q"][b], tr.q[:H + 2], rtol=0, atol=1e-12)
            np.testing.assert_allclose(got["u"][b], tr.u[:H], rtol=0, atol=0)
            np.testing.assert_allclose(got["gamma"][b], tr.gamma[:H], rtol=0, atol=0)
            np.testing.assert_allclose(got["b"][b], tr.b[:H], rtol=0, atol=0)
            np.testing.assert_allclose(got["theta"][b], tr.theta[:H], rtol=0, atol=1e-12)
        s.mpc_advance(prob["stride"])
        refs = [(ompc.rot_n_stride(d, tr, prob["stride"]) or tr, ompc.update_window(win, P.H)) for tr, win in refs]
    s.close()


@pytest.mark.gpu
def test_closed_loop_tracking_error_with_the_device_policy(gpu_required):
    """The same loop with the PRODUCT in it: CIMPCPolicy (device solver + device-side trajectory rotation)."""
    from contactimplicitmpc.jl_amd import NewtonOptions, InteriorPointOptions
    from contactimplicitmpc.jl_amd.policy import CIMPCPolicy
    d, P, prob, tabs = real_problem("quadruped", KAPPA, True)
    obj = synth.make_objective(d, H_MPC, kind="quadruped")
    pol = CIMPCPolicy(P, obj.q, obj.u, H_mpc=H_MPC, N_sample=N_SAMPLE, B=1,
                      n_opts=NewtonOptions(kappa=KAPPA, r_tol=3e-4, max_iter=5), ip_opts=InteriorPointOptions(kappa_tol=KAPPA, r_tol=1e-8))
    H_sim = 300
    q1, v1 = P.q[1].copy(), (P.q[1] - P.q[0]) / P.h
    ok, q, u, g, b = pl.simulate(pl.QuadrupedPlant(), lambda qq, t: pol(qq[t + 1][None])[0], q1, v1, H_sim, P.h / N_SAMPLE)
    pol.close()
    assert ok
    qe, ue, ge, be = pl.tracking_error(P.q, P.u, P.gamma, P.b, q, u, g, b, N_SAMPLE)
    assert qe < NOMINAL[0] * 1.5 and ue < NOMINAL[1] * 1.5 and ge < NOMINAL[2] * 1.5 and be < NOMINAL[3] * 1.5
    assert abs(ue / NOMINAL[1] - 1) < 0.05 and abs(ge / NOMINAL[2] - 1) < 0.05 and abs(be / NOMINAL[3] - 1) < 0.05
    # against the oracle-controlled loop: same plant, same controller algorithm
    Po, polo, oko, qo, uo, go, bo = _oracle_loop(H_sim)
    np.testing.assert_allclose(q[:60], qo[:60], rtol=0, atol=1e-5)       # before round-off has time to grow
    eo = pl.tracking_error(P.q, P.u, P.gamma, P.b, qo, uo, go, bo, N_SAMPLE)
    for a, c in zip((qe, ue, ge, be), eo):
        assert abs(a / c - 1) < 0.03


# ---- second closed-loop test of the reference: test/controller/mpc_flamingo.jl:1-80 --------------------------------------
#   flamingo gait_forward_36_4, H_mpc = 15, N_sample = 5, κ_mpc = 2e-4, TrackingVelocityObjective (:23-28), the policy's
#   DEFAULT mode :configurationforce (policy.jl:46), simulator options :53-60; recorded nominal tracking errors
#   q 0.0154, u 0.0829, γ 0.444, b 0.0169.  Oracle-controlled, 1000 plant steps (98 s of CPU): q 0.01268, u 0.08252,
#   γ 0.4406, b 0.01591.
NOMINAL_F = (0.0154, 0.0829, 0.444, 0.0169)
SIM_OPTS_F = oip.IPOptions(r_tol=1e-8, kappa_tol=1e-8, undercut=np.inf, gamma_reg=0.0, eps_min=0.05, max_iter=100, max_ls=25)


def _flamingo_objective(H):
    t = lambda a: np.tile(np.diag(np.asarray(a, dtype=float))[None], (H, 1, 1))
    return onewton.Objective(q=t(1e-1 * np.array([3e2, 1e-6, 3e2, 1, 1, 1, 1, 0.1, 0.1])), u=t(3e-1 * np.array([0.1, 0.1, 0.3, 0.3, 2, 2])),
                             gamma=t(1e-100 * np.ones(4)), b=t(1e-100 * np.ones(8)), v=t(1e-3 * np.array([1, 1, 1e4, 1, 1, 1, 1, 1e4, 1e4])))


def _flamingo_oracle_loop(H_sim, H_mpc=15):
    d, P, prob, tabs = real_problem("flamingo", KAPPA, False, 1)
    ref = onewton.Traj(q=P.q.copy(), u=P.u.copy(), w=P.w.copy(), gamma=P.gamma.copy(), b=P.b.copy(), theta=P.theta.copy())
    pol = pl.OraclePolicy(d, tabs, ref, prob["stride"], _flamingo_objective(H_mpc), H_mpc, N_SAMPLE, KAPPA,
                          onewton.NewtonOptions(r_tol=3e-4, max_iter=5, solver="lu"), oip.IPOptions(kappa_tol=KAPPA, r_tol=1e-8))
    q1, v1 = P.q[1].copy(), (P.q[1] - P.q[0]) / P.h
    ok, q, u, g, b = pl.simulate(pl.FlamingoPlant(), pol, q1, v1, H_sim, P.h / N_SAMPLE, opts=SIM_OPTS_F)
    return P, pol, ok, q, u, g, b


def test_flamingo_plant_equals_the_torch_model_and_gait_is_periodic():
    from contactimplicitmpc.jl_amd import lcp_models
    P, m = pl.FlamingoPlant(), lcp_models.Flamingo()
    rng = np.random.default_rng(1)
    z, th = rng.uniform(0.1, 1.0, m.nz), rng.uniform(0.1, 1.0, m.nth)
    r_t, rz_t, _ = m.linearize(z, th, 1e-3)
    np.testing.assert_allclose(P.residual(z, th, 1e-3), r_t, rtol=0, atol=1e-12)
    np.testing.assert_allclose(P.jacobian_z(z, th), rz_t, rtol=0, atol=1e-11 * np.abs(rz_t).max())
    d, Pr, prob, tabs = real_problem("flamingo", KAPPA, False, 1)
    assert np.abs(Pr.r0[:, :m.nq]).max() < 2e-6                       # the shipped gait satisfies the restated dynamics
    assert np.abs(Pr.q[0][1:] - Pr.q[-2][1:]).max() < 1e-8 and np.abs(Pr.q[1][1:] - Pr.q[-1][1:]).max() < 1e-8   # mpc_flamingo.jl:67-68


def test_flamingo_closed_loop_with_the_oracle_controller():
    P, pol, ok, q, u, g, b = _flamingo_oracle_loop(150)
    assert ok
    assert np.abs(q[0] - (P.q[1] * (1 - 1 / N_SAMPLE) + P.q[0] / N_SAMPLE)).max() < 1e-8 and np.abs(q[1] - P.q[1]).max() < 1e-8   # :65-66
    e = pl.tracking_error(P.q, P.u, P.gamma, P.b, q, u, g, b, N_SAMPLE)
    assert all(a < 1.5 * n for a, n in zip(e, NOMINAL_F))               # mpc_flamingo.jl:71-74
    assert abs(e[1] / NOMINAL_F[1] - 1) < 0.10 and abs(e[2] / NOMINAL_F[2] - 1) < 0.10      # (150 steps; 1000 steps: 0.5 %, 0.8 %)


@pytest.mark.gpu
def test_flamingo_closed_loop_with_the_device_policy(gpu_required):
    """:configurationforce mode + TrackingVelocityObjective (dense-LU KKT backend) + policy glue, third model."""
    from contactimplicitmpc.jl_amd import NewtonOptions, InteriorPointOptions
    from contactimplicitmpc.jl_amd.policy import CIMPCPolicy
    d, P, prob, tabs = real_problem("flamingo", KAPPA, False, 1)
    H_mpc, H_sim = 15, 150
    obj = _flamingo_objective(H_mpc)
    pol = CIMPCPolicy(P, obj.q, obj.u, H_mpc=H_mpc, N_sample=N_SAMPLE, B=1, mode=1, obj_gamma=obj.gamma, obj_b=obj.b, obj_v=obj.v,
                      n_opts=NewtonOptions(kappa=KAPPA, r_tol=3e-4, max_iter=5), ip_opts=InteriorPointOptions(kappa_tol=KAPPA, r_tol=1e-8))
    q1, v1 = P.q[1].copy(), (P.q[1] - P.q[0]) / P.h
    ok, q, u, g, b = pl.simulate(pl.FlamingoPlant(), lambda qq, t: pol(qq[t + 1][None])[0], q1, v1, H_sim, P.h / N_SAMPLE, opts=SIM_OPTS_F)
    pol.close()
    assert ok
    e = pl.tracking_error(P.q, P.u, P.gamma, P.b, q, u, g, b, N_SAMPLE)
    assert all(a < 1.5 * n for a, n in zip(e, NOMINAL_F))
    Po, polo, oko, qo, uo, go, bo = _flamingo_oracle_loop(H_sim)
    np.testing.assert_allclose(q[:40], qo[:40], rtol=0, atol=1e-5)
    eo = pl.tracking_error(P.q, P.u, P.gamma, P.b, qo, uo, go, bo, N_SAMPLE)
    for a, c in zip(e, eo):
        assert abs(a / c - 1) < 0.05


def test_hopper_plant_equals_the_torch_model_and_steps_along_the_gait():
    """hopper_2D (BASELINE configs[1]'s model): the numpy plant against the torch model of lcp_models.py, and one simulator
    step from two consecutive configurations of the reference's hopper gait lands on the next one."""
    from contactimplicitmpc.jl_amd import lcp_models
    P, m = pl.HopperPlant(), lcp_models.Hopper2D()
    rng = np.random.default_rng(4)
    for _ in range(3):
        z, th = rng.uniform(0.1, 1.0, m.nz), rng.uniform(0.1, 1.0, m.nth)
        r_t, rz_t, _ = m.linearize(z, th, 1e-3)
        np.testing.assert_allclose(P.residual(z, th, 1e-3), r_t, rtol=0, atol=1e-12)
        np.testing.assert_allclose(P.jacobian_z(z, th), rz_t, rtol=0, atol=1e-11 * np.abs(rz_t).max())
    import os
    from contactimplicitmpc.jl_amd import gait_io
    from real_problems import GAITS
    # (the shipped hopper gait predates the shipped model - tests/test_real_models.py: its leg-length equation is off by 4e-3)
    tr = gait_io.load_joint_traj(os.path.join(os.path.dirname(GAITS["quadruped"][1]), "hopper_gait_forward.jld2"))
    for t in (0, 9, 20, 60):
        status, it, q2, gam, b = pl.plant_step(P, tr.q[t], tr.q[t + 1], tr.u[t], np.zeros(2), P.mu_world, tr.h, pl.SIM_OPTS)
        assert status and it <= 40
        assert np.abs(q2 - tr.q[t + 2])[:3].max() < 2e-3 and abs(q2[3] - tr.q[t + 2][3]) < 2e-2


def test_reference_open_loop_tests():
    """test/simulator/open_loop.jl:1-43 restated: index / counter sequences and values of open_loop_disturbances and
    open_loop_policy (H = 4, N_sample = 3)."""
    from contactimplicitmpc.jl_amd.plant import OpenLoopDisturbance, OpenLoopPolicy
    rng = np.random.default_rng(0)
    H, N_sample = 4, 3
    idx = [1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4]
    cnt = [1, 2, 3, 1, 2, 3, 1, 2, 3, 1, 2, 3]
    w = [rng.random(4) for _ in range(H)]
    d = OpenLoopDisturbance(w, N_sample)
    u = [rng.random(4) for _ in range(H)]
    p = OpenLoopPolicy(u, N_sample)
    for t in range(1, H * N_sample + 1):
        wt, ut = d(t), p(t)
        assert (d.idx, d.cnt) == (idx[t - 1], cnt[t - 1]) and (p.idx, p.cnt) == (idx[t - 1], cnt[t - 1])
        assert np.linalg.norm(wt - w[d.idx - 1] / N_sample) < 1e-10 and np.linalg.norm(ut - u[p.idx - 1] / N_sample) < 1e-10


def test_disturbance_schedules():
    """open_loop_disturbances / impulse_disturbances, src/simulator/disturbances.jl:4-58 (index pattern of the comment at :17-20)."""
    from contactimplicitmpc.jl_amd.plant import ImpulseDisturbance, OpenLoopDisturbance
    w = [np.array([1.0, 0.0]), np.array([0.0, 2.0]), np.array([3.0, 3.0]), np.array([4.0, 0.0])]
    d = OpenLoopDisturbance(w, 3)
    got = [d(t) for t in range(1, 11)]
    idx = [1, 1, 1, 2, 2, 2, 3, 3, 3, 4]
    for g, i in zip(got, idx):
        np.testing.assert_array_equal(g, w[i - 1] / 3)
    np.testing.assert_array_equal(d(1), w[0] / 3)                          # t == 1 resets
    imp = ImpulseDisturbance([np.array([5.0, 0.0]), np.array([0.0, -7.0])], [4, 9])
    out = np.array([imp(t) for t in range(1, 11)])
    assert np.count_nonzero(out.any(axis=1)) == 2 and out[3, 0] == 5.0 and out[8, 1] == -7.0


def test_random_disturbance_schedule():
    """random_disturbances / disturbances(d::RandomDisturbance, x, t), src/simulator/disturbances.jl:63-84: H samples in
    [0, w_amp[1]) on the grid (k - 1) h, piecewise constant (searchsortedlast); the struct uses w_amp[1] whatever w_amp's length."""
    from contactimplicitmpc.jl_amd.plant import RandomDisturbance
    H, h, nw = 20, 0.01, 2
    d = RandomDisturbance(nw, [3.0], H, h, seed=4)
    assert d.w.shape == (H, 1, nw) and (d.w >= 0).all() and (d.w < 3.0).all() and d.w.max() > 1.5
    np.testing.assert_array_equal(d.t, np.arange(H) * h)
    for k in (1, 2, 7, H):                                                   # on a grid point and just before the next one: sample k
        np.testing.assert_array_equal(d((k - 1) * h), d.w[k - 1, 0])
        np.testing.assert_array_equal(d((k - 1) * h + 0.999 * h), d.w[k - 1, 0])
    np.testing.assert_array_equal(d(1e9), d.w[H - 1, 0])                     # beyond the grid: the last sample
    with pytest.raises(IndexError):
        d(-h)
    d2 = RandomDisturbance(nw, [3.0, 100.0], H, h, seed=4)                  # per-axis amplitudes: w_amp[1] is what is used (disturbances.jl:81)
    np.testing.assert_array_equal(d2.w, d.w)
    with pytest.raises(ValueError):
        RandomDisturbance(nw, [1.0, 2.0, 3.0], H, h)
    db = RandomDisturbance(3, 5.0, H, h, seed=1, B=16)                      # a Monte-Carlo batch: independent sequences per robot
    assert db(0.0).shape == (16, 3) and np.unique(db(0.0)[:, 0]).size == 16


def test_centroidal_plant_equals_the_torch_model_and_steps_along_the_trot():
    """centroidal_quadruped (BASELINE configs[4]'s model), damped and undamped: numpy plant against the torch model, and the
    undamped plant stepped from two consecutive configurations of the reference's in-place trot lands on the next one (the
    shipped reference satisfies the undamped dynamics to round-off, tests/test_real_models.py)."""
    from contactimplicitmpc.jl_amd import lcp_models
    rng = np.random.default_rng(8)
    for damped, m in ((True, lcp_models.CentroidalQuadruped()), (False, lcp_models.CentroidalQuadrupedUndamped())):
        P = pl.CentroidalPlant(damped)
        z, th = rng.uniform(0.1, 1.0, m.nz), rng.uniform(0.1, 1.0, m.nth)
        r_t, rz_t, _ = m.linearize(z, th, 1e-3)
        np.testing.assert_allclose(P.residual(z, th, 1e-3), r_t, rtol=0, atol=1e-12)
        np.testing.assert_allclose(P.jacobian_z(z, th), rz_t, rtol=0, atol=1e-11 * np.abs(rz_t).max())
    d, Pr, prob, tabs = real_problem("centroidal", 1e-3, False, 0)
    P = pl.CentroidalPlant(False)
    for t in (0, 13, 40):
        status, it, q2, gam, b = pl.plant_step(P, Pr.q[t], Pr.q[t + 1], Pr.u[t], np.zeros(3), P.mu_world, Pr.h, pl.SIM_OPTS)
        assert status and it <= 40
        assert np.abs(q2 - Pr.q[t + 2]).max() < 2e-3


def test_reference_simulator_test_quadruped_open_loop():
    """test/simulator/quadruped.jl:1-36 restated on the CPU plant: gait2 with mu_world = 0.5 satisfies the residual (norm < 1e-4
    at every knot), and the open-loop simulation of the gait's own controls over T = H steps ends within 0.025 of the reference's
    final base configuration."""
    from contactimplicitmpc.jl_amd.plant import OpenLoopPolicy
    d, P, prob, tabs = real_problem("quadruped", KAPPA, True)          # update_friction_coefficient!: mu_world of the model
    plant = pl.QuadrupedPlant()
    mu = 0.5                                                            # model.mu_world = 0.5 (:6)
    th = P.theta.copy(); th[:, -2] = mu
    s2 = mu * P.gamma - P.b.reshape(P.H, 4, 2).sum(axis=2)              # z packs s2 = mu gamma - E b (index.jl:437-441)
    z = P.z.copy(); z[:, -4:] = s2
    for t in range(P.H):
        assert np.linalg.norm(plant.residual(z[t], th[t], 0.0)) < 1e-4              # :17-20
    pol = OpenLoopPolicy(list(P.u))
    q1, v1 = P.q[1].copy(), (P.q[1] - P.q[0]) / P.h
    ok, q, u, g, b = pl.simulate(plant, lambda qq, t: pol(t + 1), q1, v1, P.H, P.h, mu=mu)
    assert ok                                                                         # @test status
    assert np.abs(P.q[-1][:3] - q[-1][:3]).max() < 0.025                              # :35


def test_reference_simulator_test_particle_drop_and_slide():
    """test/simulator/particle.jl:1-30 restated on the CPU plant: a particle dropped from 1 m is at rest at the origin after 100
    steps of 0.01 s (1e-6); thrown with v = (1, 2, 0) it ends on the ground (1e-6) with zero velocity (1e-6).  Known answers the
    reference holds for the simulator's interior-point loop - here they pin the build-defined spec of DESIGN.md section 3 run
    with the simulator's options.  Also: numpy plant = torch model."""
    from contactimplicitmpc.jl_amd import lcp_models
    P, m = pl.ParticlePlant(), lcp_models.Particle()
    rng = np.random.default_rng(9)
    z, th = rng.uniform(0.1, 1.0, m.nz), rng.uniform(0.1, 1.0, m.nth)
    r_t, rz_t, _ = m.linearize(z, th, 1e-3)
    np.testing.assert_allclose(P.residual(z, th, 1e-3), r_t, rtol=0, atol=1e-13)
    np.testing.assert_allclose(P.jacobian_z(z, th), rz_t, rtol=0, atol=1e-12 * np.abs(rz_t).max())
    h, T = 0.01, 100
    ok, q, *_ = pl.simulate(P, lambda qq, t: np.zeros(3), np.array([0.0, 0.0, 1.0]), np.zeros(3), T, h)          # DROP
    assert ok and np.all(np.abs(q[-1]) < 1e-6)
    ok, q, *_ = pl.simulate(P, lambda qq, t: np.zeros(3), np.array([0.0, 0.0, 1.0]), np.array([1.0, 2.0, 0.0]), T, h)   # SLIDE
    assert ok and abs(q[-1][2]) < 1e-6 and np.all(np.abs((q[-1] - q[-2]) / h) < 1e-6)
