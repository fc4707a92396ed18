"""Device plant step (`cimpc_plant_step`) against the CPU restatement of the same step
(oracle/plant.py), and the reference's closed loop with BOTH sides on the device: policy and plant."""
import numpy as np
import pytest

from oracle import ip as oip
from oracle import plant as pl, synth
from real_problems import real_problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,which,mu,eps_min", [("quadruped", "quadruped", 1.0, 0.25), ("flamingo", "flamingo", 0.9, 0.05)])
def test_plant_step_matches_the_cpu_restatement(gpu_required, model, which, mu, eps_min):
    from contactimplicitmpc.jl_amd import InteriorPointOptions, plant
    d, P, prob, tabs = real_problem(which, 2e-4, False, 0)
    cpu = pl.QuadrupedPlant() if model == "quadruped" else pl.FlamingoPlant()
    knots = [0, 7, 19, 33, 48, 57]
    rng = np.random.default_rng(2)
    q0 = np.stack([P.q[t] for t in knots]); q1 = np.stack([P.q[t + 1] for t in knots]) + 1e-3 * rng.standard_normal((len(knots), d.nq))
    u = np.stack([P.u[t] for t in knots]) * np.array([1.0, 0.2, 1.0, 0.2, 1.0, 0.2])[:, None]      # full and N_sample-scaled controls
    for h in (P.h, P.h / 5):
        o_cpu = oip.IPOptions(r_tol=1e-8, kappa_tol=1e-8, undercut=np.inf, gamma_reg=0.1, eps_min=eps_min, max_iter=100, max_ls=25)
        o_dev = InteriorPointOptions(r_tol=1e-8, kappa_tol=1e-8, undercut=float("inf"), eps_min=eps_min, max_iter=100, max_ls=25)
        q2, g, b, st, it = plant.plant_step(model, q0, q1, u, mu, h, opts=o_dev)
        assert st.all()
        for k in range(len(knots)):
            s_, i_, q2c, gc, bc = pl.plant_step(cpu, q0[k], q1[k], u[k], np.zeros(2), mu, h, o_cpu)
            assert s_ and abs(int(it[k]) - i_) <= 1
            np.testing.assert_allclose(q2[k], q2c, rtol=0, atol=1e-7)
            np.testing.assert_allclose(g[k], gc, rtol=0, atol=1e-5 * max(1.0, np.abs(gc).max()))
            np.testing.assert_allclose(b[k], bc, rtol=0, atol=1e-5 * max(1.0, np.abs(bc).max()))


def test_closed_loop_with_policy_and_plant_on_the_device(gpu_required):
    """test/controller/mpc_quadruped.jl with the device policy AND the device plant; two robots side by side (the second one
    starts from a perturbed state) - nothing but q1 / u crosses the host per step."""
    from contactimplicitmpc.jl_amd import InteriorPointOptions, NewtonOptions, plant
    from contactimplicitmpc.jl_amd.policy import CIMPCPolicy
    KAPPA, H_MPC, N_SAMPLE, H_sim = 2e-4, 10, 5, 300
    d, P, prob, tabs = real_problem("quadruped", KAPPA, True)
    obj = synth.make_objective(d, H_MPC, kind="quadruped")
    B = 2
    pol = CIMPCPolicy(P, obj.q, obj.u, H_mpc=H_MPC, N_sample=N_SAMPLE, B=B, n_opts=NewtonOptions(kappa=KAPPA, r_tol=3e-4, max_iter=5),
                      ip_opts=InteriorPointOptions(kappa_tol=KAPPA, r_tol=1e-8))
    q1 = np.stack([P.q[1], P.q[1]]); v1 = np.stack([(P.q[1] - P.q[0]) / P.h] * 2)
    q1[1, 1] += 0.01; q1[1, 3:] += 0.02                                     # robot 2: 1 cm higher, joints 0.02 rad off
    ok, q, u, g, b = plant.simulate("quadruped", pol, q1, v1, H_sim, P.h / N_SAMPLE, mu=1.0)
    pol.close()
    assert ok
    e0 = pl.tracking_error(P.q, P.u, P.gamma, P.b, q[:, 0], u[:, 0], g[:, 0], b[:, 0], N_SAMPLE)
    e1 = pl.tracking_error(P.q, P.u, P.gamma, P.b, q[:, 1], u[:, 1], g[:, 1], b[:, 1], N_SAMPLE)
    nominal = (0.0201, 0.0437, 0.374, 0.0789)
    assert all(a < 1.5 * n for a, n in zip(e0, nominal)) and all(a < 2.0 * n for a, n in zip(e1, nominal))
    assert abs(e0[1] / nominal[1] - 1) < 0.05 and abs(e0[2] / nominal[2] - 1) < 0.05 and abs(e0[3] / nominal[3] - 1) < 0.05


def test_hopper_plant_step_and_disturbances_match_the_cpu_restatement(gpu_required):
    """hopper_2D plant (nc = 1) and the disturbance input w (src/simulator/disturbances.jl: an impulse on one robot, an
    open-loop push on another) against the CPU restatement."""
    import os
    from contactimplicitmpc.jl_amd import InteriorPointOptions, gait_io, plant
    from real_problems import GAITS
    tr = gait_io.load_joint_traj(os.path.join(os.path.dirname(GAITS["quadruped"][1]), "hopper_gait_forward.jld2"))
    cpu = pl.HopperPlant()
    knots = [0, 9, 20, 37, 60, 81]
    rng = np.random.default_rng(5)
    q0 = np.stack([tr.q[t] for t in knots]); q1 = np.stack([tr.q[t + 1] for t in knots]) + 1e-3 * rng.standard_normal((len(knots), 4))
    u = np.stack([tr.u[t] for t in knots])
    w = np.zeros((len(knots), 2)); w[1] = [2.0, 0.0]; w[4] = [-1.0, 0.5]
    o_cpu = oip.IPOptions(r_tol=1e-8, kappa_tol=1e-8, undercut=np.inf, gamma_reg=0.1, eps_min=0.25, max_iter=100, max_ls=25)
    o_dev = InteriorPointOptions(r_tol=1e-8, kappa_tol=1e-8, undercut=float("inf"), eps_min=0.25, max_iter=100, max_ls=25)
    q2, g, b, st, it = plant.plant_step("hopper_2D", q0, q1, u, cpu.mu_world, tr.h, w=w, opts=o_dev)
    assert st.all() and g.shape == (len(knots), 1) and b.shape == (len(knots), 2)
    for k in range(len(knots)):
        s_, i_, q2c, gc, bc = pl.plant_step(cpu, q0[k], q1[k], u[k], w[k], cpu.mu_world, tr.h, o_cpu)
        assert s_ and abs(int(it[k]) - i_) <= 1
        np.testing.assert_allclose(q2[k], q2c, rtol=0, atol=1e-7)
        np.testing.assert_allclose(g[k], gc, rtol=0, atol=1e-5 * max(1.0, np.abs(gc).max()))
        np.testing.assert_allclose(b[k], bc, rtol=0, atol=1e-5 * max(1.0, np.abs(bc).max()))
    # the disturbance is felt: the pushed robots land elsewhere than the same robots without it
    q2n, *_ = plant.plant_step("hopper_2D", q0, q1, u, cpu.mu_world, tr.h, opts=o_dev)
    assert np.abs(q2[1] - q2n[1]).max() > 1e-5 and np.abs(q2[4] - q2n[4]).max() > 1e-5 and np.abs(q2[0] - q2n[0]).max() == 0.0
    # simulate() with a schedule: an impulse at step 3 changes the trajectory from q[4] on
    pol = lambda q: np.tile(tr.u[0], (q.shape[0], 1))
    v1 = (tr.q[1] - tr.q[0])[None] / tr.h
    ok0, qa, *_ = plant.simulate("hopper_2D", pol, tr.q[1][None], v1, 6, tr.h, cpu.mu_world)
    ok1, qb, *_ = plant.simulate("hopper_2D", pol, tr.q[1][None], v1, 6, tr.h, cpu.mu_world,
                                 disturbances=plant.ImpulseDisturbance([np.array([3.0, 0.0])], [3]))
    assert ok0 and ok1
    np.testing.assert_array_equal(qa[:4], qb[:4])
    assert np.abs(qa[4] - qb[4]).max() > 1e-6


@pytest.mark.parametrize("model,damped", [("centroidal_quadruped", True), ("centroidal_quadruped_undamped", False)])
def test_centroidal_plant_step_with_payload_matches_the_cpu_restatement(gpu_required, model, damped):
    """The 3-D centroidal plant (nz = 66: two Jacobian columns / LU rows per lane) against the CPU restatement, with the payload
    of BASELINE configs[4] as the disturbance w (a body force, centroidal_quadruped/model.jl:123-127)."""
    from contactimplicitmpc.jl_amd import InteriorPointOptions, plant
    d, P, prob, tabs = real_problem("centroidal", 1e-3, False, 0)
    cpu = pl.CentroidalPlant(damped)
    knots = [0, 13, 27, 40, 55]
    rng = np.random.default_rng(6)
    q0 = np.stack([P.q[t] for t in knots]); q1 = np.stack([P.q[t + 1] for t in knots]) + 1e-3 * rng.standard_normal((len(knots), 18))
    u = np.stack([P.u[t] for t in knots])
    w = np.zeros((len(knots), 3)); w[1] = [0.0, 0.0, -0.3]; w[3] = [0.1, -0.05, -0.2]          # h * (force): payload on two robots
    o_cpu = oip.IPOptions(r_tol=1e-8, kappa_tol=1e-8, undercut=np.inf, gamma_reg=0.1, eps_min=0.25, max_iter=100, max_ls=25)
    o_dev = InteriorPointOptions(r_tol=1e-8, kappa_tol=1e-8, undercut=float("inf"), eps_min=0.25, max_iter=100, max_ls=25)
    q2, g, b, st, it = plant.plant_step(model, q0, q1, u, cpu.mu_world, P.h, w=w, opts=o_dev)
    assert st.all() and g.shape == (len(knots), 4) and b.shape == (len(knots), 16)
    for k in range(len(knots)):
        s_, i_, q2c, gc, bc = pl.plant_step(cpu, q0[k], q1[k], u[k], w[k], cpu.mu_world, P.h, o_cpu)
        assert s_ and abs(int(it[k]) - i_) <= 1
        np.testing.assert_allclose(q2[k], q2c, rtol=0, atol=1e-7)
        np.testing.assert_allclose(g[k], gc, rtol=0, atol=1e-5 * max(1.0, np.abs(gc).max()))
        np.testing.assert_allclose(b[k], bc, rtol=0, atol=1e-5 * max(1.0, np.abs(bc).max()))
    q2n, *_ = plant.plant_step(model, q0, q1, u, cpu.mu_world, P.h, opts=o_dev)
    assert np.abs(q2[1] - q2n[1]).max() > 1e-6 and np.abs(q2[0] - q2n[0]).max() == 0.0


def test_centroidal_closed_loop_with_payload_on_the_device(gpu_required):
    """examples/centroidal_quadruped/continuous_trot.jl's configuration (H_mpc = 50, velocity objective, kappa 1e-3) with policy
    AND 3-D plant on the device; two robots, one carrying a 30 N payload the controller does not know about (BASELINE
    configs[4]); the example's vertical push at MPC step 10 included.  Both keep trotting in place."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("closed_loop_centroidal", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                         "scripts", "closed_loop_centroidal.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    ok, out = mod.run(steps=250, payload=(0.0, -30.0), verbose=False)
    assert ok                                                            # every plant step converged
    free, loaded = out
    assert free["height_drift_max"] < 0.01 and loaded["height_drift_max"] < 0.03
    assert loaded["height_end"] < free["height_end"] - 0.003            # the payload is felt (≈ 8 mm lower) ...
    assert free["orientation_max"] < 0.05 and loaded["orientation_max"] < 0.05      # ... and both stay level
    assert free["newton_iters"] <= 5.0 and loaded["newton_iters"] <= 5.0


def test_reference_simulator_test_quadruped_open_loop_on_the_device(gpu_required):
    """test/simulator/quadruped.jl:22-35 on the device plant: open-loop simulation of gait2's own controls (mu_world = 0.5) over
    T = H steps ends within 0.025 of the reference's final base configuration (two robots side by side: identical results; an
    open-loop gait is not stable - a robot that starts 2 mm off falls over, so only the nominal start is a test)."""
    from contactimplicitmpc.jl_amd import plant
    d, P, prob, tabs = real_problem("quadruped", 2e-4, True)
    pol = plant.OpenLoopPolicy(list(P.u))
    step = [0]
    def policy(q):
        step[0] += 1
        return np.tile(pol(step[0]), (q.shape[0], 1))
    q1 = np.stack([P.q[1], P.q[1]])
    v1 = np.stack([(P.q[1] - P.q[0]) / P.h] * 2)
    ok, q, u, g, b = plant.simulate("quadruped", policy, q1, v1, P.H, P.h, mu=0.5)
    assert ok
    assert np.abs(P.q[-1][:3] - q[-1, 0, :3]).max() < 0.025
    np.testing.assert_array_equal(q[:, 0], q[:, 1])


def test_reference_simulator_test_particle_on_the_device(gpu_required):
    """test/simulator/particle.jl:1-30 on the device plant: DROP and SLIDE side by side (two robots of one batch)."""
    from contactimplicitmpc.jl_amd import plant
    h, T = 0.01, 100
    q1 = np.array([[0.0, 0.0, 1.0], [0.0, 0.0, 1.0]]); v1 = np.array([[0.0, 0.0, 0.0], [1.0, 2.0, 0.0]])
    ok, q, u, g, b = plant.simulate("particle", lambda qq: np.zeros((2, 3)), q1, v1, T, h, mu=1.0)
    assert ok
    assert np.all(np.abs(q[-1, 0]) < 1e-6)                                               # dropped: at rest at the origin
    assert abs(q[-1, 1, 2]) < 1e-6 and np.all(np.abs((q[-1, 1] - q[-2, 1]) / h) < 1e-6)   # slid: on the ground, stopped
    cpu = pl.ParticlePlant()
    okc, qc, *_ = pl.simulate(cpu, lambda qq, t: np.zeros(3), q1[1], v1[1], T, h)
    np.testing.assert_allclose(q[:, 1], qc, rtol=0, atol=1e-7)                          # and the same path as the CPU restatement
