"""-m gpu: the altitude term RLin.alt on the device.

  rrst += [alt; 0]                         src/controller/linearized_solver.jl:370 (rlin!)
  set_altitude!(im_traj, alt)              src/controller/implicit_dynamics.jl:141-154 (same vector for every knot)
  update_altitude!                         src/controller/mpc_utils.jl:109-136
  policy(p, traj, t): update + set before newton_solve!   src/controller/policy.jl:110-117

cimpc_set_altitude takes one (nc) vector PER ROLLOUT; the kernel adds it to the first nc rows of rrst."""
import numpy as np
import pytest

from oracle import ip as oip
from oracle import newton as onewton, plant as pl, synth

from common import make_case, make_solver, oracle_scatter, oracle_sweep

pytestmark = pytest.mark.gpu


def _set_alt(tabs, a):
    for t in tabs:
        t.alt = np.asarray(a, dtype=np.float64).copy()


def _compare_sweep(d, tabs, rollouts, alt, out, H, kappa):
    """Device vs oracle per solve; tolerances max(nominal, 50 x the oracle's own last-place scatter) - common.oracle_scatter."""
    n = agree = 0
    for b in range(len(rollouts)):
        _set_alt(tabs, alt[b])
        (tr, o), sc, sens = oracle_scatter(d, tabs, rollouts[b], oip.IPOptions(kappa_tol=kappa), K=4, seed=b)
        same = (out["status"][b] == o["status"]) & (out["iters"][b] == o["iters"])
        n += same.size
        agree += int(same.sum())
        for i in np.nonzero(same & (o["status"] == 1))[0]:
            np.testing.assert_allclose(out["z"][b, i, :d.nq], o["z"][i, :d.nq], rtol=0, atol=max(1e-6, 50 * sc["z"][i]))
            np.testing.assert_allclose(out["d"][b, i], o["d"][i], rtol=0, atol=max(1e-7, 50 * sc["d"][i]))
            for k in ("dq0", "dq1", "du1"):
                np.testing.assert_allclose(out[k][b, i], o[k][i], rtol=0, atol=max(1e-6 * max(np.abs(o[k]).max(), 1.0), 50 * sc[k][i]))
    _set_alt(tabs, np.zeros(d.nc))
    return agree, n


@pytest.mark.parametrize("model,mode", [("quadruped", 0), ("pushbot", 1)])
def test_altitude_term_matches_oracle(gpu_required, model, mode):
    B, H, H_ref = 5, 6, 10
    d, prob, tabs, rollouts = make_case(model, mode, H_ref=H_ref, H=H, B=B, seed=13)
    rng = np.random.default_rng(2)
    alt = rng.uniform(-0.002, 0.006, (B, d.nc))     # millimetres (terrain height under a foot), distinct per rollout: a wrong rollout / slot index shows
    alt[2] = 0.0
    s = make_solver(d, prob, rollouts, H)
    ref0 = oracle_sweep(d, tabs, rollouts, oip.IPOptions(kappa_tol=prob["kappa"]))
    q = np.stack([tr.q for tr, _ in ref0]); th = np.stack([tr.theta for tr, _ in ref0])
    g = np.stack([tr.gamma for tr, _ in ref0]); bb = np.stack([tr.b for tr, _ in ref0])
    base = s.implicit_dynamics(q, th, g, bb, want_z=True)
    s.set_altitude(alt)
    out = s.implicit_dynamics(q, th, g, bb, want_z=True)
    agree, n = _compare_sweep(d, tabs, rollouts, alt, out, H, prob["kappa"])
    assert agree >= 0.9 * n, (agree, n)
    # the term is live: rollouts with a non-zero altitude moved, the zero one did not
    for b in range(B):
        dd = np.abs(out["d"][b] - base["d"][b]).max()
        assert (dd == 0.0) if b == 2 else (dd > 1e-6), (b, dd)
    # reset: NULL restores RLin.alt = 0
    s.set_altitude(None)
    again = s.implicit_dynamics(q, th, g, bb, want_z=True)
    np.testing.assert_array_equal(again["d"], base["d"])
    np.testing.assert_array_equal(again["iters"], base["iters"])
    s.close()


def test_newton_solve_with_altitude_matches_oracle(gpu_required):
    from contactimplicitmpc.jl_amd import NewtonOptions
    B, H, H_ref = 4, 8, 12
    d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=H_ref, H=H, B=B, seed=17, perturb=5e-3)
    obj = synth.make_objective(d, H, kind="quadruped")
    alt = np.random.default_rng(5).uniform(0.0, 0.02, (B, d.nc))
    s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-5, max_iter=4))
    s.set_altitude(alt)
    u1, it, rn = s.newton_solve(np.stack([r[2] for r in rollouts]), np.stack([r[3] for r in rollouts]))
    traj = s.trajectory(); cnt = s.rollout_counters()
    same = exact = 0
    for b, (window, ref, q0, q1) in enumerate(rollouts):
        _set_alt(tabs, alt[b])
        core = onewton.Newton(d, H, obj, onewton.NewtonOptions(r_tol=1e-5, max_iter=4, solver="lu"),
                              oip.IPOptions(kappa_tol=prob["kappa"]), prob["kappa"], ref)
        st = onewton.newton_solve(core, q0, q1, window, tabs, ref)
        assert it[b] == st.iters
        # same Newton-level decisions (iterations, evaluated sweeps); of the ~ 250-400 interior-point iterations behind them one may
        # fall on the other side of r_tol = 1e-8, which sits at the round-off floor of the residual (the oracle itself moves
        # by one or two under last-place noise on (q0, q1): rollout 3 of this case) - the values then still agree to ~ 1e-11
        if cnt["sweeps"][b] == st.sweeps and abs(int(cnt["ip_iters"][b]) - st.ip_iters) <= 2:
            same += 1
            exact += int(cnt["ip_iters"][b] == st.ip_iters)
            np.testing.assert_allclose(traj["q"][b], core.traj.q, rtol=0, atol=1e-7)
            np.testing.assert_allclose(u1[b], core.traj.u[0], rtol=0, atol=1e-7)
    _set_alt(tabs, np.zeros(d.nc))
    assert same >= B - 1 and exact >= B // 2, (same, exact)
    s.close()


def test_closed_loop_with_live_altitude(gpu_required):
    """policy.jl:110-117 cadence: before every MPC solve `update_altitude!` looks at the last N_sample simulator steps
    and `set_altitude!` hands the vector to every knot.  Device policy and oracle policy run the same loop on the same
    CPU plant; a terrain offset is added to the measured altitude so that the term is far above round-off."""
    from real_problems import real_problem
    from contactimplicitmpc.jl_amd import NewtonOptions, InteriorPointOptions, lcp_models
    from contactimplicitmpc.jl_amd.policy import CIMPCPolicy, update_altitude
    KAPPA, H_MPC, N_SAMPLE, H_SIM, OFFSET, THRESHOLD = 2e-4, 10, 5, 30, 2e-3, 0.05   # (impulses at h / N_sample stay below the default threshold 1.0)
    d, P, prob, tabs = real_problem("quadruped", KAPPA, True)
    model = lcp_models.Quadruped()
    obj = synth.make_objective(d, H_MPC, kind="quadruped")

    def loop(make_policy, set_alt):
        alt = np.zeros(d.nc)
        hist = []

        def pol(qq, t):
            if t > 0 and t % N_SAMPLE == 0:
                gh = np.array(state["g"][-N_SAMPLE:]); qh = np.array(qq[-N_SAMPLE:])
                update_altitude(model, alt, gh, qh, threshold=THRESHOLD)
                set_alt(alt + OFFSET)
                hist.append(alt.copy())
            return make_policy(qq, t)
        state = {"g": []}
        q1, v1 = P.q[1].copy(), (P.q[1] - P.q[0]) / P.h
        q = [q1 - v1 * P.h / N_SAMPLE, q1.copy()]
        us = []
        plant = pl.QuadrupedPlant()
        ok = True
        for t in range(H_SIM):
            u = pol(q, t)
            status, itn, q2, gam, b = pl.plant_step(plant, q[t], q[t + 1], u, np.zeros(plant.nw), plant.mu_world, P.h / N_SAMPLE, pl.SIM_OPTS)
            ok = ok and status
            q.append(q2); us.append(u.copy()); state["g"].append(gam)
        return ok, np.array(q), np.array(us), hist

    dev = CIMPCPolicy(P, obj.q, obj.u, H_mpc=H_MPC, N_sample=N_SAMPLE, B=1,
                      n_opts=NewtonOptions(kappa=KAPPA, r_tol=3e-4, max_iter=5), ip_opts=InteriorPointOptions(kappa_tol=KAPPA, r_tol=1e-8))
    ok_d, q_d, u_d, hist_d = loop(lambda qq, t: dev(qq[t + 1][None])[0], lambda a: dev.set_altitude(a))
    dev.close()
    ref = onewton.Traj(q=P.q.copy(), u=P.u.copy(), w=P.w.copy(), gamma=P.gamma.copy(), b=P.b.copy(), theta=P.theta.copy())
    orc = pl.OraclePolicy(d, tabs, ref, prob["stride"], obj, H_MPC, N_SAMPLE, KAPPA,
                          onewton.NewtonOptions(r_tol=3e-4, max_iter=5, solver="lu"), oip.IPOptions(kappa_tol=KAPPA, r_tol=1e-8))
    ok_o, q_o, u_o, hist_o = loop(lambda qq, t: orc(qq, t), lambda a: _set_alt(tabs, a))
    _set_alt(tabs, np.zeros(d.nc))
    assert ok_d and ok_o
    assert len(hist_d) >= 5
    # the loop WITHOUT the altitude term (same device policy): what the term changes
    dev0 = CIMPCPolicy(P, obj.q, obj.u, H_mpc=H_MPC, N_sample=N_SAMPLE, B=1,
                       n_opts=NewtonOptions(kappa=KAPPA, r_tol=3e-4, max_iter=5), ip_opts=InteriorPointOptions(kappa_tol=KAPPA, r_tol=1e-8))
    ok_0, q_0, u_0, _ = loop(lambda qq, t: dev0(qq[t + 1][None])[0], lambda a: None)
    dev0.close()
    effect_u, effect_q = np.abs(u_0 - u_o).max(), np.abs(q_0 - q_o).max()
    assert effect_u > 1e-4, effect_u                      # the term is live and far above round-off
    # Device and oracle reproduce the same effect.  (With the offset some interior-point solves of the controller sit on
    # decision boundaries: the ORACLE loop itself moves by 9e-6 in q / 2e-6 in u under a 1e-13 perturbation of q1 -
    # measured on CPU - so the comparison is relative to the size of the effect, not at 1e-7.)
    assert np.abs(u_d - u_o).max() < 0.05 * effect_u, (np.abs(u_d - u_o).max(), effect_u)
    assert np.abs(q_d - q_o).max() < 0.05 * effect_q, (np.abs(q_d - q_o).max(), effect_q)
    np.testing.assert_allclose(q_d[:7], q_o[:7], rtol=0, atol=1e-6)      # up to the first solve with a live altitude: no decision flips yet
