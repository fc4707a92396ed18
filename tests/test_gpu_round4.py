"""Round 4 device tests (VERDICT r03 "missing" #2, #6):

* re-linearisation in flight: `update!` of RLin / RZLin / RthLin for some knots between two warm-started solves
  (/root/reference/src/controller/linearized_solver.jl:497-565, linearized_step.jl:33-45; the reference's own test of the
  updated operators: test/controller/linearized_solver.jl:70-130) = `cimpc_set_linearization` called again after solves.
  Everything that caches something of a knot on the device must be seen to refresh: the table staged in LDS per launch, the
  parked interior-point iterates, the per-knot sensitivity archive that failed solves fall back to.
* per-solve interior-point time budget: InteriorPointOptions(max_time = mpc_opts.ip_max_time)
  (policy.jl:9,61; implicit_dynamics.jl:21-33) = `cimpc_ip_opts::max_time`.
"""
import os

import numpy as np
import pytest

from oracle import ip as oip, lcp, newton as onewton, synth

from common import make_case, make_solver, oracle_sweep


def _perturbed_knot(prob, t, rng, eps=2e-2):
    """A nearby linearization of knot t: same sparsity (element-wise scaling), shifted point and residual."""
    z0 = prob["z0"][t] * (1.0 + eps * rng.standard_normal(prob["z0"][t].shape))
    z0 = np.where(prob["z0"][t] > 0, np.abs(z0), z0)              # complementarity variables stay positive
    th0 = prob["th0"][t] + 1e-3 * rng.standard_normal(prob["th0"][t].shape)
    r0 = prob["r0"][t] + 1e-4 * rng.standard_normal(prob["r0"][t].shape)
    rz0 = prob["rz0"][t] * (1.0 + eps * rng.standard_normal(prob["rz0"][t].shape))
    rth0 = prob["rth0"][t] * (1.0 + eps * rng.standard_normal(prob["rth0"][t].shape))
    return z0, th0, r0, rz0, rth0


@pytest.mark.gpu
@pytest.mark.parametrize("ip_budget", [None, 5])
def test_relinearisation_between_warm_started_solves(gpu_required, ip_budget):
    """solve -> set_linearization for a subset of the knots -> warm-started solve -> again with another subset, against the
    oracle run on the updated LinTables.  ip_budget = 5 makes a good share of the solves fail, so the stale sensitivity blocks
    (kept per KNOT, implicit_dynamics.jl:71-86,169-176 - untouched by the update, like the reference's ip[t].dz) are used too.

    What is asserted, per (step, rollout) pair - measured on this case (scripts/dbg/relin_debug.py, profiles/r04/relin_debug.log):
    the oracle's two KKT backends (two roundings of the same Newton step) end up to 2e-4 apart on the re-linearised, worse
    conditioned problems, and a cold solve with identical iteration totals can sit 4.7e-7 from the oracle (a converged
    interior-point solve is unique up to kappa_tol) - so every pair is held to max(5e-6, 5 x the oracle's own
    backend distance), and the pairs on which the oracle is stable (backends 1e-9 apart, same counters as the device) are counted
    at 1e-7: at least two of them AFTER an update (there the device was measured at 3e-9 / 7e-10).  A table that is not refreshed
    somewhere (LDS copy, parked iterate, per-knot archive) shows up at 1e-3 .. 1e-2."""
    from contactimplicitmpc.jl_amd import InteriorPointOptions, NewtonOptions
    H, H_ref, B = 8, 12, 4
    d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=H_ref, H=H, B=B, seed=31, perturb=5e-3)
    obj = synth.make_objective(d, H)
    ipk = dict(max_iter=ip_budget) if ip_budget else {}
    s = make_solver(d, prob, rollouts, H, obj=obj, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"], **ipk),
                    newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-5, max_iter=4))
    mk = lambda sv: [onewton.Newton(d, H, obj, onewton.NewtonOptions(r_tol=1e-5, max_iter=4, solver=sv),
                                    oip.IPOptions(kappa_tol=prob["kappa"], **ipk), prob["kappa"], ref) for (_, ref, _, _) in rollouts]
    cores, cores2 = mk("lu"), mk("condensed")
    tabs = list(tabs)
    rng = np.random.default_rng(7)
    q0 = np.stack([r[2] for r in rollouts]); q1 = np.stack([r[3] for r in rollouts])
    tight = tight_after = same_counters = pairs = 0
    for step, knots in enumerate(([], [1, 4, 5, 10], [0, 4, 7])):
        for t in knots:                       # update!(lin, s, z, theta) of these knots, on both sides
            new = _perturbed_knot(prob, t, rng)
            s.set_linearization(t + 1, *new)
            tabs[t] = lcp.LinTable(d, *new)
        u1, it, rn = s.newton_solve(q0, q1, warm_start=step > 0)
        tr = s.trajectory(); cnt = s.rollout_counters()
        if step > 0:
            assert it.max() >= 1          # the update changed the problem: on unchanged tables a warm-started solve needs no iteration
        for b, (window, ref, a, b_) in enumerate(rollouts):
            st = onewton.newton_solve(cores[b], a, b_, window, tabs, ref, warm_start=step > 0)
            st2 = onewton.newton_solve(cores2[b], a, b_, window, tabs, ref, warm_start=step > 0)
            d_oo = max(np.abs(cores2[b].traj.u[0] - cores[b].traj.u[0]).max(), np.abs(cores2[b].traj.q - cores[b].traj.q).max())
            du = np.abs(u1[b] - cores[b].traj.u[0]).max(); dq = np.abs(tr["q"][b] - cores[b].traj.q).max()
            on_path = it[b] == st.iters and cnt["sweeps"][b] == st.sweeps
            if not ip_budget:
                assert on_path, (step, b, it[b], st.iters)
            elif not on_path:         # (5-iteration budget: a search is up to 29 evaluations deep - a flipped decision ends the comparison)
                continue
            pairs += 1
            assert max(du, dq) <= max(5e-6, 5.0 * d_oo), (step, b, du, dq, d_oo)
            same = cnt["ip_iters"][b] == st.ip_iters and cnt["ip_failures"][b] == st.ip_fail
            same_counters += int(same)
            if same and d_oo < 1e-9 and (st2.ip_iters, st2.ip_fail) == (st.ip_iters, st.ip_fail) and max(du, dq) <= 1e-7:
                tight += 1
                tight_after += int(step > 0)
        if ip_budget and step == 0:
            assert cnt["ip_failures"].sum() > 0          # the stale-block path is really exercised
    if not ip_budget:
        assert tight >= 4 and tight_after >= 2 and same_counters >= 6, (tight, tight_after, same_counters)
    else:
        assert pairs >= 6 and same_counters >= 3, (pairs, same_counters)
    s.close()


@pytest.mark.gpu
def test_relinearisation_is_seen_by_implicit_dynamics(gpu_required):
    """B3 seam after an update: d, dz, status, iters of the updated knots follow the new table, the others do not move."""
    H, H_ref, B = 8, 12, 3
    d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=H_ref, H=H, B=B, seed=33, perturb=5e-3)
    s = make_solver(d, prob, rollouts, H)
    opts = oip.IPOptions(kappa_tol=prob["kappa"])
    ref = oracle_sweep(d, tabs, rollouts, opts)
    q = np.stack([t.q for t, _ in ref]); th = np.stack([t.theta for t, _ in ref])
    before = s.implicit_dynamics(q, th)
    tabs = list(tabs)
    rng = np.random.default_rng(3)
    changed = [2, 3, 9]
    for t in changed:
        new = _perturbed_knot(prob, t, rng)
        s.set_linearization(t + 1, *new)
        tabs[t] = lcp.LinTable(d, *new)
    after = s.implicit_dynamics(q, th)
    ref2 = oracle_sweep(d, tabs, rollouts, opts)
    n = ok = 0
    for b, ((window, _, _, _), (_, o)) in enumerate(zip(rollouts, ref2)):
        for i in range(H):
            touched = int(window[i]) in changed
            if not touched:                                  # untouched knots: bit-identical to the first evaluation
                np.testing.assert_array_equal(after["d"][b, i], before["d"][b, i])
                np.testing.assert_array_equal(after["iters"][b, i], before["iters"][b, i])
                continue
            n += 1
            if after["iters"][b, i] == o["iters"][i] and after["status"][b, i] == o["status"][i]:
                ok += 1
                np.testing.assert_allclose(after["d"][b, i], o["d"][i], rtol=0, atol=1e-6)
                assert np.abs(after["d"][b, i] - before["d"][b, i]).max() > 1e-9      # and it did change
    assert n >= 3 and ok >= n - 1, (n, ok)
    s.close()


def _sweep_inputs(B=6, H=8, H_ref=12, seed=41):
    d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=H_ref, H=H, B=B, seed=seed, perturb=2e-2)
    ref = oracle_sweep(d, tabs, rollouts, oip.IPOptions(kappa_tol=prob["kappa"]))
    return d, prob, rollouts, np.stack([t.q for t, _ in ref]), np.stack([t.theta for t, _ in ref]), ref


@pytest.mark.gpu
def test_ip_time_budget(gpu_required):
    """cimpc_ip_opts::max_time: a generous budget changes nothing (bit-identical results, also through park / resume with a small
    iteration cap per launch); a budget no solve can meet ends every solve at its first check like an exhausted iteration count
    (status 0, no sensitivities written, d from the iterate it stopped at); max_iter >= 128 with a budget is refused."""
    from contactimplicitmpc.jl_amd import CIMPCSolver, CimpcError, InteriorPointOptions
    d, prob, rollouts, q, th, ref = _sweep_inputs()
    H = 8
    base = make_solver(d, prob, rollouts, H)
    want = base.implicit_dynamics(q, th)
    base.close()
    assert (want["status"] == 1).mean() > 0.8
    # generous budget, default schedule
    s = make_solver(d, prob, rollouts, H, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"], max_time=10.0))
    got = s.implicit_dynamics(q, th)
    for k in ("d", "dq0", "dq1", "du1", "status", "iters"):
        np.testing.assert_array_equal(got[k], want[k])
    s.close()
    # generous budget, solves parked every 2 iterations (the time used so far rides in the parked state)
    old = os.environ.get("CIMPC_ITER_CAP")
    os.environ["CIMPC_ITER_CAP"] = "2"
    try:
        s = make_solver(d, prob, rollouts, H, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"], max_time=10.0))
    finally:
        if old is None:
            os.environ.pop("CIMPC_ITER_CAP", None)
        else:
            os.environ["CIMPC_ITER_CAP"] = old
    got = s.implicit_dynamics(q, th)
    for k in ("d", "dq0", "dq1", "du1", "status", "iters"):
        np.testing.assert_array_equal(got[k], want[k])
    s.close()
    # a budget of 10 ns: nothing converges, nothing runs past its first iteration
    s = make_solver(d, prob, rollouts, H, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"], max_time=1e-8))
    got = s.implicit_dynamics(q, th)
    assert (got["status"] == 0).all() and got["iters"].max() <= 1
    # ... and newton_solve! with such a budget returns (every evaluation fails, the loop ends on its own terms)
    s.set_objective(*(lambda o: (o.q, o.u))(synth.make_objective(d, H)))
    u1, it, rn = s.newton_solve(np.stack([r[2] for r in rollouts]), np.stack([r[3] for r in rollouts]))
    assert np.isfinite(u1).all() and s.stats()["ip_failures"] == s.stats()["ip_solves"] > 0
    s.close()
    with pytest.raises(CimpcError):
        CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, 12, H, B=1, ip_opts=InteriorPointOptions(max_iter=200, max_time=0.5))


@pytest.mark.gpu
def test_ip_time_budget_runtime_dimension_kernel(gpu_required):
    """The same option on the runtime-dimension sweep (ip_generic.hip)."""
    from contactimplicitmpc.jl_amd import InteriorPointOptions
    d, prob, tabs, rollouts = make_case("anydims", 0, H_ref=6, H=4, B=2, seed=5, perturb=1e-2)
    ref = oracle_sweep(d, tabs, rollouts, oip.IPOptions(kappa_tol=prob["kappa"]))
    q = np.stack([t.q for t, _ in ref]); th = np.stack([t.theta for t, _ in ref])
    s = make_solver(d, prob, rollouts, 4)
    want = s.implicit_dynamics(q, th); s.close()
    s = make_solver(d, prob, rollouts, 4, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"], max_time=10.0))
    got = s.implicit_dynamics(q, th); s.close()
    for k in ("d", "dq0", "dq1", "du1", "status", "iters"):
        np.testing.assert_array_equal(got[k], want[k])
    s = make_solver(d, prob, rollouts, 4, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"], max_time=1e-8))
    got = s.implicit_dynamics(q, th); s.close()
    assert (got["status"] == 0).all() and got["iters"].max() <= 1


@pytest.mark.gpu
def test_config4_velocity_objective_h60_vs_oracle(gpu_required):
    """BASELINE configs[4] with the example's OWN objective (examples/centroidal_quadruped/continuous_trot.jl:43-47: a
    TrackingVelocityObjective, Q singular along a common x shift) at the bench leg's size H = 60, on rollouts of the bench leg's
    exact inputs (`bench.centroidal_payload_inputs(64, 60)`, payload w_z = -5 .. -30 N): the device's banded L D L^T KKT stage against
    the numpy oracle's dense LU of the assembled `jacobian!` matrix (N = 2880; a subset of the 64 rollouts - the oracle takes ~10 s per
    rollout).  VERDICT r03 "missing" #4: this form had been tested at H = 50, B = 3 only."""
    import bench
    from oracle.dims import Dims
    from oracle.newton import Traj
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
    from contactimplicitmpc.jl_amd.trajectory import Objective
    H, NB = 60, 4
    I = bench.centroidal_payload_inputs(64, H)
    m, P, kappa = I["m"], I["P"], I["kappa"]
    ro = I["rollouts"][:NB // 2] + I["rollouts"][-(NB - NB // 2):]
    Q, R, V, vt = bench.centroidal_velocity_objective(m, H)
    s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=NB, mode=0, ip_opts=InteriorPointOptions(kappa_tol=kappa, r_tol=1e-4),
                    newton_opts=NewtonOptions(kappa=kappa, r_tol=3e-5, max_iter=5))
    for t in range(P.H):
        s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
    s.set_objective(Q, R, V=V, v_target=vt)
    s.set_window(np.stack([r["window"] for r in ro]) + 1)
    s.set_reference(*(np.stack([r[k] for r in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")))
    u1, it, rn = s.newton_solve(np.stack([r["q0"] for r in ro]), np.stack([r["q1"] for r in ro]))
    tr = s.trajectory(); cnt = s.rollout_counters()
    s.close()
    d = Dims(nq=m.nq, nu=m.nu, nw=m.nw, nc=m.nc, nb=m.nb, mode=0)
    tabs = [lcp.LinTable(d, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t]) for t in range(P.H)]
    obj = Objective(q=Q, u=R, gamma=np.zeros((H, m.nc, m.nc)), b=np.zeros((H, m.nb, m.nb)), v=V, v_target=vt)
    assert np.abs(np.stack([r["w"] for r in ro])[:, :, 2]).min() >= 5.0          # the payload is live
    same = 0
    for b, r in enumerate(ro):
        ref = Traj(q=r["q"], u=r["u"], w=r["w"], gamma=r["gamma"], b=r["b"], theta=r["theta"])
        core = onewton.Newton(d, H, obj, onewton.NewtonOptions(r_tol=3e-5, max_iter=5, solver="lu"),
                              oip.IPOptions(kappa_tol=kappa, r_tol=1e-4), kappa, ref)
        st = onewton.newton_solve(core, r["q0"], r["q1"], r["window"], tabs, ref)
        assert it[b] == st.iters and st.iters >= 2, (b, it[b], st.iters)
        path = cnt["sweeps"][b] == st.sweeps and cnt["ip_iters"][b] == st.ip_iters
        du = np.abs(u1[b] - core.traj.u[0]).max() / max(1.0, np.abs(core.traj.u[0]).max())
        dq = np.abs(tr["q"][b] - core.traj.q).max()
        if path:        # kappa = 1e-3, IP r_tol 1e-4: far from the round-off floor - the shared path is held tightly
            same += 1
            assert du < 1e-6 and dq < 1e-7, (b, du, dq)
        else:
            assert du < 5e-2 and dq < 1e-2, (b, du, dq)
        np.testing.assert_allclose(rn[b], st.r_norm / core.lay.N, rtol=2e-2 if path else 0.15, atol=1e-12)
    assert same >= NB - 1, same


@pytest.mark.gpu
@pytest.mark.parametrize("form", [1, 2, 3, 4])
def test_banded_kernel_forms_agree(gpu_required, monkeypatch, form):
    """kkt_dense.hip: kkt_banded_kernel is compiled in four forms - eight or four pivots per block, window padded to a power of two or not -
    and every model in the tree takes the same one (eight, padded).  CIMPC_BANDED_FORM forces the others (read in cimpc_create); each
    applies the same operations to every entry in the same order, so one KKT solve of BASELINE configs[4] with the example's velocity
    objective (H = 60, N = 2160 after the control elimination, w = 107) must agree with the default form to the last bit.  Form 4 keeps
    the controls in the matrix (what a singular R_t falls back to: N = 2880, w = 131 - four pivots per block, unpadded window, four rows
    per DPP row in P2): another elimination order, compared to 1e-9 of the solution's scale."""
    import bench
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
    H = 60
    I = bench.centroidal_payload_inputs(2, H)
    m, P, kappa, ro = I["m"], I["P"], I["kappa"], I["rollouts"][:2]
    Q, R, V, vt = bench.centroidal_velocity_objective(m, H)

    def solve():
        s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=2, mode=0, ip_opts=InteriorPointOptions(kappa_tol=kappa, r_tol=1e-4),
                        newton_opts=NewtonOptions(kappa=kappa, r_tol=3e-5, max_iter=5))
        for t in range(P.H):
            s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
        s.set_objective(Q, R, V=V, v_target=vt)
        s.set_window(np.stack([r["window"] for r in ro]) + 1)
        s.set_reference(*(np.stack([r[k] for r in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")))
        s.implicit_dynamics(np.stack([r["q"] for r in ro]), np.stack([r["theta"] for r in ro]))
        d = np.array(s.kkt_solve(np.random.default_rng(0).standard_normal((2, s.N)), 10.0))
        s.close()
        return d

    # (the forms are variants of the ONE-chain kernel; the default for this size is the twisted pair of round 5, whose results differ
    #  in the last digits by construction - tests/test_gpu_round5.py::test_twisted_banded_ldl_vs_dense_lu compares it with both)
    monkeypatch.setenv("CIMPC_KKT_TWISTED", "0")
    monkeypatch.delenv("CIMPC_BANDED_FORM", raising=False)
    ref = solve()
    monkeypatch.setenv("CIMPC_BANDED_FORM", str(form))
    got = solve()
    assert np.isfinite(ref).all() and np.abs(ref).max() > 1.0
    if form < 4:
        np.testing.assert_array_equal(got, ref)
    else:
        assert np.abs(got - ref).max() <= 1e-9 * np.abs(ref).max(), np.abs(got - ref).max() / np.abs(ref).max()
