"""Edge cases and error behaviour of the C ABI on the device (SURVEY.md section 8b contract: int status,
no exceptions across the boundary, messages through cimpc_last_error)."""
import numpy as np
import pytest

from oracle import ip as oip, newton as onewton, synth

from common import make_case, make_solver

pytestmark = pytest.mark.gpu


def _fresh(model="hopper", H_ref=8, H=6, B=2, seed=3, set_tables=True):
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
    d, prob, tabs, rollouts = make_case(model, 0, H_ref=H_ref, H=H, B=B, seed=seed)
    s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=B, mode=0,
                    ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]), newton_opts=NewtonOptions(kappa=prob["kappa"]))
    if set_tables:
        for t in range(H_ref):
            s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
    return s, d, prob, tabs, rollouts


def test_call_order_is_enforced(gpu_required):
    from contactimplicitmpc.jl_amd import CimpcError
    s, d, prob, tabs, rollouts = _fresh(set_tables=False)
    q0 = np.stack([r[2] for r in rollouts]); q1 = np.stack([r[3] for r in rollouts])
    with pytest.raises(CimpcError, match="set_linearization"):
        s.newton_solve(q0, q1)
    for t in range(8):
        s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
    with pytest.raises(CimpcError, match="set_window"):
        s.newton_solve(q0, q1)
    s.set_window(np.stack([w for (w, _, _, _) in rollouts]) + 1)
    with pytest.raises(CimpcError, match="set_objective"):
        s.newton_solve(q0, q1)


def test_bad_arguments_are_rejected(gpu_required):
    from contactimplicitmpc.jl_amd import CIMPCSolver, CimpcError
    s, d, prob, tabs, rollouts = _fresh()
    w = np.stack([w for (w, _, _, _) in rollouts]) + 1
    bad = w.copy(); bad[0, 0] = 0                       # 1-based knots: 0 is out of range
    with pytest.raises(CimpcError, match="window"):
        s.set_window(bad)
    bad[0, 0] = 9                                       # H_ref = 8
    with pytest.raises(CimpcError, match="window"):
        s.set_window(bad)
    with pytest.raises(CimpcError):
        s.set_linearization(0, prob["z0"][0], prob["th0"][0], prob["r0"][0], prob["rz0"][0], prob["rth0"][0])
    with pytest.raises(CimpcError):                     # beyond the runtime-dimension kernel (nq, ny <= 64)
        CIMPCSolver(70, 3, 2, 1, 2, 8, 6, B=1, mode=0)
    with pytest.raises(CimpcError):
        CIMPCSolver(5, 3, 2, 12, 48, 8, 6, B=1, mode=0)      # ny = 72


def test_single_rollout_minimum_horizon_and_wrapping_window(gpu_required):
    """B = 1, H = 1 (the shortest horizon) and a window that wraps around the reference."""
    from contactimplicitmpc.jl_amd import NewtonOptions
    for H, H_ref, phase in ((1, 4, 3), (3, 4, 2)):
        d, prob, tabs, _ = make_case("hopper", 0, H_ref=H_ref, H=H, B=1, seed=6)
        ro = [synth.make_rollout(d, prob, H, phase=phase, seed=66, perturb=5e-3)]
        assert ro[0][0].max() == H_ref - 1 and np.any(np.diff(ro[0][0]) < 0)      # the window wraps
        obj = synth.make_objective(d, H, kind="hopper")
        s = make_solver(d, prob, ro, H, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-5, max_iter=3))
        u1, it, rn = s.newton_solve(ro[0][2][None], ro[0][3][None])
        window, ref, q0, q1 = ro[0]
        core = onewton.Newton(d, H, obj, onewton.NewtonOptions(r_tol=1e-5, max_iter=3, solver="lu"),
                              oip.IPOptions(kappa_tol=prob["kappa"]), prob["kappa"], ref)
        st = onewton.newton_solve(core, q0, q1, window, tabs, ref)
        assert it[0] == st.iters
        np.testing.assert_allclose(u1[0], core.traj.u[0], rtol=0, atol=1e-7)


def test_zero_newton_iterations_and_time_budget(gpu_required):
    """max_iter = 0: only the initial evaluation; an already exhausted wall-clock budget ends the solve
    silently like the reference (newton.jl:187-277) - the trajectory stays a valid iterate."""
    from contactimplicitmpc.jl_amd import NewtonOptions
    d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=10, H=8, B=12, seed=8, perturb=2e-2)
    obj = synth.make_objective(d, 8)
    q0 = np.stack([r[2] for r in rollouts]); q1 = np.stack([r[3] for r in rollouts])
    s0 = make_solver(d, prob, rollouts, 8, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-7, max_iter=0))
    u1, it, rn = s0.newton_solve(q0, q1)
    assert (it == 0).all() and np.isfinite(rn).all()
    sb = make_solver(d, prob, rollouts, 8, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-9, max_iter=50, max_time=1e-9))
    u1b, itb, rnb = sb.newton_solve(q0, q1)
    assert (itb <= 50).all() and np.isfinite(u1b).all() and np.isfinite(sb.trajectory()["q"]).all()
    assert itb.max() < 50          # the budget, not max_iter, ended it


def test_repeated_solves_are_deterministic(gpu_required):
    """The same cold-start solve twice on one handle, and on a second handle: identical bits (the schedule -
    rounds, queue order, which workgroup served what - must not leak into the arithmetic)."""
    from contactimplicitmpc.jl_amd import NewtonOptions
    d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=12, H=8, B=40, seed=12, perturb=3e-2)
    obj = synth.make_objective(d, 8)
    q0 = np.stack([r[2] for r in rollouts]); q1 = np.stack([r[3] for r in rollouts])
    outs = []
    for k in range(2):
        s = make_solver(d, prob, rollouts, 8, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-6, max_iter=5))
        for rep in range(2):
            u1, it, rn = s.newton_solve(q0, q1)
            outs.append((u1.copy(), it.copy(), s.trajectory()["q"].copy()))
    for o in outs[1:]:
        np.testing.assert_array_equal(o[1], outs[0][1])
        np.testing.assert_array_equal(o[0], outs[0][0])
        np.testing.assert_array_equal(o[2], outs[0][2])


def test_objective_target_shapes_are_validated(gpu_required):
    """cimpc_set_objective copies H * nq doubles from q_target / v_target: the host wrapper must refuse anything else
    (a (nq,) vector is broadcast over the horizon explicitly) instead of letting the C side read past the buffer."""
    d, prob, tabs, rollouts = make_case("hopper", 0, H_ref=8, H=6, B=2, seed=2)
    obj = synth.make_objective(d, 6, kind="hopper", velocity=True)
    s = make_solver(d, prob, rollouts, 6)
    vt = 0.01 * np.arange(d.nq)
    s.set_objective(obj.q, obj.u, V=obj.v, v_target=vt)                       # (nq,) -> broadcast
    s.set_objective(obj.q, obj.u, V=obj.v, v_target=np.tile(vt[None], (6, 1)))
    with pytest.raises(ValueError):
        s.set_objective(obj.q, obj.u, V=obj.v, v_target=np.zeros((5, d.nq)))    # wrong horizon
    with pytest.raises(ValueError):
        s.set_objective(obj.q, obj.u, V=obj.v, q_target=np.zeros(d.nq + 1))
    s.close()


def test_plant_step_validates_options(gpu_required):
    from contactimplicitmpc.jl_amd import InteriorPointOptions, plant
    q = np.zeros((1, 11)); u = np.zeros((1, 8))
    with pytest.raises(Exception):
        plant.plant_step("quadruped", q, q, u, mu=0.5, h=0.01, opts=InteriorPointOptions(max_iter=0))


def test_linear_solve_csc_standalone_seam(gpu_required):
    """B1 stand-alone (`linear_solve!(solver, x, A::SparseMatrixCSC, b)`, lu.jl:4-12): the matrix that is passed in is the
    matrix that is solved - the oracle's jacobian! as a sparse matrix, and the construction of test/solver/lu.jl (random
    sparse + identity), against scipy / numpy."""
    import scipy.sparse as sp
    from contactimplicitmpc.jl_amd.solver import linear_solve_csc
    from oracle import newton as onewton
    from common import oracle_sweep
    from oracle import ip as oip
    d, prob, tabs, rollouts = make_case("hopper", 0, H_ref=10, H=8, B=1, seed=3)
    obj = synth.make_objective(d, 8, kind="hopper")
    (tr, o), = oracle_sweep(d, tabs, rollouts, oip.IPOptions(kappa_tol=prob["kappa"]))
    lay = onewton.Layout(d, 8)
    R = onewton.jacobian(lay, obj, o, 1e-5, prob["kappa"])
    rng = np.random.default_rng(0)
    b = rng.standard_normal(lay.N)
    x = linear_solve_csc(sp.csc_matrix(R), b)
    np.testing.assert_allclose(R @ x, b, rtol=0, atol=1e-10 * max(1.0, np.abs(b).max()))
    n = 120
    A = sp.random(n, n, density=0.05, random_state=1, format="csc") + sp.identity(n, format="csc") * 4.0     # test/solver/lu.jl
    b = rng.standard_normal(n)
    x = linear_solve_csc(A, b)
    assert np.abs(A @ x - b).max() < 1e-10
    with pytest.raises(ValueError):
        linear_solve_csc(sp.random(4, 5, density=0.5, format="csc"), np.zeros(4))


def test_newton_status_log_matches_oracle(gpu_required):
    """cimpc_get_newton_log: the per-iteration status print_status shows (newton.jl:290-301) - step length, residual before /
    after - per rollout, against the oracle's record of the same solve."""
    from contactimplicitmpc.jl_amd import NewtonOptions
    from oracle import ip as oip, newton as onewton
    H, B = 10, 6
    d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=16, H=H, B=B, seed=11, perturb=1e-2)
    obj = synth.make_objective(d, H, kind="quadruped")
    s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-5, max_iter=5))
    u1, it, rn = s.newton_solve(np.stack([r[2] for r in rollouts]), np.stack([r[3] for r in rollouts]))
    log = s.newton_log()
    s.close()
    assert log.shape == (B, 16, 4) and it.max() >= 1
    for b, (window, ref, q0, q1) in enumerate(rollouts):
        core = onewton.Newton(d, H, obj, onewton.NewtonOptions(r_tol=1e-5, max_iter=5, solver="lu"), oip.IPOptions(kappa_tol=prob["kappa"]), prob["kappa"], ref)
        st = onewton.newton_solve(core, q0, q1, window, tabs, ref)
        assert it[b] == st.iters
        n = int(it[b])
        np.testing.assert_allclose(log[b, :n, 0], st.alphas[:n], rtol=0, atol=0)            # accepted step lengths
        if n:
            np.testing.assert_allclose(log[b, n - 1, 2], rn[b], rtol=1e-12)                   # residual after the last step
            assert (log[b, 1:n, 1] == log[b, :n - 1, 2]).all()                               # before(l+1) == after(l)


def test_linear_solve_csc_reports_a_singular_matrix(gpu_required):
    """lu.jl:4-12 contract at the stand-alone B1 seam: the reference's lu_solver throws SingularException for a singular
    matrix; the device solve reports CIMPC_ERR_SINGULAR (numpy.linalg.LinAlgError here) instead of returning Inf / NaN as OK."""
    import scipy.sparse as sp
    from contactimplicitmpc.jl_amd.solver import linear_solve_csc
    n = 12
    A = np.eye(n); A[5, 5] = 0.0                                  # an empty pivot column after elimination
    with pytest.raises(np.linalg.LinAlgError):
        linear_solve_csc(sp.csc_matrix(A), np.ones(n))
    A = np.ones((n, n))                                           # rank one
    with pytest.raises(np.linalg.LinAlgError):
        linear_solve_csc(sp.csc_matrix(A), np.ones(n))
    A = np.eye(n) * 2.0
    np.testing.assert_allclose(linear_solve_csc(sp.csc_matrix(A), np.ones(n)), 0.5 * np.ones(n))       # (the handle-less seam still works afterwards)


def test_kkt_solve_rho_reads_the_regularisation_a_linear_solver_sees(gpu_required):
    """B1 as the Julia LinearSolver uses it (julia/CIMPCHip.jl: HipKKTSolver): linear_solve!(solver, x, A, b) sees the matrix, not
    core.beta - rho = -A[N, N] (jac.reg_du, newton_jacobian.jl:185).  cimpc_kkt_solve_rho(rho) == cimpc_kkt_solve(beta) with
    rho = H beta kappa, and solves the oracle's assembled jacobian! for that beta."""
    from oracle import ip as oip, newton as onewton
    from common import oracle_sweep
    H = 8
    d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=10, H=H, B=2, seed=11)
    obj = synth.make_objective(d, H, kind="quadruped")
    s = make_solver(d, prob, rollouts, H, obj=obj)
    q = np.stack([r.q for (_, r, _, _) in rollouts]); th = np.stack([r.theta for (_, r, _, _) in rollouts])
    out = s.implicit_dynamics(q, th)
    lay = onewton.Layout(d, H)
    r = np.random.default_rng(2).standard_normal((2, lay.N))
    for beta in (1e-5, 10.0, 7.6923076923076925):
        R0 = onewton.jacobian(lay, obj, {k: out[k][0] for k in ("d", "dq0", "dq1", "du1")}, beta, prob["kappa"])
        rho = -R0[-1, -1]                                         # what the solver reads off the matrix it is handed
        assert rho > 0
        a = s.kkt_solve_rho(r, rho); c = s.kkt_solve(r, beta)
        np.testing.assert_allclose(a, c, rtol=0, atol=1e-12 * max(1.0, np.abs(c).max()))
        np.testing.assert_allclose(R0 @ a[0], r[0], rtol=0, atol=1e-8 * max(1.0, np.abs(r[0]).max()))
    s.close()


def test_b3_and_b4_seams_share_one_sensitivity_store(gpu_required):
    """The reference keeps ONE ip[t].dz per knot, written by implicit_dynamics! (B3) and by every evaluation inside
    newton_solve! (B4) alike (implicit_dynamics.jl:71-86, 169-176).  Mixed use of the two seams on one handle: a B3
    call whose solves all FAIL right after a Newton solve returns the Newton solve's accepted sensitivities - not the blocks of
    the older B3 call -, and a Newton solve after a B3 call falls back to that call's blocks."""
    from contactimplicitmpc.jl_amd import NewtonOptions
    H, B = 6, 2
    d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=8, H=H, B=B, seed=4, perturb=1e-2)
    obj = synth.make_objective(d, H, kind="quadruped")
    q = np.stack([r.q for (_, r, _, _) in rollouts]); th = np.stack([r.theta for (_, r, _, _) in rollouts])
    q0 = np.stack([r[2] for r in rollouts]); q1 = np.stack([r[3] for r in rollouts])
    s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-9, max_iter=2))
    first = s.implicit_dynamics(q, th)
    assert first["status"].all()
    u1, it, rn = s.newton_solve(q0, q1)
    assert it.min() >= 1 and s.rollout_counters()["ip_failures"].sum() == 0
    tr = s.trajectory()
    # what the accepted evaluation of that Newton solve left in the store: B3 at the accepted trajectory, on a second handle
    th_acc = th.copy()
    th_acc[:, :, :d.nq] = tr["q"][:, :H]; th_acc[:, :, d.nq:2 * d.nq] = tr["q"][:, 1:H + 1]; th_acc[:, :, 2 * d.nq:2 * d.nq + d.nu] = tr["u"]
    s2 = make_solver(d, prob, rollouts, H, obj=obj)
    acc = s2.implicit_dynamics(tr["q"], th_acc)
    s2.close()
    assert acc["status"].all() and np.abs(acc["dq1"] - first["dq1"]).max() > 1e-9      # the Newton solve really moved the blocks
    bad = np.full_like(th, np.nan)                     # no solve of this call can converge: every block keeps the store's content
    fail = s.implicit_dynamics(q, bad)
    assert not fail["status"].any()
    for k in ("dq0", "dq1", "du1"):
        np.testing.assert_allclose(fail[k], acc[k], rtol=0, atol=1e-9 * max(1.0, np.abs(acc[k]).max()))
    s.close()
