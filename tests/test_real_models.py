"""Real-model inputs for the path: the gait-file reader and the model restatements of
contactimplicitmpc/jl_amd/{gait_io,lcp_models}.py, pinned by the reference's own DATA and TESTS:

  * the shipped gaits satisfy the restated variational dynamics (quadruped gait2: 1e-5; centroidal in-place trot with
    the undamped model: round-off) - the files were produced by the reference's models, so this pins M, C, B, J, the
    integrator and the z / θ packing;
  * test/controller/linearized_solver.jl:23-67 re-stated on the TRUE hopper_2D residual (random z0, θ0 as there);
  * test/controller/implicit_dynamics.jl:7-24 re-stated: quadruped gait2, ImplicitTrajectory at κ = 1e-4,
    κ_tol = 2e-4, r_tol = 1e-8  ->  |dq2|_inf < 1e-2 at every knot.  This runs the oracle's interior point on the
    real linearizations and is the reference's known-answer test for the implicit dynamics.
"""
import os

import numpy as np
import pytest
import torch

from contactimplicitmpc.jl_amd import gait_io, lcp_models
from oracle import ip as oip
from oracle import lcp, newton as onewton, synth
from oracle.dims import Dims
from real_problems import GAITS, real_problem, real_rollout


def test_jld2_reader_reads_the_split_traj_alt_fields():
    g = gait_io.load_gait(GAITS["quadruped"][1])
    assert (g.H, g.q.shape, g.u.shape, g.gamma.shape, g.b.shape, g.psi.shape, g.eta.shape) == \
        (60, (62, 11), (60, 8), (60, 4), (60, 8), (60, 4), (60, 8))
    assert g.h == 0.015625174962268267 and g.mu == 0.5          # values recorded in SURVEY.md section 8c-6
    c = gait_io.load_gait(GAITS["centroidal"][1])
    assert (c.H, c.q.shape, c.u.shape, c.b.shape, c.h, c.mu) == (71, (73, 18), (71, 12), (71, 16), 0.01, 0.3)
    # body height of the in-place trot reference (examples/centroidal_quadruped/reference: body_height = 0.3 -> settles)
    assert 0.2 < c.q[0, 2] < 0.35 and np.all(np.isfinite(c.q))
    raw = gait_io.read_jld2(GAITS["quadruped"][1])
    assert set(raw) >= {"qm", "um", "γm", "bm", "ψm", "ηm", "μm", "hm"}


def test_jld2_reader_rejects_other_files(tmp_path):
    p = tmp_path / "x.jld2"
    p.write_bytes(b"not a gait file" * 100)
    with pytest.raises(gait_io.GaitFormatError):
        gait_io.read_jld2(p)


def _dyn_residual(model, g, kappa=1e-4):
    out = []
    for t in range(g.H):
        z = model.pack_z(g.q[t + 2], g.gamma[t], g.b[t], g.psi[t], g.eta[t])
        th = model.pack_theta(g.q[t], g.q[t + 1], g.u[t], np.zeros(model.nw), g.mu, g.h)
        out.append(model.residual(torch.as_tensor(z), torch.as_tensor(th), torch.tensor(kappa, dtype=torch.float64)).numpy())
    return np.array(out)


def test_shipped_gaits_satisfy_the_restated_dynamics():
    g = gait_io.load_gait(GAITS["quadruped"][1])
    m = lcp_models.Quadruped()
    r = _dyn_residual(m, g)
    assert np.abs(r[:, :m.nq]).max() < 2e-5                       # trajectory-optimiser tolerance of the file
    assert np.abs(r[:, m.nq:m.nq + m.nc]).max() < 1e-12           # s1 = ϕ(q2) by construction of pack_z
    c = gait_io.load_gait(GAITS["centroidal"][1])
    mu = lcp_models.CentroidalQuadrupedUndamped()
    ru = _dyn_residual(mu, c, 1e-3)
    assert np.abs(ru[:, :mu.nq]).max() < 5e-6                     # optimiser tolerance of the file (feet rows: 1e-10)
    # the damped model of the example differs by exactly the joint-friction term  -h c (q2 - q1) / h
    md = lcp_models.CentroidalQuadruped()
    rd = _dyn_residual(md, c, 1e-3)
    fr = np.stack([-(md.joint_friction().numpy()) * (c.q[t + 2] - c.q[t + 1]) for t in range(c.H)])
    np.testing.assert_allclose(rd[:, :md.nq] - ru[:, :mu.nq], fr, rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", ["hopper_2D", "quadruped", "centroidal_quadruped"])
def test_exact_derivatives_and_block_structure(name):
    m = lcp_models.MODELS[name]()
    rng = np.random.default_rng(1)
    z = rng.uniform(0.1, 1.0, m.nz)
    th = rng.uniform(0.1, 1.0, m.nth)
    r0, rz0, rth0 = m.linearize(z, th, 1e-4)
    f = lambda zz, tt: m.residual(torch.as_tensor(zz), torch.as_tensor(tt), torch.tensor(1e-4, dtype=torch.float64)).numpy()
    eps = 1e-6
    Jz = np.stack([(f(z + eps * e, th) - f(z - eps * e, th)) / (2 * eps) for e in np.eye(m.nz)], 1)
    Jt = np.stack([(f(z, th + eps * e) - f(z, th - eps * e)) / (2 * eps) for e in np.eye(m.nth)], 1)
    assert np.abs(Jz - rz0).max() < 1e-7 * max(1.0, np.abs(rz0).max())
    assert np.abs(Jt - rth0).max() < 1e-7 * max(1.0, np.abs(rth0).max())
    # [Dx Dy1 0; Rx Ry1 diag(Ry2); 0 diag(y2) diag(y1)]  (linearized_solver.jl:167-169), Ry2 = 1, rθ0[ibil, :] = 0
    d = Dims(nq=m.nq, nu=m.nu, nw=m.nw, nc=m.nc, nb=m.nb)
    ix, iy1, iy2 = d.ix, d.iy1, d.iy2
    assert np.abs(rz0[np.ix_(ix, iy2)]).max() == 0.0
    np.testing.assert_array_equal(rz0[np.ix_(iy1, iy2)], np.eye(d.ny))
    assert np.abs(rz0[np.ix_(iy2, ix)]).max() == 0.0
    np.testing.assert_allclose(rz0[np.ix_(iy2, iy1)], np.diag(z[iy2]), rtol=0, atol=1e-15)
    np.testing.assert_allclose(rz0[np.ix_(iy2, iy2)], np.diag(z[iy1]), rtol=0, atol=1e-15)
    assert np.abs(rth0[iy2, :]).max() == 0.0


def test_reference_linearized_solver_test_on_the_true_hopper():
    """test/controller/linearized_solver.jl:23-67 with the hopper_2D residual itself (z0, θ0 = rand)."""
    m = lcp_models.Hopper2D()
    d = Dims(nq=m.nq, nu=m.nu, nw=m.nw, nc=m.nc, nb=m.nb)
    rng = np.random.default_rng(0)
    for _ in range(5):
        z0, th0, kappa = rng.random(m.nz), rng.random(m.nth), 1e-4
        r0, rz0, rth0 = m.linearize(z0, th0, kappa)
        tab = lcp.LinTable(d, z0, th0, r0, rz0, rth0)
        ix, iy1, iy2 = d.ix, d.iy1, d.iy2
        for blk, ref in ((tab.Dx, rz0[np.ix_(ix, ix)]), (tab.Dy1, rz0[np.ix_(ix, iy1)]), (tab.Rx, rz0[np.ix_(iy1, ix)]),
                         (tab.Ry1, rz0[np.ix_(iy1, iy1)]), (tab.Ry2, np.diag(rz0[np.ix_(iy1, iy2)])),
                         (tab.y2, np.diag(rz0[np.ix_(iy2, iy1)])), (tab.y1, np.diag(rz0[np.ix_(iy2, iy2)]))):
            assert np.abs(blk - ref).max() < 1e-10                       # test :35-41
        lcp.rzlin(tab, z0)
        rdyn, rrst, rbil = lcp.rlin(tab, z0, th0, kappa)
        assert np.abs(r0[ix] - rdyn).max() < 1e-10 and np.abs(r0[iy1] - rrst).max() < 1e-10 \
            and np.abs(r0[iy2] - rbil).max() < 1e-10                      # test :47-49
        Delta = lcp.linear_solve_vec(tab, rdyn, rrst, rbil)
        x = np.linalg.solve(rz0, r0)
        assert np.abs(Delta - x).max() < 1e-10 * max(1.0, np.abs(x).max())   # test :52-54
        assert np.abs(rth0[iy2, :]).max() < 1e-10                           # test :58
        dz = lcp.linear_solve_mat(tab)
        X = np.linalg.solve(rz0, rth0)
        assert np.abs(dz - X).max() < 1e-10 * max(1.0, np.abs(X).max())     # test :63-67


def test_reference_implicit_dynamics_test_on_gait2():
    """test/controller/implicit_dynamics.jl:7-24: implicit_dynamics!(im_traj, ref_traj) on quadruped gait2 keeps
    |dq2|_inf < 1e-2 at every knot (κ = 1e-4 linearization, κ_tol = 2e-4, r_tol = 1e-8)."""
    d, P, prob, tabs = real_problem("quadruped", 1e-4)
    opts = oip.IPOptions(kappa_tol=2e-4, r_tol=1e-8)
    out = oip.implicit_dynamics(d, tabs, np.arange(P.H + 2), P.q, P.theta, opts)
    assert out["status"].all()
    worst = np.abs(out["d"]).max(axis=1)
    assert np.all(worst[:P.H - 1] < 1e-2), worst.max()
    assert worst.max() > 1e-4           # not trivially zero: κ relaxation + file tolerance move q2 by ~1e-3
    assert 3 <= out["iters"].min() and out["iters"].max() <= 10
    # the centroidal example configuration (continuous_trot.jl:40,60-66: κ_mpc = 1e-3, r_tol = 1e-4) behaves the same
    d2, P2, prob2, tabs2 = real_problem("centroidal", 1e-3)
    out2 = oip.implicit_dynamics(d2, tabs2, np.arange(P2.H + 2), P2.q, P2.theta, oip.IPOptions(kappa_tol=1e-3, r_tol=1e-4))
    assert out2["status"].all() and np.abs(out2["d"]).max() < 1e-2


def test_newton_solve_on_the_real_quadruped_problem():
    """The MPC step of test/controller/mpc_quadruped.jl:19-47 (objective :23-27, κ_mpc = 2e-4, r_tol 3e-4, max_iter 5)
    on the true linearizations: at the reference the first residual is already below tolerance (no Newton step); a
    perturbed initial configuration takes a few."""
    d, P, prob, tabs = real_problem("quadruped", 2e-4)
    H = 40
    obj = synth.make_objective(d, H, kind="quadruped")
    N = H * (d.nr + d.nd)
    its = []
    for perturb, phase in ((0.0, 0), (0.0, 37), (0.05, 13), (0.05, 30)):
        window, ref, q0, q1 = real_rollout(d, prob, H, phase, seed=phase, perturb=perturb)
        core = onewton.Newton(d, H, obj, onewton.NewtonOptions(r_tol=3e-4, max_iter=5, solver="condensed"),
                              oip.IPOptions(kappa_tol=2e-4), 2e-4, ref)
        st = onewton.newton_solve(core, q0, q1, window, tabs, ref)
        assert st.ip_fail == 0
        its.append(st.iters)
        if perturb == 0.0:
            assert st.iters == 0 and st.r_norm / N < 3e-4          # phase 37 wraps around the gait: stride handling
    assert max(its) >= 1


def test_update_altitude_restated():
    """mpc_utils.jl:109-136: the step with the largest normal impulse inside the window decides, per contact point."""
    from contactimplicitmpc.jl_amd.policy import update_altitude
    m = lcp_models.Quadruped()
    g = gait_io.load_gait(GAITS["quadruped"][1])
    q_hist = g.q[2:8].copy()
    q_hist[:, 1] += np.linspace(0.0, 0.05, 6)            # lift the body a little more every step
    gamma = np.zeros((6, 4))
    gamma[2, 0], gamma[4, 0] = 3.0, 2.0                   # contact 1: largest impulse at step 2
    gamma[5, 1] = 0.5                                     # contact 2: below the threshold
    gamma[1, 3] = 1.5                                     # contact 4: step 1
    alt = np.array([9.0, 9.0, 9.0, 9.0])
    update_altitude(m, alt, gamma, q_hist, threshold=1.0)
    phi = lambda q: m.phi(torch.as_tensor(q)).numpy()
    assert alt[0] == phi(q_hist[2])[0] and alt[3] == phi(q_hist[1])[3]
    assert alt[1] == 9.0 and alt[2] == 9.0


def test_reference_trajectory_test_tracking_error_of_the_repeated_gait_is_zero():
    """test/controller/trajectory.jl:1-12: flamingo gait repeated five times (repeat_ref_traj, idx_shift = [1]) against
    itself, N_sample = 1  ->  tracking_error == 0 in all four components."""
    from oracle import plant as pl
    g = gait_io.load_gait(GAITS["flamingo"][1])
    n_rep, H = 5, g.H
    shift = np.zeros(g.q.shape[1]); shift[0] = g.q[-1][0] - g.q[1][0]          # trajectory.jl:87
    q = [g.q[t] for t in range(H + 2)]
    for i in range(1, n_rep):
        q += [g.q[t + 2] + i * shift for t in range(H)]
    rep = lambda a: np.concatenate([a] * n_rep)
    e = pl.tracking_error(g.q, g.u, g.gamma, g.b, np.array(q), rep(g.u), rep(g.gamma), rep(g.b), 1)
    assert e == (0.0, 0.0, 0.0, 0.0)


def test_reference_linearized_step_test_on_the_true_quadruped():
    """test/controller/linearized_step.jl:1-30 (quadruped case): a LinearizedStep at random (z, θ) holds them and the
    residual / Jacobian evaluated there - here: LinTable built from `linearize` reproduces r0 exactly at the point."""
    m = lcp_models.Quadruped()
    d = Dims(nq=m.nq, nu=m.nu, nw=m.nw, nc=m.nc, nb=m.nb)
    rng = np.random.default_rng(5)
    z, th, kappa = rng.random(m.nz), rng.random(m.nth), 1e-5
    r0, rz0, rth0 = m.linearize(z, th, kappa)
    tab = lcp.LinTable(d, z, th, r0, rz0, rth0)
    assert np.abs(tab.x0 - z[d.ix]).max() == 0.0 and np.abs(tab.th0 - th).max() == 0.0
    rdyn, rrst, rbil = lcp.rlin(tab, z, th, kappa)          # "the linearization is exact at the linearization point"
    assert np.abs(np.concatenate([rdyn, rrst, rbil]) - r0).max() < 1e-8
    assert np.abs(lcp.dense_rz(tab, z) - rz0).max() < 1e-8


def test_joint_traj_reader_on_the_hopper_gait():
    """`load_type = :joint_traj` (trajectory.jl:181-182): the serialized ContactTraj of src/dynamics/hopper_2D/gaits/
    gait_forward.jld2.  The file predates the shipped hopper model (its z still has an older ordering and the leg-length
    equation is off by 4e-3), so it pins the READER - field order, shapes, h, κ - and the hopper dynamics only in the three
    body coordinates."""
    t = gait_io.load_joint_traj(os.path.join(os.path.dirname(GAITS["quadruped"][1]), "hopper_gait_forward.jld2"))
    assert (t.H, t.h, t.kappa) == (92, 0.01, 2e-8)
    assert (t.q.shape, t.u.shape, t.w.shape, t.gamma.shape, t.b.shape, t.z.shape, t.theta.shape) == \
        ((94, 4), (92, 2), (92, 2), (92, 1), (92, 2), (92, 12), (92, 14))
    # θ = [q0; q1; u1; w1; μ; h] of the same step (index.jl:413-415)
    k = 17
    np.testing.assert_array_equal(t.theta[k], np.concatenate([t.q[k], t.q[k + 1], t.u[k], t.w[k], [0.8], [0.01]]))
    np.testing.assert_array_equal(t.z[k][:7], np.concatenate([t.q[k + 2], t.gamma[k], t.b[k]]))
    m = lcp_models.Hopper2D()
    res = []
    for k in range(t.H):
        z = m.pack_z(t.q[k + 2], t.gamma[k], t.b[k], np.ones(1), np.ones(2))
        res.append(m.residual(torch.as_tensor(z), torch.as_tensor(t.theta[k]), torch.tensor(0.0, dtype=torch.float64)).numpy()[:4])
    res = np.abs(np.array(res)).max(axis=0)
    assert res[:3].max() < 1e-4 and res[3] < 1e-2
    with pytest.raises(gait_io.GaitFormatError):
        gait_io.load_joint_traj(GAITS["quadruped"][1])          # a :split_traj_alt file has no `traj` object


def test_make_rollout_equals_rot_n_stride_on_the_full_trajectory():
    """The closed form `lcp_models.make_rollout` / `cimpc_set_gait` use for "k applications of rot_n_stride!" (mpc_utils.jl:
    1-101) against the oracle applying it k times to the full reference trajectory: two laps of the quadruped gait."""
    from oracle import mpc as ompc
    d, P, prob, tabs = real_problem("quadruped", 2e-4)
    H = 10
    tr = onewton.Traj(q=P.q.copy(), u=P.u.copy(), w=P.w.copy(), gamma=P.gamma.copy(), b=P.b.copy(), theta=P.theta.copy())
    win = np.arange(H + 2)
    for k in range(2 * P.H + 5):
        r = lcp_models.make_rollout(P, H, k)
        np.testing.assert_array_equal(r["window"], win)
        np.testing.assert_allclose(r["q"], tr.q[:H + 2], rtol=0, atol=1e-12)
        np.testing.assert_array_equal(r["u"], tr.u[:H])
        np.testing.assert_array_equal(r["gamma"], tr.gamma[:H])
        np.testing.assert_allclose(r["theta"], tr.theta[:H], rtol=0, atol=1e-12)
        ompc.rot_n_stride(d, tr, prob["stride"])
        win = ompc.update_window(win, P.H)


def test_implicit_dynamics_on_the_flamingo_gait():
    """The check of test/controller/implicit_dynamics.jl:7-24 on the third planar model: flamingo gait_forward_36_4 (the gait of
    test/controller/mpc_flamingo.jl), linearization at κ = 2e-4: every knot converges in 4-6 iterations and stays within 2e-2 of the gait (1.1e-2 at the worst knot;
    the 1e-2 of the reference's test is recorded for the quadruped only)."""
    d, P, prob, tabs = real_problem("flamingo", 2e-4)
    assert np.abs(P.r0[:, :d.nq]).max() < 2e-6                   # the gait satisfies the restated flamingo dynamics
    out = oip.implicit_dynamics(d, tabs, np.arange(P.H + 2), P.q, P.theta, oip.IPOptions(kappa_tol=2e-4, r_tol=1e-8))
    assert out["status"].all()
    assert np.abs(out["d"]).max() < 2e-2
    assert out["iters"].max() <= 12


def test_reference_schur_ill_conditioning_test_on_the_oracle():
    """test/solver/schur.jl:19-62 restated: flamingo `gait_forward_36_4`, knot 10, z0 = max(1e-6, z), kappa0 = 0,
    M = [A B; C D] with D = rz0[irst, iy1] - rz0[irst, iy2] Diagonal(z0[iy2] ./ z0[iy1]), u = r0[idyn],
    v = r0[irst] - r0[ibil] ./ z0[iy1]; `schur_factorize!` + `schur_solve!`: |M [x; y] - [u; v]|_inf < 1e-7 - the one reference-held
    known answer for the ill-conditioned regime the interior point ends in (cond M ~ 1e11).  The device runs the same statement
    through the B2 seam (tests/test_gpu_round3_parity.py)."""
    d, P, prob, tabs = real_problem("flamingo", 0.0)
    m = P.model
    z0 = np.maximum(1e-6, P.z[9]); th0 = P.theta[9].copy()
    r0, rz0, rth0 = m.linearize(z0, th0, 0.0)
    nx, ny = d.nx, d.ny
    iy1 = np.arange(nx, nx + ny); iy2 = np.arange(nx + ny, nx + 2 * ny)
    A = rz0[:nx, :nx]; B = rz0[:nx, iy1]; C = rz0[iy1, :nx]
    D = rz0[np.ix_(iy1, iy1)] - rz0[np.ix_(iy1, iy2)] @ np.diag(z0[iy2] / z0[iy1])
    M = np.block([[A, B], [C, D]])
    u = r0[:nx]; v = r0[iy1] - r0[iy2] / z0[iy1]
    assert np.linalg.cond(M) > 1e8
    S = lcp.Schur.from_matrix(M, nx)           # Schur(M, n = nx, m = ny)
    S.factorize(D)                             # schur_factorize!(S, D)
    x, y = S.solve(u, v)                       # schur_solve!(S, u, v)
    assert np.abs(M @ np.concatenate([x, y]) - np.concatenate([u, v])).max() < 1e-7
    # the same through the callbacks the interior point uses (rzlin! + linear_solve!, linearized_solver.jl:378-444)
    tab = lcp.LinTable(d, z0, th0, r0, rz0, rth0)
    lcp.rzlin(tab, z0, 0.0)
    delta = lcp.linear_solve_vec(tab, r0[:nx], r0[iy1], r0[iy2], 0.0)
    np.testing.assert_allclose(delta[:nx], x, rtol=0, atol=1e-9 * np.abs(x).max())
    np.testing.assert_allclose(delta[iy1], y, rtol=0, atol=1e-9 * np.abs(y).max())
