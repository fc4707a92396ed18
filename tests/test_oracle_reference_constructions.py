"""Pins the oracle against the known-answer constructions of the reference's own tests
(SURVEY.md section 8c items 1-5).  The reference ships no golden vectors; its tests are
random-data self-consistency checks, re-stated here with the same sizes and tolerances.

    test/solver/qr.jl:3-13, 16-24        MGS-QR  A x = b            (1e-10)
    test/solver/schur.jl:3-16            Schur, swap D              (1e-9)
    test/controller/linearized_solver.jl:55-57, 63-67   Delta == rz0 \\ r0, dz == rz0 \\ rth0   (1e-10)
    test/controller/newton.jl:41,58,111,115-135,192-205  sizes, symmetry, block placement, residual
    test/controller/newton_structure_solver.jl:119-178   condensed solve == dense J \\ r   (1e-12 -> 1e-9 here:
                                         their blocks have cond ~ 1, ours come from IP sensitivities)
"""
import numpy as np
import pytest

from oracle import ip as oip
from oracle import lcp, newton as onewton, synth
from oracle.dims import Dims, HOPPER_2D, QUADRUPED


def test_mgs_qr_vector_n16():
    rng = np.random.default_rng(0)
    n = 16
    A = rng.random((n, n)); b = rng.random(n)
    qs, rs = lcp.mgs_factorize(A)
    x = lcp.qr_solve(qs, rs, b)
    assert np.abs(A @ x - b).max() < 1e-10
    # packed R really is the upper-triangular factor: A = Q R
    R = np.zeros((n, n))
    for j in range(n):
        for k in range(j + 1):
            R[k, j] = rs[lcp.triu_perm(k + 1, j + 1)]
    assert np.abs(np.stack(qs, axis=1) @ R - A).max() < 1e-12


def test_mgs_qr_matrix_n43_m33():
    rng = np.random.default_rng(1)
    n, m = 43, 33
    A = rng.random((n, n)); B = rng.random((n, m))
    qs, rs = lcp.mgs_factorize(A)
    X = lcp.qr_matrix_solve(qs, rs, B)
    assert np.linalg.norm(A @ X - B) < 1e-10 * 10   # Frobenius norm over 43x33 entries


def test_schur_swap_D():
    rng = np.random.default_rng(2)
    n, m = 11, 16
    M = rng.random((n + m, n + m))
    S = lcp.Schur.from_matrix(M, n)
    u = rng.random(n); v = rng.random(m); D = rng.random((m, m))
    S.factorize(D)
    x, y = S.solve(u, v)
    M1 = np.block([[S.A, S.B], [S.C, D]])
    assert np.abs(M1 @ np.concatenate([x, y]) - np.concatenate([u, v])).max() < 1e-9


@pytest.mark.parametrize("model", [HOPPER_2D, QUADRUPED])
def test_linear_solve_equals_dense(model):
    """linearized_solver.jl test: the Schur path equals the dense rz0 \\ r0 and rz0 \\ rth0 with
    rz0 = [Dx Dy1 0; Rx Ry1 diag(Ry2); 0 diag(y2) diag(y1)] (linearized_solver.jl:167-169)."""
    d = Dims(**model)
    rng = np.random.default_rng(3)
    prob = synth.make_problem(d, 3, seed=4)
    t = 1
    z0 = prob["z0"][t].copy()
    z0[d.nq:] = rng.uniform(0.05, 1.0, 2 * d.ny)          # generic (not central-path) point, like rand(nz)
    rz0 = prob["rz0"][t].copy()
    rz0[d.nq + d.ny:, d.nq:d.nq + d.ny] = np.diag(z0[d.iy2])
    rz0[d.nq + d.ny:, d.nq + d.ny:] = np.diag(z0[d.iy1])
    r0 = rng.random(d.nz)
    tab = lcp.LinTable(d, z0, prob["th0"][t], r0, rz0, prob["rth0"][t])
    # block slices (test :35-41)
    assert np.abs(tab.Dx - rz0[:d.nq, :d.nq]).max() < 1e-10
    assert np.abs(tab.Ry2 - np.diag(rz0[d.iy1][:, d.iy2])).max() < 1e-10
    lcp.rzlin(tab, z0)
    rdyn, rrst, rbil = lcp.rlin(tab, z0, prob["th0"][t], 0.0)
    assert np.abs(rdyn - r0[:d.nq]).max() < 1e-10 and np.abs(rrst - r0[d.iy1]).max() < 1e-10
    Delta = lcp.linear_solve_vec(tab, r0[:d.nq], r0[d.iy1], r0[d.iy2])
    assert np.abs(Delta - np.linalg.solve(rz0, r0)).max() < 1e-9
    dz = lcp.linear_solve_mat(tab)
    ref = np.linalg.solve(rz0, prob["rth0"][t])
    assert np.abs(dz - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
    assert np.abs(prob["rth0"][t][d.iy2, :]).max() < 1e-10          # rth0[ibil,:] == 0 (test :61)


def _newton_fixture(H=6, seed=5):
    d = Dims(**QUADRUPED)
    prob = synth.make_problem(d, 8, seed=seed)
    tabs = [lcp.LinTable(d, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t]) for t in range(8)]
    window, ref, q0, q1 = synth.make_rollout(d, prob, H, phase=2, seed=9, perturb=2e-2)
    tr = ref.copy(); tr.q[0], tr.q[1] = q0, q1; tr.update_theta(d, 0); tr.update_theta(d, 1)
    im = oip.implicit_dynamics(d, tabs, window, tr.q, tr.theta, oip.IPOptions())
    return d, prob, tabs, window, ref, tr, im


def test_newton_layout_and_jacobian_blocks():
    H = 6
    d, prob, tabs, window, ref, tr, im = _newton_fixture(H)
    lay = onewton.Layout(d, H)
    assert lay.N == H * (d.nr + d.nd)                                     # newton.jl test :41,58
    obj = synth.make_objective(d, H)
    beta, kappa = 1e-5, prob["kappa"]
    R = onewton.jacobian(lay, obj, im, beta, kappa)
    assert np.abs(R - R.T).max() == 0.0                                   # symmetric (:111)
    for t in range(H):
        assert np.array_equal(R[np.ix_(lay.pq(t), lay.pq(t))], obj.q[t])   # (:115-118)
        assert np.array_equal(R[np.ix_(lay.pu(t), lay.pu(t))], obj.u[t])
        assert np.all(R[lay.pz(t), lay.dual(t)] == -1.0)                   # IV == -1 (:122)
        assert np.array_equal(R[np.ix_(lay.dual(t), lay.pu(t))], im["du1"][t])
        if t >= 1:
            assert np.array_equal(R[np.ix_(lay.dual(t), lay.pq(t - 1))], im["dq1"][t])
        if t >= 2:
            assert np.array_equal(R[np.ix_(lay.dual(t), lay.pq(t - 2))], im["dq0"][t])
    dual = np.arange(H * lay.nr, lay.N)
    assert np.allclose(R[dual, dual], -H * beta * kappa, rtol=1e-14)       # reg_du quirk
    # residual formula spot check (:192-205)
    nu = np.random.default_rng(0).standard_normal((H, d.nd))
    r = onewton.residual(lay, obj, nu, im, tr, ref)
    t = 2
    expect = obj.q[t] @ (tr.q[t + 2] - ref.q[t + 2]) - nu[t] + im["dq1"][t + 1].T @ nu[t + 1] + im["dq0"][t + 2].T @ nu[t + 2]
    assert np.abs(r[lay.pq(t)] - expect).max() < 1e-12
    assert np.abs(r[lay.dual(t)] - im["d"][t]).max() == 0.0


@pytest.mark.parametrize("dense_q", [False, True])
def test_condensed_solve_equals_dense(dense_q):
    """newton_structure_solver.jl test: Delta from the condensed solve == J \\ r."""
    H = 8
    d, prob, tabs, window, ref, tr, im = _newton_fixture(H, seed=6)
    lay = onewton.Layout(d, H)
    obj = synth.make_objective(d, H, dense_q=dense_q)
    r = np.random.default_rng(1).standard_normal(lay.N)
    for beta in (1e-5, 10.0):
        R = onewton.jacobian(lay, obj, im, beta, prob["kappa"])
        x = np.linalg.solve(R, r)
        y = onewton.kkt_solve_condensed(lay, obj, im, beta, prob["kappa"], r)
        assert np.abs(x - y).max() < 1e-9 * max(1.0, np.abs(x).max())


def test_ip_converges_like_reference_boundary_tests():
    """test/solver/ldl.jl:104-111 + test/controller/implicit_dynamics.jl:22-24 pin the RoboDojo
    boundary only through the converged answer: status, |r|inf < r_tol, dz != 0, |dq2| < 1e-2."""
    d = Dims(**QUADRUPED)
    prob = synth.make_problem(d, 10, seed=7, kappa=1e-4)     # kappa = 1e-4 as in the reference test
    opts = oip.IPOptions(kappa_tol=1e-4)
    for t in range(10):
        tab = lcp.LinTable(d, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
        z = oip.z_initialize(d, prob["q_ref"][t + 2])
        status, iters, dz = oip.interior_point_solve(tab, z, prob["th0"][t], opts)
        assert status
        rdyn, rrst, rbil = lcp.rlin(tab, z, prob["th0"][t], 0.0)
        assert max(np.abs(rdyn).max(), np.abs(rrst).max()) < opts.r_tol
        assert np.abs(rbil).max() < opts.kappa_tol
        assert np.abs(dz).sum() != 0.0
        assert np.abs(z[:d.nq] - prob["q_ref"][t + 2]).max() < 1e-2
        assert np.all(z[d.nq:] > 0.0)


@pytest.mark.parametrize("model", ["quadruped", "hopper", "centroidal", "hopper3d"])
def test_adjoint_form_of_the_sensitivity_pass(model):
    """The device's :configuration sensitivity pass (ip_kernel_impl.h: sensitivities, Model::ADJ) forms the x rows of
    linear_solve!(dz, rz, rth) (linearized_solver.jl:451-479) as K0 + A2 Gs with nx solves against M^T; the oracle's model of
    that schedule against the reference's column-by-column routine, at converged interior-point iterates (where M carries
    y2 / y1 on its diagonal: cond ~ 1e6).  The two differ by the rounding of a QR of M against a QR of M^T, ~ cond * eps."""
    from common import MODELS
    d = Dims(**MODELS[model])
    prob = synth.make_problem(d, 6, seed=5, kappa=2e-4)
    opts = oip.IPOptions(kappa_tol=2e-4)
    nths = 2 * d.nq + d.nu
    for t in range(6):
        tab = lcp.LinTable(d, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
        z = oip.z_initialize(d, prob["q_ref"][t + 2])
        status, iters, dz = oip.interior_point_solve(tab, z, prob["th0"][t], opts)
        assert status
        reg = opts.kappa_tol * opts.gamma_reg
        lcp.rzlin(tab, z, reg=reg)
        want = lcp.linear_solve_mat(tab, reg=reg)[d.ix, :nths]
        got = lcp.linear_solve_mat_x_adjoint(tab, reg=reg, ncols=nths)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9 * max(1.0, np.abs(want).max()))


def test_configurationforce_elimination_identity():
    """The identity the device uses to solve :configurationforce KKT systems with the :configuration solvers
    (newton_kernels.hip: cf_reduce_*): with contact-impulse weights G,
        Dnu_y = G Dy - r_y ,   (I + rho G) Dy = S Dtheta + rho r_y - r_nuy ,
    and the system for (Du, Dq, Dnu_q) is the :configuration system with right-hand side r_x + S^T r_y (+ O(G)).
    Checked in numpy on the oracle's jacobian! for G = 1e-100 (the reference's value): elimination == dense solve."""
    from oracle.dims import Dims as D_
    d = D_(**QUADRUPED, mode=1)
    H = 6
    prob = synth.make_problem(d, 8, seed=2)
    tabs = [lcp.LinTable(d, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t]) for t in range(8)]
    window, ref, q0, q1 = synth.make_rollout(d, prob, H, phase=1, seed=3, perturb=1e-2)
    tr = ref.copy(); tr.q[0], tr.q[1] = q0, q1; tr.update_theta(d, 0); tr.update_theta(d, 1)
    im = oip.implicit_dynamics(d, tabs, window, tr.q, tr.theta, oip.IPOptions(), gamma=tr.gamma, b=tr.b)
    obj = synth.make_objective(d, H, kind="quadruped")
    lay = onewton.Layout(d, H)
    beta, kappa = 1e-5, prob["kappa"]
    R = onewton.jacobian(lay, obj, im, beta, kappa)
    r = np.random.default_rng(0).standard_normal(lay.N)
    x = np.linalg.solve(R, r)
    # the reduced (:configuration) problem: same sensitivities restricted to the q rows
    d0 = D_(**QUADRUPED, mode=0)
    lay0 = onewton.Layout(d0, H)
    nq, nu, ny, nr, nd = d.nq, d.nu, d.nc + d.nb, d.nr, d.nd
    im0 = dict(d=im["d"][:, :nq], dq0=im["dq0"][:, :nq], dq1=im["dq1"][:, :nq], du1=im["du1"][:, :nq])
    obj0 = onewton.Objective(q=obj.q, u=obj.u)
    R0 = onewton.jacobian(lay0, obj0, im0, beta, kappa)
    nr0 = nu + nq
    r0 = np.zeros(lay0.N)
    Sy = lambda i, key: im[key][i][nq:]                          # y rows of the sensitivities of step i
    for j in range(H):
        ry = lambda i: r[i * nr + nu:i * nr + nu + ny]
        r0[j * nr0:j * nr0 + nu] = r[j * nr:j * nr + nu] + Sy(j, "du1").T @ ry(j)
        v = r[j * nr + nu + ny:j * nr + nr].copy()
        if j + 1 < H: v += Sy(j + 1, "dq1").T @ ry(j + 1)
        if j + 2 < H: v += Sy(j + 2, "dq0").T @ ry(j + 2)
        r0[j * nr0 + nu:(j + 1) * nr0] = v
        r0[H * nr0 + j * nq:H * nr0 + (j + 1) * nq] = r[H * nr + j * nd:H * nr + j * nd + nq]
    x0 = np.linalg.solve(R0, r0)
    rho = H * beta * kappa
    for j in range(H):
        np.testing.assert_allclose(x[j * nr:j * nr + nu], x0[j * nr0:j * nr0 + nu], rtol=0, atol=1e-9 * max(1, np.abs(x).max()))
        np.testing.assert_allclose(x[j * nr + nu + ny:(j + 1) * nr], x0[j * nr0 + nu:(j + 1) * nr0], rtol=0, atol=1e-9 * max(1, np.abs(x).max()))
        dq0 = x0[(j - 2) * nr0 + nu:(j - 1) * nr0] if j >= 2 else np.zeros(nq)
        dq1 = x0[(j - 1) * nr0 + nu:j * nr0] if j >= 1 else np.zeros(nq)
        du = x0[j * nr0:j * nr0 + nu]
        dy = Sy(j, "dq0") @ dq0 + Sy(j, "dq1") @ dq1 + Sy(j, "du1") @ du + rho * r[j * nr + nu:j * nr + nu + ny] \
            - r[H * nr + j * nd + nq:H * nr + (j + 1) * nd]
        np.testing.assert_allclose(x[j * nr + nu:j * nr + nu + ny], dy, rtol=0, atol=1e-9 * max(1, np.abs(x).max()))
        np.testing.assert_allclose(x[H * nr + j * nd + nq:H * nr + (j + 1) * nd], -r[j * nr + nu:j * nr + nu + ny], rtol=0, atol=1e-9 * max(1, np.abs(x).max()))


@pytest.mark.parametrize("model,velocity", [("hopper", True), ("quadruped", True), ("quadruped", False)])
def test_interleaved_banded_ldl(model, velocity):
    """Structure behind the banded KKT backend (kkt_dense.hip: kkt_banded_kernel), on the oracle's jacobian!: in the
    interleaved ordering the matrix has half-bandwidth 3 (nr + nd) - 1 - nu, D has exactly H nd negative pivots
    (quasi-definite), and the blocked sliding-window L D L^T WITHOUT pivoting (oracle/banded.py, the kernel's algorithm)
    reproduces the dense solve."""
    from oracle import banded
    from oracle.dims import HOPPER_2D as HP
    d = Dims(**(HP if model == "hopper" else QUADRUPED))
    H = 5
    prob = synth.make_problem(d, 7, seed=4)
    tabs = [lcp.LinTable(d, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t]) for t in range(7)]
    window, ref, q0, q1 = synth.make_rollout(d, prob, H, phase=1, seed=3, perturb=1e-2)
    tr = ref.copy(); tr.q[0], tr.q[1] = q0, q1; tr.update_theta(d, 0); tr.update_theta(d, 1)
    im = oip.implicit_dynamics(d, tabs, window, tr.q, tr.theta, oip.IPOptions())
    obj = synth.make_objective(d, H, kind=model, velocity=velocity)
    if velocity:
        obj.v = obj.v * 1e3; obj.__post_init__()
    lay = onewton.Layout(d, H)
    R = onewton.jacobian(lay, obj, im, 1e-5, prob["kappa"])
    p = banded.interleave_perm(lay, d)
    A = R[np.ix_(p, p)]
    w = banded.half_bandwidth(d)
    i, j = np.nonzero(A)
    assert np.abs(i - j).max() == min(w, lay.N - 1)
    r = np.random.default_rng(0).standard_normal(lay.N)
    x, lmax = banded.blocked_ldl_solve(A, r[p], min(w, lay.N - 1))
    xs = np.linalg.solve(A, r[p])
    back = np.abs(A @ x - r[p]).max() / (np.abs(A).sum(axis=1).max() * np.abs(x).max() + np.abs(r).max())
    assert back < 1e-12 and lmax < 1e5
    np.testing.assert_allclose(x, xs, rtol=0, atol=1e-11 * np.linalg.cond(A) * max(1.0, np.abs(xs).max()))


@pytest.mark.parametrize("model,velocity", [("hopper", True), ("quadruped", True), ("quadruped", False)])
def test_banded_ldl_control_elimination_and_chain_form(model, velocity):
    """The two things round 4 changed in the banded KKT backend, on the oracle's jacobian!:
    (1) the controls are eliminated first - u_i sits in its own block only, so the Schur complement on the controls touches the
    nu_i x nu_i blocks and nothing else, the kept matrix [q_{i+2}, nu_i] per step is still quasi-definite with half-bandwidth
    3 (nr - nu + nd) - 1, and the recovered solution is the dense one;
    (2) the factorisation is organised as diagonal block / rows below / lower-triangle update per eight pivots in a power-of-two
    window (oracle/banded.py: chain_bulk_ldl_solve) - the same factors as the one-panel form."""
    from oracle import banded
    from oracle.dims import HOPPER_2D as HP
    d = Dims(**(HP if model == "hopper" else QUADRUPED))
    H = 5
    prob = synth.make_problem(d, 7, seed=4)
    tabs = [lcp.LinTable(d, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t]) for t in range(7)]
    window, ref, q0, q1 = synth.make_rollout(d, prob, H, phase=1, seed=3, perturb=1e-2)
    tr = ref.copy(); tr.q[0], tr.q[1] = q0, q1; tr.update_theta(d, 0); tr.update_theta(d, 1)
    im = oip.implicit_dynamics(d, tabs, window, tr.q, tr.theta, oip.IPOptions())
    obj = synth.make_objective(d, H, kind=model, velocity=velocity)
    if velocity:
        obj.v = obj.v * 1e3; obj.__post_init__()
    lay = onewton.Layout(d, H)
    R = onewton.jacobian(lay, obj, im, 1e-5, prob["kappa"])
    r = np.random.default_rng(1).standard_normal(lay.N)
    xs = np.linalg.solve(R, r)
    A, rr, recover = banded.eliminate_controls(R, r, lay, d)
    kept, us = banded.reduced_perm(lay, d)
    # (1) only the dual diagonal blocks change, the band shrinks, the inertia is that of the duals
    diff = np.abs(A - R[np.ix_(kept, kept)]) > 0
    s = d.nr - d.nu + d.nd
    for i, j in zip(*np.nonzero(diff)):
        assert i // s == j // s and i % s >= d.nr - d.nu and j % s >= d.nr - d.nu, (i, j)
    w = banded.half_bandwidth_reduced(d)
    i, j = np.nonzero(A)
    assert np.abs(i - j).max() == min(w, len(kept) - 1) and w < banded.half_bandwidth(d)
    assert (np.linalg.eigvalsh(A) < 0).sum() == H * d.nd
    scale = max(1.0, np.abs(xs).max())
    x1, lmax1 = banded.blocked_ldl_solve(A, rr, min(w, len(kept) - 1))
    np.testing.assert_allclose(recover(x1), xs, rtol=0, atol=1e-11 * np.linalg.cond(R) * scale)
    # ... and entry by entry it is what the kernel generates when a row enters its window (BandRows::desc restated: banded.reduced_entry)
    rho = H * 1e-5 * prob["kappa"]
    gen = np.array([[banded.reduced_entry(d, H, obj, im, rho, i_, j_) if j_ <= i_ else 0.0 for j_ in range(len(kept))] for i_ in range(len(kept))])
    np.testing.assert_allclose(gen, np.tril(A), rtol=0, atol=1e-12 * np.abs(A).max())
    # (2) the chain / bulk organisation gives the same factors: same solution to round-off of the SAME operations
    x2, lmax2 = banded.chain_bulk_ldl_solve(A, rr, min(w, len(kept) - 1))
    assert lmax2 == pytest.approx(lmax1, rel=1e-12) and lmax2 < 1e5
    np.testing.assert_allclose(x2, x1, rtol=0, atol=1e-13 * np.linalg.cond(A) * max(1.0, np.abs(x1).max()))
    np.testing.assert_allclose(recover(x2), xs, rtol=0, atol=1e-11 * np.linalg.cond(R) * scale)


@pytest.mark.parametrize("model,H", [("hopper", 4), ("hopper", 9), ("quadruped", 5), ("quadruped", 12)])
def test_twisted_condensed_solve(model, H):
    """Two-ended (twisted) block factorisation of the dual Schur complement Y (recurrences of
    newton_structure_solver/methods.jl:466-557 run from both ends towards a 2 x 2 block system in the middle): for EVERY position of
    the middle the solution is that of the one-ended condensed solve and of the reference's dense LU of jacobian! - the CPU statement
    of the next KKT kernel (two chains of H / 2 block steps), oracle/newton.py: kkt_solve_condensed_twisted."""
    from oracle.dims import HOPPER_2D as HP
    d = Dims(**(HP if model == "hopper" else QUADRUPED))
    prob = synth.make_problem(d, H + 2, seed=4)
    tabs = [lcp.LinTable(d, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t]) for t in range(H + 2)]
    window, ref, q0, q1 = synth.make_rollout(d, prob, H, phase=1, seed=3, perturb=1e-2)
    tr = ref.copy(); tr.q[0], tr.q[1] = q0, q1; tr.update_theta(d, 0); tr.update_theta(d, 1)
    im = oip.implicit_dynamics(d, tabs, window, tr.q, tr.theta, oip.IPOptions())
    obj = synth.make_objective(d, H, kind=model, velocity=False)
    lay = onewton.Layout(d, H)
    r = np.random.default_rng(0).standard_normal(lay.N)
    for beta in (1e-5, 1e-2, 10.0):
        R = onewton.jacobian(lay, obj, im, beta, prob["kappa"])
        xd = onewton.kkt_solve_lu(R, r)
        x0 = onewton.kkt_solve_condensed(lay, obj, im, beta, prob["kappa"], r)
        scale = max(1.0, np.abs(xd).max())
        for split in [None] + list(range(H - 1)):
            x1 = onewton.kkt_solve_condensed_twisted(lay, obj, im, beta, prob["kappa"], r, split=split)
            assert np.abs(x1 - x0).max() <= 1e-9 * scale, (beta, split)
            assert np.abs(x1 - xd).max() <= 1e-10 * np.linalg.cond(R) * scale, (beta, split)


@pytest.mark.parametrize("model,H", [("hopper", 9), ("quadruped", 12), ("quadruped", 20)])
def test_twisted_device_schedule(model, H):
    """The twisted solve restated with the DEVICE kernel's data flow (oracle/newton.py: kkt_solve_condensed_twisted_device - the
    bottom chain forms its blocks of the reversed matrix from the operands it streams downwards, rings indexed by the chain step,
    two trace steps, the top chain corrected in its last two rows, one-level backward passes on stage C's records, split primal
    recovery; csrc/newton_impl.h: kkt_body<..., TW>): for every admissible split the solution of the one-ended condensed solve and
    of the dense LU of jacobian! (newton_structure_solver/methods.jl:466-557 from both ends; lu.jl:4-12)."""
    from oracle.dims import HOPPER_2D as HP
    d = Dims(**(HP if model == "hopper" else QUADRUPED))
    prob = synth.make_problem(d, H + 2, seed=4)
    tabs = [lcp.LinTable(d, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t]) for t in range(H + 2)]
    window, ref, q0, q1 = synth.make_rollout(d, prob, H, phase=1, seed=3, perturb=1e-2)
    tr = ref.copy(); tr.q[0], tr.q[1] = q0, q1; tr.update_theta(d, 0); tr.update_theta(d, 1)
    im = oip.implicit_dynamics(d, tabs, window, tr.q, tr.theta, oip.IPOptions())
    obj = synth.make_objective(d, H, kind=model, velocity=False)
    lay = onewton.Layout(d, H)
    r = np.random.default_rng(0).standard_normal(lay.N)
    assert onewton.kkt_tw_split(40) == 22 and onewton.kkt_tw_split(60) == 32 and onewton.kkt_tw_split(10) == 6      # (= the header's formula)
    for beta in (1e-5, 1e-2, 10.0):
        R = onewton.jacobian(lay, obj, im, beta, prob["kappa"])
        xd = onewton.kkt_solve_lu(R, r)
        x0 = onewton.kkt_solve_condensed(lay, obj, im, beta, prob["kappa"], r)
        scale = max(1.0, np.abs(xd).max())
        for split in [None] + list(range(2, H - 3)):
            x1 = onewton.kkt_solve_condensed_twisted_device(lay, obj, im, beta, prob["kappa"], r, split=split)
            assert np.abs(x1 - x0).max() <= 1e-9 * scale, (beta, split)
            assert np.abs(x1 - xd).max() <= 1e-10 * np.linalg.cond(R) * scale, (beta, split)


@pytest.mark.parametrize("N,w,RB", [(300, 37, 8), (256, 27, 8), (540, 107, 8), (411, 53, 4)])
def test_twisted_banded_factorisation(N, w, RB):
    """The banded L D L^T as two chains (oracle/banded.py: twisted_chain_bulk_ldl_solve = the CPU statement of kkt_banded_twisted_kernel:
    reversed matrix with decoupled dummy rows, zero middle block, trace handed to the top chain at a block boundary, given middle
    values in the bottom chain's back substitution) equals the dense solve and the one-chain form on random symmetric quasi-definite
    band matrices; the split lands on block boundaries."""
    from oracle import banded
    rng = np.random.default_rng(N + w)
    A = np.zeros((N, N))
    for i in range(N):
        for j in range(max(0, i - w), i):
            A[i, j] = A[j, i] = 0.05 * rng.standard_normal()
    sign = np.where(rng.random(N) < 0.5, 1.0, -1.0)
    A += np.diag(sign * (3.0 + np.abs(A).sum(axis=1)))
    b = rng.standard_normal(N)
    pad, Nb, m2 = banded.twisted_split(N, w, RB)
    assert m2 % RB == 0 and Nb % RB == 0 and 0 <= pad < RB and m2 + w + (Nb - pad) == N
    x0 = np.linalg.solve(A, b)
    x1, _ = banded.chain_bulk_ldl_solve(A, b, w, RB)
    x2 = banded.twisted_chain_bulk_ldl_solve(A, b, w, RB)
    np.testing.assert_allclose(x1, x0, rtol=0, atol=1e-13 * max(1.0, np.abs(x0).max()))
    np.testing.assert_allclose(x2, x0, rtol=0, atol=1e-13 * max(1.0, np.abs(x0).max()))
    assert banded.twisted_split(2160, 107, 8) == (3, 960, 1096)           # centroidal H = 60: the device's split (kkt_dense.hip: banded_twisted_split)
