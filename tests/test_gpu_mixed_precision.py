"""-m gpu: BASELINE configs[4] - the condensed KKT solve with its Schur-block products on the fp32 MFMA
(v_mfma_f32_16x16x4_f32), fp64 refinement against the matrix-free KKT operator and an fp64 fallback
(kkt_backend = CIMPC_KKT_CONDENSED_MIXED; newton_kernels.hip: launch_kkt_mixed).

Reference anchors: the condensed solve restates src/controller/newton_structure_solver/methods.jl:386-557; the guard for
ill-conditioned Schur blocks follows the situation test/solver/schur.jl:19-62 constructs; the configuration is
examples/centroidal_quadruped/continuous_trot.jl:37-73 (centroidal_quadruped, H_mpc = 50, kappa = 1e-3) with a payload
(a body-force disturbance w in theta: src/dynamics/centroidal_quadruped/model.jl:121-125, continuous_trot.jl:80-81)."""
import numpy as np
import pytest

from oracle import newton as onewton, synth

from common import make_case, make_solver
from real_problems import real_problem, real_rollout

pytestmark = pytest.mark.gpu
MIXED, FP64 = 3, 0


@pytest.mark.parametrize("model,H,H_ref", [("centroidal", 60, 71), ("quadruped", 40, 60), ("hopper", 20, 24)])
def test_mixed_kkt_solve_matches_fp64_and_dense(gpu_required, model, H, H_ref):
    """B1 seam: mixed-precision solve == fp64 condensed solve == numpy's dense LU of the assembled R, to 1e-7 relative."""
    from contactimplicitmpc.jl_amd import NewtonOptions
    B = 3
    d, prob, tabs, rollouts = make_case(model, 0, H_ref=H_ref, H=H, B=B, seed=5)
    obj = synth.make_objective(d, H, kind=model, dense_q=True)
    q = np.stack([r.q for (_, r, _, _) in rollouts]); th = np.stack([r.theta for (_, r, _, _) in rollouts])
    lay = onewton.Layout(d, H)
    r = np.random.default_rng(0).standard_normal((B, lay.N))
    sols = {}
    for backend in (MIXED, FP64):
        s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], kkt_backend=backend))
        out = s.implicit_dynamics(q, th)
        assert out["status"].all()
        sols[backend] = {}
        fb = {}
        for beta in (10.0, 1e-5):
            sols[backend][beta] = s.kkt_solve(r, beta)
            fb[beta] = s.kkt_fallbacks() if backend == MIXED else 0
        if backend == MIXED:
            fallbacks = fb
        s.close()
    for beta in (1e-5, 10.0):
        a, c = sols[MIXED][beta], sols[FP64][beta]
        for b in range(B):
            scale = max(1.0, np.abs(c[b]).max())
            assert np.isfinite(a[b]).all()
            np.testing.assert_allclose(a[b], c[b], rtol=0, atol=1e-7 * scale)
            im = {k: out[k][b] for k in ("d", "dq0", "dq1", "du1")}
            R = onewton.jacobian(lay, obj, im, beta, prob["kappa"])
            # the refinement criterion itself: |r - R x|_inf <= 1e-10 max(1, |r|_inf), whichever precision got there
            assert np.abs(R @ a[b] - r[b]).max() <= 2e-9 * max(1.0, np.abs(r[b]).max())
            np.testing.assert_allclose(a[b], np.linalg.solve(R, r[b]), rtol=0, atol=1e-7 * scale)
    # beta = 10 (every Newton iteration after the first: newton.jl:280): rho = H beta kappa ~ 0.1, no system needs the fp64
    # solve.  beta = beta_init = 1e-5 (cold start): rho ~ 1e-7, the dual Schur complement is beyond fp32 - fallback allowed.
    assert fallbacks[10.0] == 0, fallbacks


def test_mixed_kkt_falls_back_on_ill_conditioned_blocks(gpu_required):
    """Objective weights spread over 12 orders of magnitude: the fp32 Schur blocks lose their digits (or a pivot), the
    refinement stalls and the fp64 solve takes over - the answer is still the fp64 one."""
    from contactimplicitmpc.jl_amd import NewtonOptions
    H, H_ref, B = 12, 16, 3
    d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=H_ref, H=H, B=B, seed=7)
    obj = synth.make_objective(d, H, kind="quadruped")
    w = np.logspace(-9, 3, d.nq)
    obj.q = np.tile(np.diag(w)[None], (H, 1, 1))
    q = np.stack([r.q for (_, r, _, _) in rollouts]); th = np.stack([r.theta for (_, r, _, _) in rollouts])
    lay = onewton.Layout(d, H)
    r = np.random.default_rng(1).standard_normal((B, lay.N))
    sols = {}
    for backend in (MIXED, FP64):
        s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], kkt_backend=backend))
        s.implicit_dynamics(q, th)
        sols[backend] = s.kkt_solve(r, 1e-5)
        if backend == MIXED:
            fallbacks = s.kkt_fallbacks()
        s.close()
    assert fallbacks >= 1, "expected the guard to trip on these blocks"
    for b in range(B):
        np.testing.assert_allclose(sols[MIXED][b], sols[FP64][b], rtol=0, atol=1e-7 * max(1.0, np.abs(sols[FP64][b]).max()))


def test_centroidal_payload_newton_solve_mixed_vs_fp64(gpu_required):
    """newton_solve! on the real centroidal problem (inplace_trot_v7.jld2, continuous_trot.jl settings) with a payload
    disturbance w != 0 on every rollout: the mixed-precision backend reproduces the fp64 one (same Newton iterations and
    sweeps, controls to 1e-7 relative)."""
    from contactimplicitmpc.jl_amd import InteriorPointOptions, NewtonOptions, lcp_models
    kappa, H, B = 1e-3, 50, 6
    d, P, prob, tabs = real_problem("centroidal", kappa)
    rng = np.random.default_rng(4)
    rollouts = []
    for b in range(B):
        window, ref, q0, q1 = real_rollout(d, prob, H, int(rng.integers(0, P.H)), seed=10 + b, perturb=0.02)
        ref.w[:] = np.array([0.0, 0.0, -rng.uniform(5.0, 30.0)]) * (b > 0)      # payload: a constant downward body force (rollout 0: none)
        ref.update_theta(d)
        rollouts.append((window, ref, q0, q1))
    obj = synth.make_objective(d, H, kind="quadruped")
    obj.q = np.tile(lcp_models.relative_state_cost([1.0, 1, 1], 3e-1 * np.ones(3), [0.2, 0.2, 1.0])[None], (H, 1, 1))
    obj.u = np.tile((3e-3 * np.eye(d.nu))[None], (H, 1, 1))
    obj.__post_init__()
    res = {}
    for backend in (MIXED, FP64):
        s = make_solver(d, prob, rollouts, H, obj=obj, ip_opts=InteriorPointOptions(kappa_tol=kappa, r_tol=1e-4),
                        newton_opts=NewtonOptions(kappa=kappa, r_tol=3e-5, max_iter=5, kkt_backend=backend))
        u1, it, rn = s.newton_solve(np.stack([r[2] for r in rollouts]), np.stack([r[3] for r in rollouts]))
        res[backend] = (u1, it, rn, s.trajectory(), s.rollout_counters(), s.kkt_fallbacks() if backend == MIXED else 0)
        s.close()
    a, c = res[MIXED], res[FP64]
    assert c[1].max() >= 2                                           # real Newton work
    np.testing.assert_array_equal(a[1], c[1])
    np.testing.assert_array_equal(a[4]["sweeps"], c[4]["sweeps"])
    same = a[4]["ip_iters"] == c[4]["ip_iters"]
    assert same.sum() >= B - 1                                       # (a 1e-10 difference of a direction may flip one interior-point count)
    scale = max(1.0, np.abs(c[0]).max())
    np.testing.assert_allclose(a[0][same], c[0][same], rtol=0, atol=1e-7 * scale)
    np.testing.assert_allclose(a[3]["q"][same], c[3]["q"][same], rtol=0, atol=1e-7)
    # the payload is live: loaded rollouts answer with different controls than the unloaded one would
    assert np.abs(rollouts[1][1].theta[:, d.iw1]).max() > 1.0
    assert a[5] <= 0.2 * int(c[1].sum())                             # at most a fifth of the systems needed the fp64 fallback
