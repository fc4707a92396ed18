"""Round 6 device tests.

* `cimpc_mpc_solve` (one C call per MPC step: window, reference, altitude, q0, q1 in - u1 / iterations / residual and the
  trajectory out) against the five-call sequence it replaces (cimpc_set_altitude, cimpc_set_window, cimpc_set_reference,
  cimpc_newton_solve, cimpc_get_trajectory): bit-identical results over a warm-started loop whose window moves every step - the
  call sequence of the drop-in under the reference's unchanged policy (/root/reference/src/controller/policy.jl:113-142).
"""
import numpy as np
import pytest

from oracle import synth

from common import make_case, make_solver

pytestmark = pytest.mark.gpu


def _shifted(d, prob, rollouts, H, k):
    """Rollout inputs k MPC steps later: phase + k windows / references of the same problem (new objects, same seeds)."""
    out = []
    H_ref = prob["z0"].shape[0]
    for b, (window, ref, q0, q1) in enumerate(rollouts):
        ph = (int(window[0]) + k) % H_ref
        w2, r2, _, _ = synth.make_rollout(d, prob, H, phase=ph, seed=1000 + b, perturb=0.0)
        out.append((w2, r2))
    return out


@pytest.mark.parametrize("model,mode,B", [("quadruped", 0, 3), ("hopper", 0, 1), ("pushbot", 1, 2)])
def test_mpc_solve_equals_the_five_call_sequence(model, mode, B):
    H, H_ref = 8, 12
    d, prob, tabs, rollouts = make_case(model, mode, H_ref=H_ref, H=H, B=B, seed=3, perturb=1e-2)
    obj = synth.make_objective(d, H)
    sa = make_solver(d, prob, rollouts, H, obj=obj)
    sb = make_solver(d, prob, rollouts, H, obj=obj)
    q0 = np.stack([r[2] for r in rollouts]); q1 = np.stack([r[3] for r in rollouts])
    rng = np.random.default_rng(0)
    for k in range(4):
        sh = _shifted(d, prob, rollouts, H, k)
        win = np.stack([w for (w, _) in sh]) + 1
        ref = {f: np.stack([getattr(r, f) for (_, r) in sh]) for f in ("q", "u", "w", "gamma", "b", "theta")}
        alt = 1e-3 * rng.standard_normal((B, d.nc)) if k >= 2 else None
        # (a) the five calls
        if alt is not None:
            sa.set_altitude(alt)
        sa.set_window(win)
        sa.set_reference(ref["q"], ref["u"], ref["w"], ref["gamma"], ref["b"], ref["theta"])
        u1a, ita, rna = sa.newton_solve(q0, q1, warm_start=k > 0)
        ta = sa.trajectory()
        # (b) one call
        u1b, itb, rnb, tb = sb.mpc_solve(q0, q1, window=win, reference=ref, alt=alt, warm_start=k > 0, which=("q", "u", "gamma", "b", "nu"))
        assert np.array_equal(ita, itb) and np.array_equal(u1a, u1b) and np.array_equal(rna, rnb)
        for f in ("q", "u", "gamma", "b", "nu"):
            assert np.array_equal(ta[f], tb[f]), f
        q0, q1 = q1, ta["q"][:, 2].copy()
    # unchanged window / reference: NULL inputs keep what is resident
    u1a, ita, rna = sa.newton_solve(q0, q1, warm_start=True)
    u1b, itb, rnb, _ = sb.mpc_solve(q0, q1, warm_start=True, which=())
    assert np.array_equal(ita, itb) and np.array_equal(u1a, u1b) and np.array_equal(rna, rnb)
    sa.close(); sb.close()


def test_mpc_solve_validates_its_window():
    from contactimplicitmpc.jl_amd import CimpcError
    H, H_ref = 6, 8
    d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=H_ref, H=H, B=2, seed=1)
    s = make_solver(d, prob, rollouts, H, obj=synth.make_objective(d, H))
    q0 = np.stack([r[2] for r in rollouts]); q1 = np.stack([r[3] for r in rollouts])
    bad = np.stack([w for (w, _, _, _) in rollouts]) + 1
    bad[1, 3] = H_ref + 1
    with pytest.raises(CimpcError):
        s.mpc_solve(q0, q1, window=bad)
    s.close()


# ---- twisted hand-over: epoch-valued flags, time-out -> one-ended fallback ----------------------
def _tw_case(model, H, H_ref, B):
    d, prob, tabs, rollouts = make_case(model, 0, H_ref=H_ref, H=H, B=B, seed=5, perturb=2e-2)
    obj = synth.make_objective(d, H, dense_q=True)
    return d, prob, rollouts, obj


def _newton(s, rollouts, warm=False):
    u1, it, rn = s.newton_solve(np.stack([r[2] for r in rollouts]), np.stack([r[3] for r in rollouts]), warm_start=warm)
    return u1, it, rn, s.trajectory()


@pytest.mark.parametrize("B", [1, 3])
def test_twisted_timeout_falls_back_to_the_one_ended_kernel(monkeypatch, B):
    """A hand-over wait of the twisted KKT kernel that times out (forced: a bound of ONE poll) must not reach the caller as NaN:
    the rollout's KKT stage is queued again, the host serves it with the one-ended kernel and counts it.  Then, on the SAME handle
    with the default bound, the next solve runs the twisted kernel again and is right - a partner that raised its flag after the
    waiter gave up cannot be taken for this solve's (epoch-valued flags; round 5 kept 0/1 flags that the consumer reset)."""
    from contactimplicitmpc.jl_amd import NewtonOptions
    H, H_ref = 28, 30
    d, prob, rollouts, obj = _tw_case("quadruped", H, H_ref, B)
    opts = NewtonOptions(kappa=prob["kappa"], r_tol=1e-6, max_iter=3)
    monkeypatch.setenv("CIMPC_KKT_TWISTED", "0")
    s0 = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=opts)
    ref_cold = _newton(s0, rollouts)
    ref_warm = _newton(s0, rollouts, warm=True)
    assert s0.kkt_twisted() == 0
    s0.close()
    monkeypatch.setenv("CIMPC_KKT_TWISTED", "1")
    s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=opts)
    s.debug_set_tw_spins(1)
    got = _newton(s, rollouts)
    n_fb = s.kkt_twisted_fallbacks()
    assert n_fb > 0, "the forced time-out did not happen"
    assert np.isfinite(got[0]).all() and np.isfinite(got[2]).all()
    # the repeated stages ran on the one-ended kernel: the discrete path and the numbers are the one-ended handle's
    assert np.array_equal(got[1], ref_cold[1])
    np.testing.assert_allclose(got[0], ref_cold[0], rtol=0, atol=1e-9 * max(1.0, np.abs(ref_cold[0]).max()))
    np.testing.assert_allclose(got[3]["q"], ref_cold[3]["q"], rtol=0, atol=1e-9)
    # default bound again, same handle: the twisted kernel runs (and is not fooled by flags the aborted launches left behind)
    s.debug_set_tw_spins(0)
    n_tw = s.kkt_twisted()
    got2 = _newton(s, rollouts, warm=True)
    assert s.kkt_twisted() > n_tw and s.kkt_twisted_fallbacks() == n_fb
    assert np.array_equal(got2[1], ref_warm[1])
    np.testing.assert_allclose(got2[0], ref_warm[0], rtol=0, atol=1e-5 * max(1.0, np.abs(ref_warm[0]).max()))
    assert np.isfinite(got2[3]["q"]).all()
    s.close()


def test_twisted_timeout_at_the_b1_seam(monkeypatch):
    """cimpc_kkt_solve (B1) with the twisted kernel and a forced time-out: repeated at once on the one-ended kernel."""
    H, H_ref, B = 28, 30, 2
    d, prob, rollouts, obj = _tw_case("quadruped", H, H_ref, B)
    rng = np.random.default_rng(3)
    q = np.stack([ro.q for (_, ro, _, _) in rollouts]); th = np.stack([ro.theta for (_, ro, _, _) in rollouts])
    res = {}
    for tw in (0, 1):
        monkeypatch.setenv("CIMPC_KKT_TWISTED", str(tw))
        s = make_solver(d, prob, rollouts, H, obj=obj)
        s.implicit_dynamics(q, th)
        r = np.random.default_rng(3).standard_normal((B, s.N))
        if tw:
            s.debug_set_tw_spins(1)
        res[tw] = s.kkt_solve(r, 10.0)
        if tw:
            assert s.kkt_twisted_fallbacks() > 0
            s.debug_set_tw_spins(0)
            again = s.kkt_solve(r, 10.0)      # the twisted kernel itself, after aborted launches on the same flags
            assert np.isfinite(again).all()
            np.testing.assert_allclose(again, res[0], rtol=0, atol=1e-9 * np.abs(res[0]).max())
        s.close()
    assert np.array_equal(res[0], res[1])


def test_twisted_banded_timeout_falls_back(monkeypatch):
    """The same for the twisted banded LDL^T (velocity objective): time-out -> the one-chain kernel."""
    from contactimplicitmpc.jl_amd import NewtonOptions
    H, H_ref, B = 24, 26, 2
    d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=H_ref, H=H, B=B, seed=7, perturb=1e-2)
    obj = synth.make_objective(d, H, velocity=True)
    opts = NewtonOptions(kappa=prob["kappa"], r_tol=1e-6, max_iter=2)
    out = {}
    for tw in (0, 1):
        monkeypatch.setenv("CIMPC_KKT_TWISTED", str(tw))
        s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=opts)
        if tw:
            if s.kkt_twisted() == 0 and False:
                pass
            s.debug_set_tw_spins(1)
        out[tw] = _newton(s, rollouts)
        if tw:
            if s.kkt_twisted() == 0:
                s.close()
                pytest.skip("the twisted banded kernel is not taken at this size")
            assert s.kkt_twisted_fallbacks() > 0
        s.close()
    assert np.array_equal(out[0][1], out[1][1])
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=0, atol=1e-9 * max(1.0, np.abs(out[0][0]).max()))
