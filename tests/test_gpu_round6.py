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


# ---- lazy commit of the accepted sensitivities (NewtonDev::good_src) --------------------------------------------------------------
@pytest.mark.parametrize("B,H,H_ref,ip_iter", [(1, 28, 30, 100), (3, 12, 14, 6), (96, 24, 26, 7), (160, 24, 26, 7)])
def test_lazy_sensitivity_commit_is_bit_identical_to_the_copy(monkeypatch, B, H, H_ref, ip_iter):
    """Round 6: an accept records WHICH slot holds each step's sensitivities and the next KKT stage moves the blocks (or the flush
    at the end of the solve) instead of the decision kernel copying 105 KB per accepting rollout.  The numbers must not change at
    all: same solves, same stale-sensitivity rule (a failed interior-point solve falls back to the block of the knot's last
    successful one - forced here with a budget of 6-7 interior-point iterations), on one rollout, a small batch, the twisted /
    overlapped rounds (B = 96) and the hybrid schedule (B = 160); then a warm-started solve and the B1 / B3 seams on the state
    the solve left (dz_good must be complete again after the flush)."""
    from contactimplicitmpc.jl_amd import InteriorPointOptions, NewtonOptions
    d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=H_ref, H=H, B=B, seed=9, perturb=3e-2)
    obj = synth.make_objective(d, H)
    q0 = np.stack([r[2] for r in rollouts]); q1 = np.stack([r[3] for r in rollouts])
    res = {}
    for lazy in (0, 1):
        monkeypatch.setenv("CIMPC_LAZY_DZ", str(lazy))
        s = make_solver(d, prob, rollouts, H, obj=obj, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"], max_iter=ip_iter),
                        newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-7, max_iter=4))
        a = s.newton_solve(q0, q1)
        ta = s.trajectory()
        fails = int(s.rollout_counters()["ip_failures"].sum())
        b = s.newton_solve(q1, ta["q"][:, 2].copy(), warm_start=True)
        tb = s.trajectory()
        r = np.random.default_rng(1).standard_normal((B, s.N))
        k = s.kkt_solve(r, 10.0) if False else None        # (B1 needs B3 on the handle first: below)
        out = s.implicit_dynamics(tb["q"], np.stack([ro.theta for (_, ro, _, _) in rollouts]))
        k = s.kkt_solve(r, 10.0)
        c = s.newton_solve(q0, q1, warm_start=True)         # B4 after B3: dz_good <- slot 0 hand-over, then the lazy path again
        res[lazy] = (a, ta, b, tb, out, k, c, fails)
        s.close()
    (a0, ta0, b0, tb0, o0, k0, c0, f0), (a1, ta1, b1, tb1, o1, k1, c1, f1) = res[0], res[1]
    if ip_iter < 100:
        assert f0 > 0, "no failing interior-point solve: the stale-sensitivity rule is not exercised"
    for x, y in ((a0, a1), (b0, b1), (c0, c1)):
        assert np.array_equal(x[1], y[1]) and np.array_equal(x[0], y[0]) and np.array_equal(x[2], y[2])
    for f in ("q", "u", "nu"):
        assert np.array_equal(ta0[f], ta1[f]) and np.array_equal(tb0[f], tb1[f])
    assert np.array_equal(k0, k1) and np.array_equal(o0["dq0"], o1["dq0"]) and f0 == f1


# ---- duo KKT kernel: the two-ended condensed solve as two one-wave chains in one workgroup (kkt_kernel_duo) ------------------------
def _duo_solve(monkeypatch, duo, d, prob, rollouts, obj, H, r, betas, nb=None):
    monkeypatch.setenv("CIMPC_KKT_TWISTED", "1" if duo else "0")
    monkeypatch.setenv("CIMPC_KKT_DUO", "2" if duo else "0")      # 2: the duo kernel at the B1 seam too
    if nb is not None:
        monkeypatch.setenv("CIMPC_KKT_TW_NB", str(nb))
    else:
        monkeypatch.delenv("CIMPC_KKT_TW_NB", raising=False)
    s = make_solver(d, prob, rollouts, H, obj=obj)
    q = np.stack([ro.q for (_, ro, _, _) in rollouts]); th = np.stack([ro.theta for (_, ro, _, _) in rollouts])
    out = s.implicit_dynamics(q, th)
    deltas = {beta: s.kkt_solve(r, beta) for beta in betas}
    n_tw = s.kkt_twisted()
    s.close()
    return out, deltas, n_tw


@pytest.mark.parametrize("model,H,H_ref,B", [("quadruped", 40, 60, 5), ("hopper", 24, 30, 3), ("flamingo", 30, 36, 2), ("particle", 26, 30, 2)])
def test_duo_kkt_vs_dense_lu(monkeypatch, model, H, H_ref, B):
    """B1 seam through the duo kernel (two one-wave chains in one workgroup, the recurrences of
    /root/reference/src/controller/newton_structure_solver/methods.jl:466-557 from both ends) against numpy's dense LU of the
    oracle's `jacobian!` matrix: 1e-10 of the solution's scale at beta = 10, the one-ended kernel's own level at the cold start's
    beta = 1e-5 (cond > 1e8)."""
    from oracle import newton as onewton
    d, prob, rollouts, obj = _tw_case(model, H, H_ref, B)
    lay = onewton.Layout(d, H)
    r = np.random.default_rng(0).standard_normal((B, lay.N))
    betas = (1e-5, 1e-2, 10.0)
    out1, one, n1 = _duo_solve(monkeypatch, False, d, prob, rollouts, obj, H, r, betas)
    out2, two, n2 = _duo_solve(monkeypatch, True, d, prob, rollouts, obj, H, r, betas)
    assert n1 == 0 and n2 == len(betas)
    for beta in betas:
        for b in range(B):
            im = {k: out2[k][b] for k in ("d", "dq0", "dq1", "du1")}
            x = np.linalg.solve(onewton.jacobian(lay, obj, im, beta, prob["kappa"]), r[b])
            sc = max(1.0, np.abs(x).max())
            assert np.isfinite(two[beta][b]).all()
            e1 = np.abs(one[beta][b] - x).max() / sc
            e2 = np.abs(two[beta][b] - x).max() / sc
            assert e2 <= (1e-10 if beta >= 1.0 else 1e-7), (beta, b, e2)
            assert e2 <= max(4.0 * e1, 1e-12), (beta, b, e1, e2)


def test_duo_kkt_every_split(monkeypatch):
    from oracle import newton as onewton
    H, H_ref, B = 26, 30, 2
    d, prob, rollouts, obj = _tw_case("quadruped", H, H_ref, B)
    lay = onewton.Layout(d, H)
    r = np.random.default_rng(1).standard_normal((B, lay.N))
    for nb in list(range(2, 8)) + list(range(H - 9, H - 3)):
        out, two, n = _duo_solve(monkeypatch, True, d, prob, rollouts, obj, H, r, (10.0,), nb=nb)
        assert n == 1
        for b in range(B):
            im = {k: out[k][b] for k in ("d", "dq0", "dq1", "du1")}
            x = np.linalg.solve(onewton.jacobian(lay, obj, im, 10.0, prob["kappa"]), r[b])
            np.testing.assert_allclose(two[10.0][b], x, rtol=0, atol=1e-10 * max(1.0, np.abs(x).max()), err_msg=f"nb = {nb}")


@pytest.mark.parametrize("B", [96, 200])
def test_newton_solve_duo_vs_packed_in_overlapped_rounds(monkeypatch, B):
    """newton_solve! of a batch whose rounds run the KKT stage next to the sweep (B >= 64, lock-step / hybrid): duo kernel against
    the packed one-wave kernel - same Newton iterations on (nearly) every rollout, controls equal where the discrete paths agree."""
    from contactimplicitmpc.jl_amd import NewtonOptions
    H, H_ref = 28, 30
    d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=H_ref, H=H, B=B, seed=13, perturb=5e-3)
    obj = synth.make_objective(d, H)
    q0 = np.stack([r[2] for r in rollouts]); q1 = np.stack([r[3] for r in rollouts])
    res = {}
    for duo in (0, 1):
        monkeypatch.setenv("CIMPC_KKT_DUO", str(duo))
        monkeypatch.setenv("CIMPC_KKT_TWISTED", "2" if duo else "0")      # (2: the three-wave twisted kernel stays out of the overlapped rounds)
        s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-5, max_iter=4))
        u1, it, rn = s.newton_solve(q0, q1)
        cnt = s.rollout_counters()
        res[duo] = (u1, it, rn, cnt, s.kkt_twisted())
        s.close()
    a, b = res[0], res[1]
    assert a[4] == 0 and b[4] > 0, (a[4], b[4])
    assert np.isfinite(b[0]).all()
    # Two KKT solves 1e-12 apart: a cold solve of this synthetic batch is hundreds of interior-point solves deep and most rollouts
    # leave the other run's discrete path somewhere (the ORACLE leaves its own path on 90 % of such rollouts
    # under 1-ulp input noise).  On the same path the controls agree; off it both runs must be solutions of the same quality, and
    # the batch statistics must not move.
    same = close_it = close_r = 0
    for k in range(B):
        if a[1][k] == b[1][k] and a[3]["sweeps"][k] == b[3]["sweeps"][k] and a[3]["ip_iters"][k] == b[3]["ip_iters"][k]:
            same += 1
            np.testing.assert_allclose(b[0][k], a[0][k], rtol=0, atol=5e-4 * max(1.0, np.abs(a[0][k]).max()))
        close_it += int(abs(int(a[1][k]) - int(b[1][k])) <= 1)
        close_r += int(b[2][k] <= 4.0 * a[2][k] + 1e-9 and a[2][k] <= 4.0 * b[2][k] + 1e-9)
    assert same >= 0.2 * B, (same, B)
    assert close_it >= 0.95 * B and close_r >= 0.9 * B, (close_it, close_r, B)
    assert abs(int((a[2] < 1e-5).sum()) - int((b[2] < 1e-5).sum())) <= max(2, B // 10)
    assert abs(float(a[3]["sweeps"].mean()) - float(b[3]["sweeps"].mean())) <= 0.05 * float(a[3]["sweeps"].mean())
    assert abs(np.median(a[2]) / np.median(b[2]) - 1.0) < 0.25


# ---- compiled sweep for 32 < ny <= 64 (centroidal_quadruped_wall, ny = 48): 64-lane groups, one problem per wavefront --------------
def test_centroidal_wall_runs_the_compiled_64_lane_sweep():
    """/root/reference/src/dynamics/centroidal_quadruped_wall/model.jl:204-207 (nc = 8: ny = 2 nc + nb = 48) ran on the runtime-dimension
    kernel through round 5.  :configuration mode now has a compiled instantiation (ip_model_centroidal_wall.hip: the lane-group
    solver on 64-lane groups, R packed in LDS, the iteration's operators staged, the once-per-solve blocks read from L2) - seen
    here by its table layout (the adjoint pass's constants are part of it) - and :configurationforce keeps the runtime-dimension
    kernel.  Both against the oracle: implicit_dynamics! on every knot."""
    from contactimplicitmpc.jl_amd import InteriorPointOptions
    from common import oracle_sweep
    from oracle import ip as oip
    H, H_ref, B = 6, 8, 3
    for mode in (0, 1):
        d, prob, tabs, rollouts = make_case("centroidal_wall", mode, H_ref=H_ref, H=H, B=B, seed=4, perturb=1e-2)
        s = make_solver(d, prob, rollouts, H, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]))
        tab_doubles, _ = s.query_sizes()
        nx, ny, nth, nths, G = 18, 48, 53, 48, 64
        generic = ny * G + 2 * nx * G + 2 * ny * G + 2 * nx * G + 2 * nth * G      # W .. RthRst at lane stride 64, no Gs / K0 / AiB
        assert (tab_doubles > generic + nths * G) == (mode == 0), (mode, tab_doubles)
        ref = oracle_sweep(d, tabs, rollouts, oip.IPOptions(kappa_tol=prob["kappa"]))
        out = s.implicit_dynamics(np.stack([t.q for t, _ in ref]), np.stack([t.theta for t, _ in ref]),
                                  gamma=np.stack([t.gamma for t, _ in ref]) if mode else None, b=np.stack([t.b for t, _ in ref]) if mode else None)
        n = agree = loose = 0
        for b, (_, o) in enumerate(ref):
            for i in range(H):
                n += 1
                if out["iters"][b, i] == o["iters"][i] and out["status"][b, i] == o["status"][i]:
                    agree += 1
                    if o["status"][i] == 1:      # (a failed solve stops at an iterate that round-off moves: converged solves only)
                        # (the configuration rows; in :configurationforce mode d also holds y1 - [gamma; b], which a converged solve fixes
                        #  only up to the complementarity tolerance kappa_tol - DESIGN.md section 2)
                        np.testing.assert_allclose(out["d"][b, i][:d.nq], o["d"][i][:d.nq], rtol=0, atol=1e-5)
                        np.testing.assert_allclose(out["d"][b, i], o["d"][i], rtol=0, atol=5.0 * prob["kappa"])
                        # sensitivities: 1e-4 of the block's scale - except on an ill-conditioned knot (this case has one: 15 iterations, the
                        # iterate moves by 1e-6 and the sensitivities by 1e-2 when the summation order of a dot product changes, as the
                        # oracle's own do under last-place noise): bounded at 5e-2, at most a tenth of the solves
                        e = max(np.abs(out[k][b, i][:d.nq] - o[k][i][:d.nq]).max() / max(1.0, np.abs(o[k][i]).max()) for k in ("dq0", "dq1", "du1"))
                        assert e <= 5e-2, (mode, b, i, e)
                        loose += e > 1e-4
        assert agree >= 0.9 * n, (mode, agree, n)
        assert loose <= max(2, 0.1 * n), (mode, loose, n)      # (the knot is solved by two of the three rollouts)
        s.close()


# ---- KKT stage of the persistent kernel as two cooperating jobs (the chains of the twisted solve, newton_async_impl.h) -----------------
@pytest.mark.parametrize("model,H,H_ref,B,sched", [("quadruped", 40, 44, 8, "single"), ("quadruped", 28, 30, 24, "single"), ("hopper", 26, 30, 6, "single"),
                                                   ("quadruped", 32, 36, 96, "hybrid"), ("quadruped", 32, 36, 48, "hybrid_default")])
def test_async_twisted_kkt_jobs_vs_the_one_ended_job(monkeypatch, model, H, H_ref, B, sched):
    """`newton_solve!` (/root/reference/src/controller/newton.jl:169-288) in ONE persistent launch (and in the hybrid schedule's tail) with
    every KKT stage run as two cooperating jobs - bottom chain pushed first, FIFO claims - against the same solve with the one-ended
    three-wave job: identical Newton iteration counts, controls as close as two KKT solves 1e-12 apart leave them (the synthetic batch
    is chaotic: compared where the discrete paths agree), no hand-over timed out."""
    from contactimplicitmpc.jl_amd import NewtonOptions
    d, prob, tabs, rollouts = make_case(model, 0, H_ref=H_ref, H=H, B=B, seed=7, perturb=1e-2)
    obj = synth.make_objective(d, H, kind=model)
    q0 = np.stack([r[2] for r in rollouts]); q1 = np.stack([r[3] for r in rollouts])
    res = {}
    # ("hybrid_default": the library's own choice - 32 < B < 256 hands over at max(32, B / 4) active rollouts and takes the two-job stage
    #  in tails of at most 32 - against the one-ended job under the same schedule)
    on = 1 if sched == "hybrid_default" else 2
    for tw in (0, on):
        monkeypatch.setenv("CIMPC_ASYNC_KKT_TW", str(tw))
        if sched == "hybrid":
            monkeypatch.setenv("CIMPC_ASYNC_TAIL", "64")
        s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-6, max_iter=4))
        u1, it, rn = s.newton_solve(q0, q1)
        cnt = s.rollout_counters()
        u1w, itw, rnw = s.newton_solve(q0, q1, warm_start=True)
        res[tw] = (u1, it, rn, u1w, itw, cnt, s.kkt_twisted(), s.kkt_twisted_fallbacks())
        s.close()
    a, b = res[0], res[on]
    assert b[6] > a[6], "the twisted jobs did not run"
    assert b[7] == 0
    assert np.isfinite(b[0]).all() and np.isfinite(b[3]).all()
    # (same bounds as the lock-step form of this comparison, test_gpu_round5.py::test_newton_solve_twisted_vs_one_ended: on the same
    #  discrete path the controls agree to 1e-5 of their scale - the conditioning of the Newton system at its residual floor -, off it
    #  both runs are solutions of the same quality)
    same = (a[1] == b[1]) & (a[5]["sweeps"] == b[5]["sweeps"]) & (a[5]["ip_iters"] == b[5]["ip_iters"])
    assert np.abs(a[1].astype(int) - b[1].astype(int)).max() <= 1
    assert same.mean() >= 0.25, (a[1], b[1])      # (H = 40: a cold solve is ~250 interior-point solves deep, half the rollouts flip one iteration count)
    for k in range(B):
        tol = ((1e-5 if H < 40 else 1e-4) if model == "quadruped" else 2e-3) if same[k] else 2e-3      # (H = 40: 4e-5 seen on equal counters)
        np.testing.assert_allclose(b[0][k], a[0][k], rtol=0, atol=tol * max(1.0, np.abs(a[0][k]).max()))
    assert np.abs(a[4] - b[4]).max() <= 1


def test_async_twisted_kkt_job_timeout_is_served_one_ended(monkeypatch):
    """Forced time-out of the hand-overs inside the persistent kernel (a bound of ONE poll): the chain that finishes last queues the
    rollout's KKT stage again as a one-ended job - finite results on the one-ended solve's discrete path, the fallbacks counted; the
    next solve on the same handle (default bound) runs the twisted jobs again and is right."""
    from contactimplicitmpc.jl_amd import NewtonOptions
    H, H_ref, B = 28, 30, 8
    d, prob, rollouts, obj = _tw_case("quadruped", H, H_ref, B)
    opts = NewtonOptions(kappa=prob["kappa"], r_tol=1e-6, max_iter=3)
    monkeypatch.setenv("CIMPC_ASYNC_KKT_TW", "0")
    s0 = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=opts)
    ref_cold = _newton(s0, rollouts)
    ref_warm = _newton(s0, rollouts, warm=True)
    s0.close()
    monkeypatch.setenv("CIMPC_ASYNC_KKT_TW", "2")
    s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=opts)
    s.debug_set_tw_spins(1)
    got = _newton(s, rollouts)
    n_fb = s.kkt_twisted_fallbacks()
    assert n_fb > 0, "the forced time-out did not happen"
    assert np.isfinite(got[0]).all() and np.isfinite(got[2]).all()
    assert np.array_equal(got[1], ref_cold[1])
    np.testing.assert_allclose(got[0], ref_cold[0], rtol=0, atol=1e-7 * max(1.0, np.abs(ref_cold[0]).max()))
    s.debug_set_tw_spins(0)
    got2 = _newton(s, rollouts, warm=True)
    assert np.abs(got2[1] - ref_warm[1]).max() <= 1
    assert np.isfinite(got2[0]).all() and np.isfinite(got2[3]["q"]).all()
    s.close()


def test_twisted_solves_next_to_a_saturating_foreign_kernel(monkeypatch):
    """the twisted chains' hand-over relies on both workgroups of a rollout being resident.  Here the device is kept
    busy by a foreign workload on another stream (a queue of large fp64 matrix products from torch) while warm-started Newton solves of
    three rollouts run their KKT stages on the twisted kernel: every result must be finite and equal to the one-ended kernels' - whether
    the chains met in time or a hand-over timed out and the stage was repeated one-ended (counted, never NaN)."""
    import torch
    from contactimplicitmpc.jl_amd import NewtonOptions
    H, H_ref, B = 28, 30, 3
    d, prob, rollouts, obj = _tw_case("quadruped", H, H_ref, B)
    opts = NewtonOptions(kappa=prob["kappa"], r_tol=1e-6, max_iter=3)
    monkeypatch.setenv("CIMPC_KKT_TWISTED", "0")
    s0 = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=opts)
    ref = [_newton(s0, rollouts)] + [_newton(s0, rollouts, warm=True) for _ in range(3)]
    s0.close()
    monkeypatch.setenv("CIMPC_KKT_TWISTED", "1")
    s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=opts)
    a = torch.randn(6144, 6144, dtype=torch.float64, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(30):                     # ~0.2 s of queued foreign work: every CU busy while the solves run
            a = (a @ a) * 1e-4
    got = [_newton(s, rollouts)] + [_newton(s, rollouts, warm=True) for _ in range(3)]
    busy = not side.query()                     # (the foreign queue outlived the solves: they did run next to it)
    torch.cuda.synchronize()
    assert s.kkt_twisted() > 0
    for g, r in zip(got, ref):
        assert np.isfinite(g[0]).all() and np.isfinite(g[2]).all() and np.isfinite(g[3]["q"]).all()
        assert np.abs(g[1].astype(int) - r[1].astype(int)).max() <= 1
        np.testing.assert_allclose(g[0], r[0], rtol=0, atol=1e-5 * max(1.0, np.abs(r[0]).max()))
    assert busy, "the foreign workload finished before the solves did - enlarge it"
    s.close()
