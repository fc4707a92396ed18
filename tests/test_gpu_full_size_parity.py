"""-m gpu: oracle parity at BASELINE.json's FULL size, on bench.py's exact inputs.

`bench.build_inputs(512, 40, 60, 1234, 0.05)` = the 512 quadruped rollouts of the headline bench line.  They go
  (a) through the B3 seam (`implicit_dynamics!`, 20 480 interior-point solves) and
  (b) through the B4 seam (`newton_solve!`, cold start, the timed step of bench.py)
on the device and through the single-thread C restatement `oracle/cimpc_ref.c` (all 512 rollouts, ~10 s).
The measured agreement rates are printed and written to `gpurun_out/parity_full_size.json` (committed copy:
`profiles/r02/parity_full_size.json`), and bounded below by the asserts.

Arbiter for the solves whose discrete outcome (status, iteration count) differs: the SAME oracle is re-run on
inputs perturbed by one unit in the last place (theta * (1 +- 2^-52), K draws).  A solve whose oracle outcome
changes under such a perturbation sits on a round-off decision boundary; the test requires that the device/oracle
disagreements concentrate on those solves (and records the contingency table) - a disagreement on a solve that is
insensitive to last-place perturbations would be a divergence, not round-off.
"""
import json
import os
import time

import numpy as np
import pytest

from oracle import ip as oip
from oracle import newton as onewton
from oracle.cref import CRef

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, H_REF, B = 40, 60, 512
RECORD = {}


def _record(key, val):
    RECORD[key] = val
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_full_size.json"), "w") as f:
        json.dump(RECORD, f, indent=1, sort_keys=True)
    print("[parity_full_size]", key, json.dumps(val))


@pytest.fixture(scope="module")
def case():
    import bench
    d, prob, obj, ro = bench.build_inputs(B, H, H_REF, seed=1234, perturb=0.05)
    cr = CRef(d, H_REF, H, prob, obj, oip.IPOptions(kappa_tol=prob["kappa"]), onewton.NewtonOptions(r_tol=3e-4, max_iter=5), prob["kappa"])
    return d, prob, obj, ro, cr


def _solver(d, prob, obj, ro):
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
    s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_REF, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]),
                    newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-4, max_iter=5))
    for t in range(H_REF):
        s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
    s.set_objective(obj.q, obj.u)
    s.set_window(np.stack([w for (w, _, _, _) in ro]) + 1)
    s.set_reference(np.stack([r.q for (_, r, _, _) in ro]), np.stack([r.u for (_, r, _, _) in ro]), np.stack([r.w for (_, r, _, _) in ro]),
                    np.stack([r.gamma for (_, r, _, _) in ro]), np.stack([r.b for (_, r, _, _) in ro]), np.stack([r.theta for (_, r, _, _) in ro]))
    return s


def _first_sweep_inputs(d, ro):
    """The trajectory newton_solve! sweeps first: the reference with (q0, q1) substituted (newton.jl:130-167)."""
    qs, ths = [], []
    for (window, ref, q0, q1) in ro:
        tr = ref.copy()
        tr.q[0], tr.q[1] = q0, q1
        tr.update_theta(d, 0)
        tr.update_theta(d, 1)
        qs.append(tr.q); ths.append(tr.theta)
    return np.stack(qs), np.stack(ths)


def test_implicit_dynamics_full_size_vs_oracle(gpu_required, case):
    d, prob, obj, ro, cr = case
    q, th = _first_sweep_inputs(d, ro)
    s = _solver(d, prob, obj, ro)
    out = s.implicit_dynamics(q, th, want_z=True)
    s.close()
    t0 = time.perf_counter()
    ref = [cr.implicit_dynamics(ro[b][0], q[b], th[b]) for b in range(B)]
    t_oracle = time.perf_counter() - t0
    st = np.stack([r["status"] for r in ref]); it = np.stack([r["iters"] for r in ref])
    same = (out["status"] == st) & (out["iters"] == it)
    conv = same & (st == 1)
    dd = np.abs(out["d"] - np.stack([r["d"] for r in ref])).max(axis=2)
    # sensitivities, per solve: largest absolute deviation of the consumed block [dq0 | dq1 | du1] and its scale
    dev_raw = np.transpose(np.concatenate([out["dq0"], out["dq1"], out["du1"]], axis=3), (0, 1, 3, 2))      # (B, H, nths, nd) like dz_raw
    ora_raw = np.stack([r["dz_raw"] for r in ref])
    dz_abs = np.abs(dev_raw - ora_raw).max(axis=(2, 3))
    dz_scale = np.maximum(np.abs(ora_raw).max(axis=(2, 3)), 1.0)
    dz_err = float((dz_abs / dz_scale)[conv].max())
    zz = np.abs(out["z"] - np.stack([r["z"] for r in ref])).max(axis=2)
    # ---- arbiter: last-place input perturbations of the oracle itself -------------------------------------------
    K = 8
    rng = np.random.default_rng(99)
    sens = np.zeros((B, H), dtype=bool)
    sc_d = np.zeros((B, H)); sc_z = np.zeros((B, H)); sc_dz = np.zeros((B, H))
    o_d = np.stack([r["d"] for r in ref]); o_z = np.stack([r["z"] for r in ref]); o_dz = np.stack([r["dz_raw"] for r in ref])
    for k in range(K):
        for b in range(B):
            sg = rng.integers(0, 2, th[b].shape) * 2 - 1
            thp = th[b] * (1.0 + sg * 2.0 ** -52)
            r = cr.implicit_dynamics(ro[b][0], q[b], thp)
            keep = (r["status"] == st[b]) & (r["iters"] == it[b])
            sens[b] |= ~keep
            sc_d[b] = np.maximum(sc_d[b], np.where(keep, np.abs(r["d"] - o_d[b]).max(axis=1), 0.0))
            sc_z[b] = np.maximum(sc_z[b], np.where(keep, np.abs(r["z"] - o_z[b]).max(axis=1), 0.0))
            sc_dz[b] = np.maximum(sc_dz[b], np.where(keep, np.abs(r["dz_raw"] - o_dz[b]).max(axis=(1, 2)), 0.0))
    flip = ~same
    n = flip.size
    table = {"flipped_and_sensitive": int((flip & sens).sum()), "flipped_not_sensitive": int((flip & ~sens).sum()),
             "agree_and_sensitive": int((~flip & sens).sum()), "agree_not_sensitive": int((~flip & ~sens).sum())}
    rate_in = table["flipped_and_sensitive"] / max(int(flip.sum()), 1)
    base_rate = float(sens.mean())
    qs = lambda a: {"median": float(np.quantile(a, 0.5)), "q90": float(np.quantile(a, 0.9)), "q95": float(np.quantile(a, 0.95)), "q99": float(np.quantile(a, 0.99)), "max": float(a.max())}
    # value agreement: nominal tolerance wherever the oracle itself is stable, 50 x its own last-place scatter elsewhere
    tol_d = np.maximum(1e-7, 50 * sc_d); tol_z = np.maximum(1e-6, 50 * sc_z)
    tol_dz = np.maximum(1e-6 * dz_scale, 50 * sc_dz)
    out_d = conv & (dd > 1e-7)
    out_dz = conv & (dz_abs > 1e-6 * dz_scale)
    _record("implicit_dynamics", {
        "solves": int(n), "identical_status_and_iters": int(same.sum()), "agreement_rate": float(same.mean()),
        "iters_diff_histogram": {str(k): int(v) for k, v in zip(*np.unique((out["iters"] - it)[flip], return_counts=True))},
        "status_differs": int((out["status"] != st).sum()),
        "abs_d_diff_on_agreeing_converged": qs(dd[conv]), "abs_z_diff_on_agreeing_converged": qs(zz[conv]),
        "oracle_own_d_scatter_under_1ulp": qs(sc_d[conv]), "oracle_own_z_scatter_under_1ulp": qs(sc_z[conv]),
        "solves_with_d_diff_above_1e-7": int(out_d.sum()), "of_those_within_50x_oracle_scatter": int((out_d & (dd <= tol_d)).sum()),
        "max_rel_dz_diff_on_agreeing_converged": dz_err,
        "rel_dz_diff_on_agreeing_converged": qs((dz_abs / dz_scale)[conv]), "oracle_own_rel_dz_scatter_under_1ulp": qs((sc_dz / dz_scale)[conv]),
        "solves_with_rel_dz_diff_above_1e-6": int(out_dz.sum()), "of_those_within_50x_oracle_dz_scatter": int((out_dz & (dz_abs <= tol_dz)).sum()),
        "worst_dz_solves": [{"b": int(b_), "i": int(i_), "rel_dz": float((dz_abs / dz_scale)[b_, i_]), "oracle_scatter_rel": float((sc_dz / dz_scale)[b_, i_]),
                             "iters": int(it[b_, i_]), "d_diff": float(dd[b_, i_])}
                            for b_, i_ in zip(*np.unravel_index(np.argsort(np.where(conv, dz_abs / dz_scale, 0.0), axis=None)[-5:], dz_abs.shape))],
        "max_abs_d_diff_on_flipped": float(dd[flip & (st == 1) & (out["status"] == 1)].max()) if flip.any() else 0.0,
        "oracle_seconds": t_oracle, "ulp_arbiter": dict(table, draws=K, sensitive_share_of_flipped=rate_in, sensitive_share_overall=base_rate)})
    assert same.mean() >= 0.999, same.mean()
    # the bulk at nominal fp64 tolerances (about 3 % of these solves are ill-conditioned enough that the ORACLE ITSELF moves
    # by 1e-7 .. 2e-5 in d under last-place input noise - recorded above as oracle_own_d_scatter_under_1ulp)
    assert (dd[conv] < 1e-7).mean() >= 0.95 and np.median(dd[conv]) < 1e-11 and (zz[conv] < 1e-6).mean() >= 0.9
    assert (dd[conv] <= tol_d[conv]).all(), int((dd[conv] > tol_d[conv]).sum())                # the rest: explained by the oracle's own scatter
    # sensitivities (they are the Jacobian data of the KKT stage): 1e-6 relative wherever the oracle itself is stable under
    # last-place input noise, inside 50 x its own scatter elsewhere - the same rule as d and z
    assert ((dz_abs / dz_scale)[conv] < 1e-6).mean() >= 0.9 and np.median((dz_abs / dz_scale)[conv]) < 1e-10
    assert (dz_abs[conv] <= tol_dz[conv]).all(), (int((dz_abs[conv] > tol_dz[conv]).sum()), RECORD["implicit_dynamics"]["worst_dz_solves"])
    assert (zz[conv] <= tol_z[conv]).mean() > 0.999
    # every flip is a one-iteration (or status-at-the-boundary) move of a solve that ALSO converged to the same point
    both = flip & (st == 1) & (out["status"] == 1)
    if both.any():
        assert np.abs((out["iters"] - it)[both]).max() <= 2
        assert dd[both].max() < 1e-3          # kappa_tol-level indeterminacy of a converged point
    # round-off, not divergence: disagreements sit on solves the oracle itself flips under 1-ulp input noise
    if flip.sum() >= 5:
        assert rate_in >= 0.6 and rate_in > 5 * base_rate, (rate_in, base_rate)
    else:
        assert table["flipped_not_sensitive"] <= 2, table


def test_newton_solve_full_size_vs_oracle(gpu_required, case):
    d, prob, obj, ro, cr = case
    s = _solver(d, prob, obj, ro)
    q0 = np.stack([r[2] for r in ro]); q1 = np.stack([r[3] for r in ro])
    u1, it, rn = s.newton_solve(q0, q1)
    traj = s.trajectory(); cnt = s.rollout_counters()
    s.close()
    t0 = time.perf_counter()
    ref = [cr.newton_solve(ro[b][0], ro[b][1], ro[b][2], ro[b][3], solver=1) for b in range(B)]
    t_oracle = time.perf_counter() - t0
    N = H * (d.nr + d.nd)
    o_it = np.array([r["iters"] for r in ref]); o_sw = np.array([r["sweeps"] for r in ref])
    o_ip = np.array([r["ip_iters"] for r in ref]); o_fail = np.array([r["ip_fail"] for r in ref])
    o_rn = np.array([r["r_norm"] for r in ref])            # (already |r|_1 / N)
    o_u1 = np.stack([r["u"][0] for r in ref])
    same_it = it == o_it
    same_path = same_it & (cnt["sweeps"] == o_sw) & (cnt["ip_iters"] == o_ip)
    du = np.abs(u1 - o_u1).max(axis=1)
    dq = np.array([np.abs(traj["q"][b] - ref[b]["q"]).max() for b in range(B)])
    # last-place perturbations of (q0, q1): how far does the oracle itself move?
    rng = np.random.default_rng(7)
    K = 3
    o_flip = np.zeros(B, dtype=bool); o_du = np.zeros(B)
    for k in range(K):
        for b in range(B):
            p0 = q0[b] * (1.0 + (rng.integers(0, 2, q0[b].shape) * 2 - 1) * 2.0 ** -52)
            p1 = q1[b] * (1.0 + (rng.integers(0, 2, q1[b].shape) * 2 - 1) * 2.0 ** -52)
            r = cr.newton_solve(ro[b][0], ro[b][1], p0, p1, solver=1)
            o_flip[b] |= (r["iters"] != o_it[b]) or (r["sweeps"] != o_sw[b]) or (r["ip_iters"] != o_ip[b])
            o_du[b] = max(o_du[b], np.abs(r["u"][0] - o_u1[b]).max())
    off = ~same_path
    qs = lambda a: {"median": float(np.quantile(a, 0.5)), "q90": float(np.quantile(a, 0.9)), "max": float(a.max())} if a.size else {}
    _record("newton_solve", {
        "rollouts": B, "same_newton_iters": int(same_it.sum()), "same_newton_iters_rate": float(same_it.mean()),
        "same_discrete_path (iters, sweeps, ip_iters)": int(same_path.sum()), "same_path_rate": float(same_path.mean()),
        "newton_iters_mean_device": float(it.mean()), "newton_iters_mean_oracle": float(o_it.mean()),
        "sweeps_mean_device": float(cnt["sweeps"].mean()), "sweeps_mean_oracle": float(o_sw.mean()),
        "ip_iters_mean_device": float(cnt["ip_iters"].mean()), "ip_iters_mean_oracle": float(o_ip.mean()),
        "ip_failures_device": int(cnt["ip_failures"].sum()), "ip_failures_oracle": int(o_fail.sum()),
        "u1_diff_same_path": qs(du[same_path]), "q_diff_same_path": qs(dq[same_path]), "u1_diff_off_path": qs(du[off]),
        "converged_device": int((rn < 3e-4).sum()), "converged_oracle": int((o_rn < 3e-4).sum()),
        "r_norm_median_device": float(np.median(rn)), "r_norm_median_oracle": float(np.median(o_rn)),
        "oracle_seconds": t_oracle,
        "ulp_arbiter": {"draws": K, "oracle_path_changes_under_1ulp": int(o_flip.sum()), "off_path_and_oracle_sensitive": int((off & o_flip).sum()),
                        "off_path_not_sensitive": int((off & ~o_flip).sum()), "oracle_u1_move_under_1ulp_of_sensitive": qs(o_du[o_flip])}})
    # A cold-start solve of this batch is ~650 interior-point solves deep (14 sweeps x 40 + line-search decisions): under
    # 1-ulp input noise the ORACLE ITSELF leaves its discrete path on ~90 % of the rollouts (recorded above).  What parity
    # means at this size: (1) the same Newton iteration count on the large majority, (2) device-vs-oracle path differences
    # only where the oracle is itself last-place sensitive, (3) off-path results no further from the oracle than the
    # oracle moves under last-place noise, (4) on-path results tight, (5) identical batch statistics.
    assert same_it.mean() >= 0.9, same_it.mean()
    assert (off & ~o_flip).sum() <= 0.05 * B, int((off & ~o_flip).sum())
    if off.any() and o_flip.any():
        assert np.median(du[off]) <= 3 * np.median(o_du[o_flip]) + 1e-9, (np.median(du[off]), np.median(o_du[o_flip]))
        assert np.quantile(du[off], 0.9) <= 5 * np.quantile(o_du[o_flip], 0.9) + 1e-9
    if same_path.any():
        assert np.median(du[same_path]) < 1e-7 and np.median(dq[same_path]) < 1e-7
    assert abs(int((rn < 3e-4).sum()) - int((o_rn < 3e-4).sum())) <= 0.03 * B
    assert abs(it.mean() - o_it.mean()) < 0.05 and abs(cnt["sweeps"].mean() - o_sw.mean()) < 0.5
    assert abs(cnt["ip_iters"].mean() / o_ip.mean() - 1) < 0.02
    assert abs(np.median(rn) / np.median(o_rn) - 1) < 0.05


@pytest.mark.parametrize("name,dims,kind,Hh,Hr,mode", [
    ("hopper_h20 (BASELINE configs[1])", dict(nq=4, nu=2, nw=2, nc=1, nb=2), "hopper", 20, 30, 0),
    ("pushbot_h10_configurationforce (BASELINE configs[0])", dict(nq=2, nu=2, nw=2, nc=2, nb=4), "pushbot", 10, 16, 1),
    ("quadruped_h40 (BASELINE configs[2])", dict(nq=11, nu=8, nw=2, nc=4, nb=8), "quadruped", 40, 60, 0),
])
def test_single_rollout_configs_vs_oracle(gpu_required, name, dims, kind, Hh, Hr, mode):
    """The B = 1 configurations of BASELINE.json exactly as bench.py's latency legs build them (`mpc_loop_latency`):
    cold-start newton_solve!, then three warm-started MPC steps with rot_n_stride! / update_window! on the device, against
    the numpy oracle (reference-default dense LU) running the same loop."""
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions, synthetic as synth
    from contactimplicitmpc.jl_amd.trajectory import Dims
    from oracle import lcp, mpc as ompc
    d = Dims(**dims, mode=mode)
    prob = synth.make_problem(d, Hr, seed=1)
    obj = synth.make_objective(d, Hh, kind=kind)
    window, ref, q0, q1 = synth.make_rollout(d, prob, Hh, phase=0, seed=7, perturb=0.01)
    s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, Hr, Hh, B=1, mode=mode, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]),
                    newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-4, max_iter=5))
    for t in range(Hr):
        s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
    if mode == 1:
        s.set_objective(obj.q, obj.u, obj.gamma, obj.b)
    else:
        s.set_objective(obj.q, obj.u)
    stride = np.zeros(d.nq)
    stride[0] = prob["q_ref"][-2][0] - prob["q_ref"][0][0]
    s.set_gait(prob["q_ref"], prob["u_ref"], prob["th0"], stride, w=prob["w_ref"], gamma=prob["gamma_ref"], b=prob["b_ref"])
    tabs = [lcp.LinTable(d, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t]) for t in range(Hr)]
    full = onewton.Traj(q=prob["q_ref"].copy(), u=prob["u_ref"].copy(), w=prob["w_ref"].copy(), gamma=prob["gamma_ref"].copy(),
                        b=prob["b_ref"].copy(), theta=prob["th0"].copy())
    win = np.arange(Hh + 2)
    cut = lambda T: onewton.Traj(q=T.q[:Hh + 2].copy(), u=T.u[:Hh].copy(), w=T.w[:Hh].copy(), gamma=T.gamma[:Hh].copy(), b=T.b[:Hh].copy(), theta=T.theta[:Hh].copy())
    core = onewton.Newton(d, Hh, obj, onewton.NewtonOptions(r_tol=3e-4, max_iter=5, solver="lu"), oip.IPOptions(kappa_tol=prob["kappa"]), prob["kappa"], cut(full))
    a, b_ = q0.copy(), q1.copy()
    rec = []
    on_path = True
    for k in range(4):
        u1, it, rn = s.newton_solve(a[None], b_[None], warm_start=k > 0)
        tr = s.trajectory(); cnt = s.rollout_counters()
        st = onewton.newton_solve(core, a, b_, win, tabs, cut(full), warm_start=k > 0)
        same = int(it[0]) == st.iters and int(cnt["sweeps"][0]) == st.sweeps and int(cnt["ip_iters"][0]) == st.ip_iters
        du = float(np.abs(u1[0] - core.traj.u[0]).max()); dq = float(np.abs(tr["q"][0] - core.traj.q).max())
        rec.append({"step": k, "newton_iters": [int(it[0]), st.iters], "sweeps": [int(cnt["sweeps"][0]), st.sweeps],
                    "ip_iters": [int(cnt["ip_iters"][0]), st.ip_iters], "same_path": same, "u1_diff": du, "q_diff": dq})
        on_path = on_path and same
        if on_path:     # same discrete history so far: the warm-start state is the same on both sides
            assert du < 1e-6 and dq < 1e-6, rec
        else:           # after a flipped interior-point decision the two loops carry different warm starts
            assert du < 0.05 * max(1.0, np.abs(core.traj.u[0]).max()), rec
        # advance both loops from the ORACLE's planned next configuration (keeps the two loops on one input sequence)
        nxt = core.traj.q[2].copy()
        s.mpc_advance(stride)
        ompc.rot_n_stride(d, full, stride)
        win = ompc.update_window(win, Hr)
        a, b_ = b_, nxt
    s.close()
    _record("single_rollout/" + name, rec)
    # (the synthetic quadruped loop of bench.py is not a stabilising controller: its residual grows step by step and the
    #  solves become chaotic - one interior-point solve that jams on one side changes the warm start of every later step;
    #  tight bounds hold while the two loops share their history, which the recorded table shows per step)
    for r in rec[:2]:       # cold start and the first warm-started step: same Newton iterations / sweeps, same control
        assert r["newton_iters"][0] == r["newton_iters"][1] and r["sweeps"][0] == r["sweeps"][1] and r["u1_diff"] < 1e-5, rec
