"""-m gpu: the parity holes VERDICT r02 lists (next-round item 1), all on committed inputs and through the C ABI.

(a) BASELINE configs[4] AS NAMED: `bench.centroidal_payload_inputs(64, 60)` - the bench leg's exact inputs (inplace_trot_v7, H = 60,
    64 rollouts = 512 / 8 GPUs, payload w_z = -5 .. -30 N; examples/centroidal_quadruped/continuous_trot.jl:37-81,
    src/dynamics/centroidal_quadruped/model.jl:121-125) through `newton_solve!` on BOTH condensed KKT backends (fp64 and the fp32-MFMA
    mixed-precision one) against `oracle/cimpc_ref.c`, with the last-place arbiter of test_gpu_full_size_parity.py.
(c) the full-size run of the real workload: `bench.real_problem_inputs(512, 40)` (quadruped gait2, 512 rollouts, the
    `real_problem` leg of the bench line) against the oracle - Newton iterations identical on >= 99 %; full discrete paths measured
    against the oracle's own path-change rate under last-place input noise (the expectation of >= 95 % identical paths does not
    hold for the oracle against itself: 76 % per draw).
(d) the reference's known answer for the ill-conditioned regime the interior point ends in, test/solver/schur.jl:19-62 (flamingo,
    knot 10, z0 = max(1e-6, z), kappa = 0, |M [x; y] - [u; v]|_inf < 1e-7), on the device's factor / solve (B2 seam:
    cimpc_ip_linear_solve), next to test/controller/linearized_solver.jl:55-57 (Delta = rz0 \\ r0, 1e-10) and rlin! (B2: cimpc_ip_residual).
"""
import json
import os
import time

import numpy as np
import pytest

from oracle import ip as oip, lcp
from oracle import newton as onewton
from oracle.cref import CRef
from oracle.dims import Dims
from oracle.newton import Traj

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORD = {}
ORACLE_EXTRA = {}      # side results of the last _oracle call: same-path scatter, flip rate per draw


def _record(key, val):
    RECORD[key] = val
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity_round3.json"), "w") as f:
        json.dump(RECORD, f, indent=1, sort_keys=True)
    print("[parity_round3]", key, json.dumps(val))


def _qs(a):
    a = np.asarray(a, dtype=float)
    return {"median": float(np.quantile(a, 0.5)), "q90": float(np.quantile(a, 0.9)), "max": float(a.max())} if a.size else {}


def _device(I, H, backend, ip_r_tol, newton_r_tol):
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
    m, P, kappa, ro = I["m"], I["P"], I["kappa"], I["rollouts"]
    s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=len(ro), mode=0, ip_opts=InteriorPointOptions(kappa_tol=kappa, r_tol=ip_r_tol),
                    newton_opts=NewtonOptions(kappa=kappa, r_tol=newton_r_tol, max_iter=5, kkt_backend=backend))
    for t in range(P.H):
        s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
    s.set_objective(I["Q"], I["R"])
    s.set_window(np.stack([r["window"] for r in ro]) + 1)
    s.set_reference(*(np.stack([r[k] for r in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")))
    u1, it, rn = s.newton_solve(np.stack([r["q0"] for r in ro]), np.stack([r["q1"] for r in ro]))
    out = dict(u1=u1, it=it, rn=rn, q=s.trajectory()["q"], cnt=s.rollout_counters(), fallbacks=s.kkt_fallbacks() if backend == 3 else 0)
    s.close()
    return out


def _oracle(I, H, ip_r_tol, newton_r_tol, draws):
    """oracle/cimpc_ref.c on every rollout (condensed KKT) + `draws` re-runs with (q0, q1) and the theta of EVERY interior-point
    solve of the run perturbed in the last place."""
    from contactimplicitmpc.jl_amd.trajectory import Objective
    m, P, kappa, ro = I["m"], I["P"], I["kappa"], I["rollouts"]
    d = Dims(nq=m.nq, nu=m.nu, nw=m.nw, nc=m.nc, nb=m.nb, mode=0)
    prob = dict(z0=P.z, th0=P.theta, r0=P.r0, rz0=P.rz0, rth0=P.rth0)
    cr = CRef(d, P.H, H, prob, Objective(q=I["Q"], u=I["R"]), oip.IPOptions(kappa_tol=kappa, r_tol=ip_r_tol),
              onewton.NewtonOptions(r_tol=newton_r_tol, max_iter=5), kappa)
    refs = [Traj(q=r["q"], u=r["u"], w=r["w"], gamma=r["gamma"], b=r["b"], theta=r["theta"]) for r in ro]
    t0 = time.perf_counter()
    base = [cr.newton_solve(r["window"], refs[b], r["q0"], r["q1"], solver=1) for b, r in enumerate(ro)]
    t_oracle = time.perf_counter() - t0
    B = len(ro)
    rng = np.random.default_rng(7)
    flip = np.zeros(B, dtype=bool); move = np.zeros(B); move_same = np.zeros(B); per_draw = []
    for _ in range(draws):
        n_flip = 0
        for b, r in enumerate(ro):
            p0 = r["q0"] * (1.0 + (rng.integers(0, 2, r["q0"].shape) * 2 - 1) * 2.0 ** -52)
            p1 = r["q1"] * (1.0 + (rng.integers(0, 2, r["q1"].shape) * 2 - 1) * 2.0 ** -52)
            # ... and the reference trajectory's theta at EVERY step (the device's arithmetic differs from the oracle's in the last
            # place at every knot of the first sweep, not only at the two steps that read q0 / q1)
            rp = Traj(q=refs[b].q, u=refs[b].u, w=refs[b].w, gamma=refs[b].gamma, b=refs[b].b,
                      theta=refs[b].theta * (1.0 + (rng.integers(0, 2, refs[b].theta.shape) * 2 - 1) * 2.0 ** -52))
            # ... and theta of every LATER evaluation too (the candidates of the line searches): oracle arbiter mode
            cr.set_noise(int(rng.integers(1, 2 ** 62)))
            o = cr.newton_solve(r["window"], rp, p0, p1, solver=1)
            cr.set_noise(0)
            f = (o["iters"] != base[b]["iters"]) or (o["sweeps"] != base[b]["sweeps"]) or (o["ip_iters"] != base[b]["ip_iters"])
            flip[b] |= f; n_flip += int(f)
            du = np.abs(o["u"][0] - base[b]["u"][0]).max()
            move[b] = max(move[b], du)
            if not f:
                move_same[b] = max(move_same[b], du)      # the oracle's own last-place scatter while it stays on its path
        per_draw.append(n_flip / B)
    ORACLE_EXTRA["move_same"] = move_same; ORACLE_EXTRA["flip_rate_per_draw"] = per_draw

    def rerun_dense_lu(idx):      # the same rollouts through the oracle's dense-LU KKT backend: |u1_lu - u1_condensed|
        return np.array([np.abs(cr.newton_solve(ro[b]["window"], refs[b], ro[b]["q0"], ro[b]["q1"], solver=0)["u"][0] - base[b]["u"][0]).max() for b in idx])
    ORACLE_EXTRA["rerun_dense_lu"] = rerun_dense_lu
    return base, flip, move, t_oracle, d


def _compare(dev, base, flip, move):
    o_it = np.array([r["iters"] for r in base]); o_sw = np.array([r["sweeps"] for r in base]); o_ip = np.array([r["ip_iters"] for r in base])
    o_fail = np.array([r["ip_fail"] for r in base]); o_rn = np.array([r["r_norm"] for r in base]); o_u1 = np.stack([r["u"][0] for r in base])
    same_it = dev["it"] == o_it
    same_path = same_it & (dev["cnt"]["sweeps"] == o_sw) & (dev["cnt"]["ip_iters"] == o_ip)
    du = np.abs(dev["u1"] - o_u1).max(axis=1)
    dq = np.array([np.abs(dev["q"][b] - base[b]["q"]).max() for b in range(len(base))])
    off = ~same_path
    rec = {"rollouts": len(base), "same_newton_iters": int(same_it.sum()), "same_discrete_path (iters, sweeps, ip_iters)": int(same_path.sum()),
           "newton_iters_mean": [float(dev["it"].mean()), float(o_it.mean())], "sweeps_mean": [float(dev["cnt"]["sweeps"].mean()), float(o_sw.mean())],
           "ip_iters_mean": [float(dev["cnt"]["ip_iters"].mean()), float(o_ip.mean())], "ip_failures": [int(dev["cnt"]["ip_failures"].sum()), int(o_fail.sum())],
           "r_norm_median": [float(np.median(dev["rn"])), float(np.median(o_rn))], "u1_scale": float(np.abs(o_u1).max()),
           "u1_diff_same_path": _qs(du[same_path]), "q_diff_same_path": _qs(dq[same_path]), "u1_diff_off_path": _qs(du[off]),
           "ulp_arbiter": {"oracle_path_changes_under_1ulp": int(flip.sum()), "off_path_and_oracle_sensitive": int((off & flip).sum()),
                           "off_path_not_sensitive": int((off & ~flip).sum()), "oracle_u1_move_under_1ulp_of_sensitive": _qs(move[flip])}}
    return rec, same_it, same_path, du, dq


@pytest.fixture(scope="module")
def config4():
    import bench
    H, B = 60, 64
    I = bench.centroidal_payload_inputs(B, H)
    base, flip, move, t_oracle, d = _oracle(I, H, I["ip_r_tol"], I["newton_r_tol"], draws=2)
    return I, H, base, flip, move, t_oracle


@pytest.mark.parametrize("backend,name", [(0, "fp64_kkt"), (3, "mixed_fp32_mfma_kkt")])
def test_config4_centroidal_payload_h60_newton_solve_vs_oracle(gpu_required, config4, backend, name):
    """BASELINE configs[4] as named, on the bench leg's exact inputs, against oracle/cimpc_ref.c - either KKT backend."""
    I, H, base, flip, move, t_oracle = config4
    dev = _device(I, H, backend, I["ip_r_tol"], I["newton_r_tol"])
    rec, same_it, same_path, du, dq = _compare(dev, base, flip, move)
    rec["oracle_seconds"] = t_oracle
    rec["kkt_fp64_fallbacks"] = int(dev["fallbacks"]); rec["kkt_systems"] = int(dev["it"].sum())
    _record("config4_centroidal_payload_h60/" + name, rec)
    B = len(base)
    scale = max(1.0, rec["u1_scale"])
    assert np.abs(np.stack([r["w"] for r in I["rollouts"]])[:, :, 2]).min() >= 5.0          # the payload is live on every rollout
    assert int(np.array([r["iters"] for r in base]).max()) >= 2                            # real Newton work
    assert same_it.mean() >= 0.95, rec
    off = ~same_path
    assert (off & ~flip).sum() <= max(2, 0.05 * B), rec          # path differences only where the oracle is itself last-place sensitive
    assert same_path.sum() >= 0.5 * B, rec
    # controls: tight on the shared path; off the path no further from the oracle than a few times its own last-place motion
    assert du[same_path].max() <= 1e-6 * scale and np.median(dq[same_path]) <= 1e-7, rec
    if off.any():
        bound = 10.0 * (np.quantile(move[flip], 0.9) if flip.any() else 0.0) + 1e-5 * scale
        assert np.quantile(du[off], 0.9) <= bound, (rec, bound)
    assert abs(rec["newton_iters_mean"][0] - rec["newton_iters_mean"][1]) < 0.05
    assert abs(rec["sweeps_mean"][0] - rec["sweeps_mean"][1]) < 0.3
    assert abs(rec["ip_iters_mean"][0] / rec["ip_iters_mean"][1] - 1) < 0.01
    assert rec["ip_failures"][0] == rec["ip_failures"][1]
    if backend == 3:      # the fp64 fallback is the exception (cold-start beta only)
        assert dev["fallbacks"] <= 0.3 * int(dev["it"].sum()), rec


def test_real_gait2_full_size_newton_solve_vs_oracle(gpu_required):
    """The full-size run of the bench's `real_problem` leg (gait2, 512 rollouts, H = 40) against the oracle.  Identical Newton
    iteration counts on >= 99 % of the rollouts.  The full discrete path (every one of a rollout's ~230 interior-point solves with
    the same iteration count) is NOT a 95 % statement on this workload either: the oracle itself leaves its path on ~24 % of the
    rollouts when (q0, q1) move by one unit in the last place (r_tol = 1e-8 sits on the round-off floor of a solve's last full
    step) - what is asserted is that the device leaves the oracle's path NO MORE OFTEN than the oracle leaves
    its own under last-place noise, on the rollouts where the oracle is sensitive, and that values agree tightly wherever the
    oracle itself is stable."""
    import bench
    H, B = 40, 512
    I = bench.real_problem_inputs(B, H, 0.05)
    base, flip, move, t_oracle, d = _oracle(I, H, 1e-8, 3e-4, draws=4)
    move_same, per_draw = ORACLE_EXTRA["move_same"], ORACLE_EXTRA["flip_rate_per_draw"]
    dev = _device(I, H, 0, 1e-8, 3e-4)
    rec, same_it, same_path, du, dq = _compare(dev, base, flip, move)
    rec["oracle_seconds"] = t_oracle
    rec["converged"] = [int((dev["rn"] < 3e-4).sum()), int((np.array([r["r_norm"] for r in base]) < 3e-4).sum())]
    rec["oracle_off_own_path_rate_per_1ulp_draw"] = per_draw
    rec["device_off_oracle_path_rate"] = float((~same_path).mean())
    rec["oracle_u1_scatter_on_own_path"] = _qs(move_same[~flip])
    stable = same_path & ~flip & (move_same <= 1e-9)          # rollouts the oracle holds to 1e-9 under four last-place draws
    rec["u1_diff_where_oracle_is_stable"] = dict(_qs(du[stable]), rollouts=int(stable.sum()))
    _record("real_gait2_h40_512", rec)
    scale = max(1.0, rec["u1_scale"])
    assert same_it.mean() >= 0.99, rec
    off = ~same_path
    assert off.mean() <= 1.5 * float(np.mean(per_draw)) + 0.02, rec          # no more path changes than the oracle's own last-place rate
    assert (off & ~flip).sum() <= 0.05 * B, rec                               # ... and on the rollouts where the oracle is sensitive
    # `stable`: rollouts the oracle holds to 1e-9 through four draws of last-place noise on (q0, q1) AND on theta of every one of its
    # ~230 interior-point solves (oracle arbiter mode).  The device must agree tightly there; what it may still do is hit, once in a
    # while, the sensitive direction of an ill-conditioned solve that four draws did not hit - bounded by the oracle's own worst
    # same-path motion under that noise.  (Recorded beside it: the oracle with its other KKT backend on those rollouts.)
    outl = np.flatnonzero(stable & (du > 1e-7 * scale))
    rec["stable_rollouts_with_u1_diff_above_1e-7"] = int(outl.size)
    lu_move = ORACLE_EXTRA["rerun_dense_lu"](outl) if outl.size else np.zeros(0)
    rec["outliers"] = [{"rollout": int(b_), "u1_diff_device_vs_oracle": float(du[b_]), "oracle_u1_move_dense_lu_vs_condensed_kkt": float(m_)}
                       for b_, m_ in zip(outl, lu_move)]
    rec["oracle_worst_same_path_u1_move"] = float(move_same.max())
    _record("real_gait2_h40_512", rec)
    assert stable.sum() >= 0.25 * B and np.quantile(du[stable], 0.95) <= 1e-7 * scale and outl.size <= 0.03 * stable.sum(), rec
    assert outl.size == 0 or du[outl].max() <= 10 * move_same.max() + 1e-7 * scale, rec
    assert np.median(du[same_path]) <= 1e-9 * scale and np.median(dq[same_path]) <= 1e-9, rec
    # on-path rollouts that contain an ill-conditioned solve: no further off than 50 x the oracle's own scatter, in distribution
    assert np.quantile(du[same_path], 0.9) <= 50 * np.quantile(move_same[~flip], 0.9) + 1e-7 * scale, rec
    assert rec["ip_failures"][0] == rec["ip_failures"][1]
    assert abs(rec["converged"][0] - rec["converged"][1]) <= 0.01 * B
    assert abs(rec["ip_iters_mean"][0] / rec["ip_iters_mean"][1] - 1) < 5e-3
    assert abs(rec["sweeps_mean"][0] - rec["sweeps_mean"][1]) < 0.05


# --------------------------------------------------------------------------------------------------------------------------
# (d) B2 seam: the reference's known answers for rlin! / rzlin! / linear_solve! on the device arithmetic
# --------------------------------------------------------------------------------------------------------------------------
def _one_knot_solver(model, z0, th0, r0, rz0, rth0):
    from contactimplicitmpc.jl_amd import CIMPCSolver
    s = CIMPCSolver(model.nq, model.nu, model.nw, model.nc, model.nb, 1, 1, B=1, mode=0)
    s.set_linearization(1, z0, th0, r0, rz0, rth0)
    return s


def test_reference_schur_ill_conditioning_test_on_the_device(gpu_required):
    """test/solver/schur.jl:19-62: flamingo, knot 10 of gait_forward_36_4, z0 = max(1e-6, z), kappa0 = 0; M = [A B; C D] with
    D = rz0[irst, iy1] - rz0[irst, iy2] Diagonal(z0[iy2] ./ z0[iy1]); u = r0[idyn], v = r0[irst] - r0[ibil] ./ z0[iy1];
    schur_factorize!, schur_solve!: |M [x; y] - [u; v]|_inf < 1e-7.  Device: cimpc_ip_linear_solve at (z0, r0, reg = 0) returns
    Delta with Delta[ix] = x, Delta[iy1] = y (linearized_solver.jl:424-444 with Ry2 = the model's rz0[irst, iy2] diagonal)."""
    from real_problems import real_problem
    d, P, prob, tabs = real_problem("flamingo", 0.0)
    m = P.model
    z0 = np.maximum(1e-6, P.z[9]); th0 = P.theta[9].copy()
    r0, rz0, rth0 = m.linearize(z0, th0, 0.0)
    nx, ny = d.nx, d.ny
    iy1 = np.arange(nx, nx + ny); iy2 = np.arange(nx + ny, nx + 2 * ny)
    A = rz0[:nx, :nx]; Bm = rz0[:nx, iy1]; Cm = rz0[iy1, :nx]
    Ry2 = np.diag(rz0[np.ix_(iy1, iy2)])
    D = rz0[np.ix_(iy1, iy1)] - np.diag(Ry2 * z0[iy2] / z0[iy1])
    M1 = np.block([[A, Bm], [Cm, D]])
    u = r0[:nx]; v = r0[iy1] - Ry2 * r0[iy2] / z0[iy1]
    assert np.linalg.cond(M1) > 1e6                                   # the regime the test is about
    # the reference's own formula has Ry2 = I for this model (rst = s2 - ...: d rst / d y2 = I on the y2 rows it keeps)
    np.testing.assert_allclose(v, r0[iy1] - r0[iy2] / z0[iy1], rtol=0, atol=1e-12 * max(1.0, np.abs(v).max()))
    s = _one_knot_solver(m, z0, th0, r0, rz0, rth0)
    delta = s.ip_linear_solve(1, z0, r0, reg=0.0)[0]
    s.close()
    x, y = delta[:nx], delta[iy1]
    res = np.abs(M1 @ np.concatenate([x, y]) - np.concatenate([u, v])).max()
    # oracle side of the same statement (oracle/lcp.py: Schur + MGS-QR)
    tab = lcp.LinTable(d, z0, th0, r0, rz0, rth0)
    lcp.rzlin(tab, z0, 0.0)
    od = lcp.linear_solve_vec(tab, r0[:nx], r0[iy1], r0[iy2], 0.0)
    res_o = np.abs(M1 @ np.concatenate([od[:nx], od[iy1]]) - np.concatenate([u, v])).max()
    _record("schur_jl_19_62_flamingo_knot10", {"cond_M": float(np.linalg.cond(M1)), "residual_device": float(res), "residual_oracle": float(res_o),
                                                "max_abs_delta_diff_device_vs_oracle": float(np.abs(delta - od).max()), "delta_scale": float(np.abs(od).max())})
    assert res_o < 1e-7 and res < 1e-7, (res, res_o)
    # third block row of rz Delta = r: y2 .* Dy1 + y1 .* Dy2 = rbil
    assert np.abs(z0[iy2] * delta[iy1] + z0[iy1] * delta[iy2] - r0[iy2]).max() < 1e-9 * max(1.0, np.abs(r0[iy2]).max())


@pytest.mark.parametrize("which", ["quadruped", "flamingo", "centroidal"])
def test_reference_linearized_solver_test_on_the_device(gpu_required, which):
    """test/controller/linearized_solver.jl:40-57 on the TRUE model residuals, device arithmetic: rlin!(z0, theta0, kappa) reproduces
    r0 (the linearization is exact at its point), rlin! at a displaced point equals the dense linear model, and
    linear_solve!(Delta, rz, r) = rz0 \\ r0 to 1e-10 relative (well-conditioned point: kappa = 1e-3 central path)."""
    from real_problems import real_problem
    kappa = 1e-3
    d, P, prob, tabs = real_problem(which, kappa)
    m = P.model
    t = 4
    z0, th0, r0, rz0, rth0 = P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t]
    s = _one_knot_solver(m, z0, th0, r0, rz0, rth0)
    rng = np.random.default_rng(3)
    n = 7                                             # not a multiple of the problems per wavefront: ragged last workgroup
    zs = np.stack([z0] + [z0 * (1.0 + 0.05 * rng.standard_normal(z0.shape)) for _ in range(n - 1)])
    ths = np.stack([th0] + [th0 + 1e-2 * rng.standard_normal(th0.shape) for _ in range(n - 1)])
    r_dev = s.ip_residual(1, zs, ths, kappa)
    nx, ny = d.nx, d.ny
    scale = max(1.0, np.abs(r0).max())
    np.testing.assert_allclose(r_dev[0], r0, rtol=0, atol=1e-12 * scale)          # exact at the point (linearized_solver.jl:47-52)
    for k in range(n):
        rd, rr, rb = lcp.rlin(tabs[t], zs[k], ths[k], kappa)
        ref = np.concatenate([rd, rr, rb])
        np.testing.assert_allclose(r_dev[k], ref, rtol=0, atol=1e-11 * max(1.0, np.abs(ref).max()))
    # linear_solve! against the dense solve (linearized_solver.jl:55-57), at z0 and at the displaced points (positive y1, y2)
    zs = np.abs(zs) + 1e-3
    rs = np.stack([r0] + [rng.standard_normal(r0.shape) for _ in range(n - 1)])
    delta = s.ip_linear_solve(1, zs, rs, reg=0.0)
    s.close()
    for k in range(n):
        Mz = lcp.dense_rz(tabs[t], zs[k])
        ref = np.linalg.solve(Mz, rs[k])
        np.testing.assert_allclose(delta[k], ref, rtol=0, atol=1e-10 * max(1.0, np.abs(ref).max()) * max(1.0, np.linalg.cond(Mz) * 1e-6))
        lcp.rzlin(tabs[t], zs[k], 0.0)
        od = lcp.linear_solve_vec(tabs[t], rs[k][:nx], rs[k][nx:nx + ny], rs[k][nx + ny:], 0.0)
        np.testing.assert_allclose(delta[k], od, rtol=0, atol=1e-10 * max(1.0, np.abs(od).max()) * max(1.0, np.linalg.cond(Mz) * 1e-6))
