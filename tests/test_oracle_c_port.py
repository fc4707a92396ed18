"""The C restatement (oracle/cimpc_ref.c, the CPU baseline of bench.py) against the numpy oracle.
Borderline interior-point solves may flip one discrete decision between two fp64 summation
orders (numpy's pairwise/BLAS dots vs. the C loops); such solves are excluded knot by knot."""
import numpy as np

from oracle import ip as oip
from oracle import lcp, newton as onewton, synth
from oracle.cref import CRef
from oracle.dims import Dims, QUADRUPED, PUSHBOT


def _setup(model=QUADRUPED, mode=0, H_ref=12, H=8, seed=1):
    d = Dims(**model, mode=mode)
    prob = synth.make_problem(d, H_ref, seed=seed)
    tabs = [lcp.LinTable(d, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t]) for t in range(H_ref)]
    obj = synth.make_objective(d, H)
    return d, prob, tabs, obj


def test_c_sweep_matches_numpy():
    for model, mode in ((QUADRUPED, 0), (PUSHBOT, 1)):
        d, prob, tabs, obj = _setup(model, mode)
        H = 8
        cr = CRef(d, 12, H, prob, obj if mode == 0 else None, oip.IPOptions(), onewton.NewtonOptions(r_tol=3e-4, max_iter=5), prob["kappa"])
        n = agree = 0
        for seed in range(4):
            window, ref, q0, q1 = synth.make_rollout(d, prob, H, phase=3 * seed, seed=seed, perturb=2e-2)
            tr = ref.copy(); tr.q[0], tr.q[1] = q0, q1; tr.update_theta(d, 0); tr.update_theta(d, 1)
            o = oip.implicit_dynamics(d, tabs, window, tr.q, tr.theta, oip.IPOptions(), gamma=tr.gamma, b=tr.b)
            c = cr.implicit_dynamics(window, tr.q, tr.theta, tr.gamma, tr.b)
            for i in range(H):
                n += 1
                if o["iters"][i] == c["iters"][i] and o["status"][i] == c["status"][i]:
                    agree += 1
                    if o["status"][i]:
                        assert np.abs(o["z"][i] - c["z"][i]).max() < 1e-6
                        assert np.abs(o["d"][i] - c["d"][i]).max() < 1e-6
                        for k in ("dq0", "dq1", "du1"):
                            assert np.abs(o[k][i] - c[k][i]).max() < 1e-6 * max(1.0, np.abs(o[k][i]).max())
        assert agree >= 0.9 * n


def test_c_kkt_backends_match_dense_numpy():
    d, prob, tabs, obj = _setup()
    H = 8
    cr = CRef(d, 12, H, prob, obj, oip.IPOptions(), onewton.NewtonOptions(), prob["kappa"])
    window, ref, q0, q1 = synth.make_rollout(d, prob, H, phase=1, seed=3, perturb=2e-2)
    tr = ref.copy(); tr.q[0], tr.q[1] = q0, q1; tr.update_theta(d, 0); tr.update_theta(d, 1)
    c = cr.implicit_dynamics(window, tr.q, tr.theta)
    lay = onewton.Layout(d, H)
    r = np.random.default_rng(0).standard_normal(lay.N)
    for beta in (1e-5, 10.0):
        R = onewton.jacobian(lay, obj, c, beta, prob["kappa"])
        x = np.linalg.solve(R, r)
        for solver in (0, 1):
            y = cr.kkt_solve(c["dz_raw"], beta, r, solver)
            assert np.abs(x - y).max() < 1e-8 * max(1.0, np.abs(x).max())


def test_c_newton_matches_numpy():
    d, prob, tabs, obj = _setup(H=8)
    H = 8
    nopt = onewton.NewtonOptions(r_tol=1e-5, max_iter=5, solver="lu")
    cr = CRef(d, 12, H, prob, obj, oip.IPOptions(), nopt, prob["kappa"])
    same = 0
    for seed in range(4):
        window, ref, q0, q1 = synth.make_rollout(d, prob, H, phase=2 * seed, seed=10 + seed, perturb=1e-2)
        core = onewton.Newton(d, H, obj, nopt, oip.IPOptions(), prob["kappa"], ref)
        st = onewton.newton_solve(core, q0, q1, window, tabs, ref)
        for solver in (0, 1):
            c = cr.newton_solve(window, ref, q0, q1, solver=solver)
            assert c["iters"] == st.iters
            if c["ip_iters"] == st.ip_iters and c["sweeps"] == st.sweeps:
                same += 1
                assert np.abs(c["q"] - core.traj.q).max() < 1e-6
                assert np.abs(c["u"] - core.traj.u).max() < 1e-6
    assert same >= 4      # the rest took a different (equally valid) discrete IP path under roundoff
