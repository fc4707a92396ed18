#!/usr/bin/env python3
"""Generates tests/golden/*.npz: seeded synthetic inputs and the oracle's outputs for the path.

The reference ships no golden vectors for this path (random-data
self-consistency tests only) and cannot be executed here (no Julia; RoboDojo.jl absent), so the
fixtures are produced by the numpy oracle (oracle/), which is itself pinned by
tests/test_oracle_reference_constructions.py.  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import ip as oip, newton as onewton, synth  # noqa: E402
from common import make_case, oracle_sweep  # noqa: E402


def one(name, model, mode, H_ref, H, B, seed, perturb, newton, velocity=False):
    d, prob, tabs, rollouts = make_case(model, mode, H_ref=H_ref, H=H, B=B, seed=seed, perturb=perturb)
    opts = oip.IPOptions(kappa_tol=prob["kappa"])
    ref = oracle_sweep(d, tabs, rollouts, opts)
    out = dict(model=model, mode=mode, H_ref=H_ref, H=H, B=B, kappa=prob["kappa"],
               z0=prob["z0"], th0=prob["th0"], r0=prob["r0"], rz0=prob["rz0"], rth0=prob["rth0"],
               window=np.stack([w for (w, _, _, _) in rollouts]),
               q_ref=np.stack([r.q for (_, r, _, _) in rollouts]), u_ref=np.stack([r.u for (_, r, _, _) in rollouts]),
               w_ref=np.stack([r.w for (_, r, _, _) in rollouts]), gamma_ref=np.stack([r.gamma for (_, r, _, _) in rollouts]),
               b_ref=np.stack([r.b for (_, r, _, _) in rollouts]), theta_ref=np.stack([r.theta for (_, r, _, _) in rollouts]),
               q0=np.stack([r[2] for r in rollouts]), q1=np.stack([r[3] for r in rollouts]),
               sweep_q=np.stack([t.q for t, _ in ref]), sweep_theta=np.stack([t.theta for t, _ in ref]),
               sweep_gamma=np.stack([t.gamma for t, _ in ref]), sweep_b=np.stack([t.b for t, _ in ref]))
    for k in ("d", "dq0", "dq1", "du1", "status", "iters", "z"):
        out["sweep_" + k] = np.stack([o[k] for _, o in ref])
    if newton:
        obj = synth.make_objective(d, H, kind=model, velocity=velocity)
        out["obj_q"], out["obj_u"] = obj.q, obj.u
        if mode == 1:
            out["obj_gamma"], out["obj_b"] = obj.gamma, obj.b
        if velocity:
            obj.v = obj.v * 1e3
            obj.v_target = 0.01 * np.random.default_rng(seed).standard_normal((H, d.nq))
            obj.q_target = None
            obj.__post_init__()
            out["obj_v"], out["obj_q_target"], out["obj_v_target"] = obj.v, obj.q_target, obj.v_target
        qs, us, nus, its, rns, ipit = [], [], [], [], [], []
        for (window, rf, q0, q1) in rollouts:
            core = onewton.Newton(d, H, obj, onewton.NewtonOptions(r_tol=1e-5, max_iter=5, solver="lu"), opts, prob["kappa"], rf)
            st = onewton.newton_solve(core, q0, q1, window, tabs, rf)
            qs.append(core.traj.q.copy()); us.append(core.traj.u.copy()); nus.append(core.nu.copy())
            its.append(st.iters); rns.append(st.r_norm / core.lay.N); ipit.append(st.ip_iters)
        out.update(newton_q=np.stack(qs), newton_u=np.stack(us), newton_nu=np.stack(nus), newton_iters=np.array(its),
                   newton_rnorm=np.array(rns), newton_ip_iters=np.array(ipit), newton_r_tol=1e-5, newton_max_iter=5)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: getattr(v, "shape", v) for k, v in out.items() if k.startswith("sweep_d") or k == "newton_iters"})


if __name__ == "__main__":
    np.seterr(all="ignore")
    one("hopper_cfg", "hopper", 0, 6, 5, 3, 31, 1e-2, True)
    one("quadruped_cfg", "quadruped", 0, 8, 6, 3, 32, 1e-2, True)
    one("pushbot_cf", "pushbot", 1, 5, 4, 2, 33, 1e-2, False)
    # newton_solve! through the reference-default dense-LU KKT: :configurationforce, and the velocity objective
    one("hopper_cf_newton", "hopper", 1, 8, 6, 3, 34, 5e-3, True)
    one("hopper_velocity_newton", "hopper", 0, 8, 6, 3, 35, 5e-3, True, velocity=True)
