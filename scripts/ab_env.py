"""A/B of ONE environment knob on the latency-bound legs, both settings in one process on one box (the knobs are read at cimpc_create):
    python scripts/ab_env.py CIMPC_KKT_TWISTED 0 1 [--cent] [--b64]
Legs: B = 1 cold newton_solve! (quadruped H = 40), the warm-started B = 1 loops of bench.py, optionally the centroidal
configs[4] leg (64 rollouts, H = 60) and the quadruped batch of 64 rollouts."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def cold_b1(device=0, B=1, n=30):
    import torch
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
    d, prob, obj, ro = bench.build_inputs(B, 40, 60, 1234, 0.05)
    s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, 60, 40, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]),
                    newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-4, max_iter=5), device=device)
    for t in range(60):
        s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
    s.set_objective(obj.q, obj.u)
    s.set_window(np.stack([w for (w, _, _, _) in ro]) + 1)
    s.set_reference(np.stack([r.q for (_, r, _, _) in ro]), np.stack([r.u for (_, r, _, _) in ro]), np.stack([r.w for (_, r, _, _) in ro]),
                    np.stack([r.gamma for (_, r, _, _) in ro]), np.stack([r.b for (_, r, _, _) in ro]), np.stack([r.theta for (_, r, _, _) in ro]))
    q0 = torch.tensor(np.stack([r[2] for r in ro]), dtype=torch.float64, device="cuda")
    q1 = torch.tensor(np.stack([r[3] for r in ro]), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    s.profile_enable(1); s.profile_reset()
    for _ in range(5):
        s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
    torch.cuda.synchronize()
    p = s.profile_read()
    u1, it, rn = s.newton_info()
    tw = s.kkt_twisted()
    s.close()
    return {"ms": round(ms, 4), "kkt_ms": round(p["kkt_ms"] / 5, 4), "kkt_launches": p.get("kkt_launches", 0) / 5, "sweep_ms": round(p["ip_sweep_ms"] / 5, 4),
            "resid_ms": round(p["resid_ms"] / 5, 4), "iters": [int(x) for x in it[:4]], "twisted_launches": tw}


def main():
    knob, vals = sys.argv[1], [v for v in sys.argv[2:] if not v.startswith("--")]
    flags = [v for v in sys.argv[2:] if v.startswith("--")]
    for rep in range(2):
        for v in vals:
            os.environ[knob] = v
            out = {"cold_b1": cold_b1()}
            out["quadruped_loop"] = round(bench.mpc_loop_latency(bench.QUADRUPED, "quadruped", 40, 60, 0)["ms_per_mpc_step"], 4)
            out["hopper_loop"] = round(bench.mpc_loop_latency(dict(nq=4, nu=2, nw=2, nc=1, nb=2), "hopper", 20, 30, 0)["ms_per_mpc_step"], 4)
            out["gait2_loop"] = round(bench.real_mpc_loop_latency(40, 0)["ms_per_mpc_step"], 4)
            if "--b64" in flags:
                out["cold_b64"] = cold_b1(B=64, n=10)
            if "--cent" in flags and rep == 0:
                c = bench.centroidal_payload_leg(64, 60, 0)
                out["centroidal"] = {k: c[k] for k in c if k in ("fp64_kkt", "velocity_objective")}
            print(knob, v, json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
