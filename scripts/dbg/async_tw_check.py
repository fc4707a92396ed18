"""The persistent kernel's KKT stage as two cooperating jobs (CIMPC_ASYNC_KKT_TW=1, the chains of the twisted solve) against the one-ended
job (=0): same Newton iterations, controls, and the time of a cold solve.  python scripts/dbg/async_tw_check.py [B ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa
import bench
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
H, H_ref = 40, 60
for B in [int(a) for a in sys.argv[1:]] or [8, 64]:
    d, prob, obj, ro = bench.build_inputs(B, H, H_ref, seed=1234, perturb=0.05)
    q0 = np.stack([r[2] for r in ro]); q1 = np.stack([r[3] for r in ro])
    outs = {}
    for tw in (0, 1, 0, 1):
        os.environ["CIMPC_ASYNC_KKT_TW"] = str(tw)
        s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]),
                        newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-4, max_iter=5))
        for t in range(H_ref):
            s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
        s.set_objective(obj.q, obj.u)
        s.set_window(np.stack([w for (w, _, _, _) in ro]) + 1)
        s.set_reference(*[np.stack([getattr(r, k) for (_, r, _, _) in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")])
        ts = []
        for rep in range(6):
            t0 = time.perf_counter()
            u1, it, rn = s.newton_solve(q0, q1)
            ts.append(time.perf_counter() - t0)
        print("B", B, "tw", tw, "ms/solve (host arrays) min %.3f med %.3f" % (1e3 * min(ts), 1e3 * sorted(ts)[len(ts) // 2]), "newton iters", int(it.sum()),
              "converged", int((rn < 3e-4).sum()), "twisted", s.kkt_twisted(), "fallbacks", s.kkt_twisted_fallbacks(), "finite", bool(np.isfinite(u1).all()), flush=True)
        outs.setdefault(tw, (u1.copy(), it.copy(), rn.copy()))
        s.close()
    a, b = outs[0], outs[1]
    same = a[1] == b[1]
    print("B", B, "same newton iterations on", int(same.sum()), "of", B, "| max |u1 diff| on those %.3e" % (np.abs(a[0] - b[0])[same].max() if same.any() else -1), flush=True)
