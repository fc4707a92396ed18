"""centroidal_quadruped_wall (ny = 48, compiled 64-lane sweep): sensitivities of every converged knot against the oracle's, per solve.
CIMPC_LIB selects the library.  python scripts/dbg/wall_dz_cmp.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
from common import make_case, make_solver, oracle_sweep
from oracle import ip as oip
from contactimplicitmpc.jl_amd import InteriorPointOptions
H, H_ref, B = 6, 8, 3
d, prob, tabs, rollouts = make_case("centroidal_wall", 0, H_ref=H_ref, H=H, B=B, seed=4, perturb=1e-2)
s = make_solver(d, prob, rollouts, H, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]))
ref = oracle_sweep(d, tabs, rollouts, oip.IPOptions(kappa_tol=prob["kappa"]))
out = s.implicit_dynamics(np.stack([t.q for t, _ in ref]), np.stack([t.theta for t, _ in ref]))
for b, (_, o) in enumerate(ref):
    for i in range(H):
        same = out["iters"][b, i] == o["iters"][i] and out["status"][b, i] == o["status"][i]
        e = [np.abs(out[k][b, i][:d.nq] - o[k][i][:d.nq]).max() / max(1.0, np.abs(o[k][i]).max()) for k in ("dq0", "dq1", "du1")]
        print("b %d i %d iters %d/%d status %d/%d same %d  d err %.2e  dz err %.2e %.2e %.2e" % (b, i, out["iters"][b, i], o["iters"][i], out["status"][b, i], o["status"][i], same,
              np.abs(out["d"][b, i][:d.nq] - o["d"][i][:d.nq]).max(), *e))
s.close()
