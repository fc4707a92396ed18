"""One banded KKT solve (centroidal H = 60, velocity objective, B1 seam) with the library CIMPC_LIB names; the solution goes to
gpurun_out/banded_delta_$TAG.npy so that two builds can be compared entry by entry (python scripts/dbg/banded_cmp.py cmp A B)."""
import os, sys
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "cmp":
    a, b = (np.load("gpurun_out/banded_delta_%s.npy" % t) for t in sys.argv[2:4])
    d = np.abs(a - b)
    print("cmp %s %s: max |diff| %.3e  rel %.3e  identical entries %d / %d" % (sys.argv[2], sys.argv[3], d.max(), d.max() / np.abs(a).max(), int((a == b).sum()), a.size))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
H = 60
I = bench.centroidal_payload_inputs(2, H)
m, P, kappa, ro = I["m"], I["P"], I["kappa"], I["rollouts"][:2]
Q, R, V, vt = bench.centroidal_velocity_objective(m, H)
s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=2, mode=0, ip_opts=InteriorPointOptions(kappa_tol=kappa, r_tol=1e-4),
                newton_opts=NewtonOptions(kappa=kappa, r_tol=3e-5, max_iter=5))
for t in range(P.H):
    s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
s.set_objective(Q, R, V=V, v_target=vt)
s.set_window(np.stack([r["window"] for r in ro]) + 1)
s.set_reference(*(np.stack([r[k] for r in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")))
s.implicit_dynamics(np.stack([r["q"] for r in ro]), np.stack([r["theta"] for r in ro]))
rhs = np.random.default_rng(0).standard_normal((2, s.N))
d = s.kkt_solve(rhs, 10.0)
np.save("gpurun_out/banded_delta_%s.npy" % os.environ.get("TAG", "x"), np.asarray(d))
print("solved", np.abs(d).max())
