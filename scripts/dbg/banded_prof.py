"""Phase clocks of the banded LDL^T KKT kernel (library built with -DCIMPC_KKT_PROF -DCIMPC_BANDED_PROF, selected through CIMPC_LIB):
one centroidal H = 60 solve with the example's velocity objective through the B1 seam."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
H = 60
I = bench.centroidal_payload_inputs(2, H)
m, P, kappa, ro = I["m"], I["P"], I["kappa"], I["rollouts"][:1]
Q, R, V, vt = bench.centroidal_velocity_objective(m, H)
s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=1, mode=0, ip_opts=InteriorPointOptions(kappa_tol=kappa, r_tol=1e-4),
                newton_opts=NewtonOptions(kappa=kappa, r_tol=3e-5, max_iter=5))
for t in range(P.H):
    s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
s.set_objective(Q, R, V=V, v_target=vt)
s.set_window(np.stack([r["window"] for r in ro]) + 1)
s.set_reference(*(np.stack([r[k] for r in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")))
s.implicit_dynamics(np.stack([r["q"] for r in ro]), np.stack([r["theta"] for r in ro]))
rhs = np.random.default_rng(0).standard_normal((1, s.N))
for _ in range(3):
    s.kkt_solve(rhs, 10.0)
lib = s.lib
lib.cimpc_debug_read_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
out = (C.c_longlong * 32)()
lib.cimpc_debug_read_stats(s.h, out, 32)
v = np.array(list(out), dtype=np.float64)
names0 = ["pre-pass + fill", "P1 diagonal block", "fetch issue + barrier", "P2 rows", "barrier", "stores + commit", "update", "barrier",
          "back substitution", "control recovery"]
tot = v[8:18].sum()
print("wavefront 0 (k-cycles, share):")
for n, x in zip(names0, v[8:18]):
    print("  %-22s %9.1f  %5.1f %%" % (n, x / 1e3, 100 * x / tot))
print("  total %.1f k-cycles" % (tot / 1e3))
print("wavefront 4 (k-cycles):", dict(zip(names0, np.round(v[20:30] / 1e3, 1))))
