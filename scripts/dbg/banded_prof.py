"""Phase clocks of the banded LDL^T KKT kernel (library built with -DCIMPC_KKT_PROF -DCIMPC_BANDED_PROF, selected through CIMPC_LIB):
one centroidal H = 60 solve with the example's velocity objective through the B1 seam."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
H = 60
I = bench.centroidal_payload_inputs(2, H)
m, P, kappa, ro = I["m"], I["P"], I["kappa"], I["rollouts"][:1] * 64
Q, R, V, vt = bench.centroidal_velocity_objective(m, H)
s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=64, mode=0, ip_opts=InteriorPointOptions(kappa_tol=kappa, r_tol=1e-4),
                newton_opts=NewtonOptions(kappa=kappa, r_tol=3e-5, max_iter=5))
for t in range(P.H):
    s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
s.set_objective(Q, R, V=V, v_target=vt)
s.set_window(np.stack([r["window"] for r in ro]) + 1)
s.set_reference(*(np.stack([r[k] for r in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")))
s.implicit_dynamics(np.stack([r["q"] for r in ro]), np.stack([r["theta"] for r in ro]))
rhs = np.random.default_rng(0).standard_normal((1, s.N)).repeat(64, 0)
for _ in range(3):
    s.kkt_solve(rhs, 10.0)
lib = s.lib
lib.cimpc_debug_read_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
out = (C.c_longlong * 168)()
lib.cimpc_debug_read_stats(s.h, out, 168)
v = np.array(list(out), dtype=np.float64)[8:].reshape(16, 10) / 1e3
names0 = ["pre+fill+P1(0)", "P2", "barrier", "tile00 | rows/rhs/stores", "P1 next | tiles", "fetch", "barrier", "-", "back subst", "recovery"]
print("k-cycles per phase (rollout 0 of 64), one row per wavefront; total of wavefront 0: %.1f" % v[0].sum())
print("wave " + " ".join("%17s" % n for n in names0))
for wv_ in range(16):
    print("%4d " % wv_ + " ".join("%17.1f" % x for x in v[wv_]))
