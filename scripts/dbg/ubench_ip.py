"""Shader-clock cost of the interior-point primitives on a lone workgroup (diagnostic library built with -DCIMPC_UBENCH):
   make -C contactimplicitmpc/jl_amd/csrc BUILD=build_ub OUT=../libcimpc_ub.so EXTRA=-DCIMPC_UBENCH -j8
   CIMPC_LIB=contactimplicitmpc/jl_amd/libcimpc_ub.so python scripts/dbg/ubench_ip.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import bench
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions, _lib
MODEL = sys.argv[1] if len(sys.argv) > 1 else "quadruped"      # quadruped | centroidal (the 32-lane throughput build)
H, H_ref, B = 40, 60, 1
if MODEL == "quadruped":
    d, prob, obj, ro = bench.build_inputs(B, H, H_ref, seed=1234, perturb=0.05)
else:
    from common import make_case
    d, prob, tabs, ro = make_case("centroidal", 0, H_ref=H_ref, H=H, B=B, seed=3, perturb=5e-3, kappa=1e-3)
ipo = InteriorPointOptions(kappa_tol=prob["kappa"])
s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=B, mode=0, ip_opts=ipo, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-4, max_iter=5))
for t in range(H_ref):
    s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
lib = _lib.load()
lib.cimpc_debug_table_ptr.restype = C.c_void_p
lib.cimpc_debug_table_ptr.argtypes = [C.c_void_p]
tp = lib.cimpc_debug_table_ptr(s.h)
f = getattr(lib, "cimpc_ubench_ip_" + MODEL)
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_longlong)]
o = _lib.IpOpts()
lib.cimpc_default_ip_opts(C.byref(o))
o.kappa_tol = prob["kappa"]
names = ["residual", "violations (2 max-reductions)", "factorize", "linear_solve", "step_length", "iterate (whole)"]
reps = 200
for waves in ((1, 2, 4) if MODEL == "quadruped" else (1, 4, 8)):
    buf = (C.c_longlong * 64)()
    rc = f(tp, C.byref(o), waves, reps, buf)
    v = np.array(list(buf)).reshape(8, 8)
    print("waves per workgroup %d (rc %d):" % (waves, rc))
    for k, n in enumerate(names):
        print("   %-32s %8.0f clocks per call (wave 0)   %s" % (n, v[0, k] / reps, " ".join("%.0f" % (v[w, k] / reps) for w in range(1, min(waves, 8)))))
s.close()
