"""Diagnostic: clock accounting of the lock-step sweep per wave (library built with -DCIMPC_SWEEP_PROF, selected through
CIMPC_LIB).  usage: CIMPC_LIB=.../libcimpc_prof.so python scripts/sweep_prof.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions, _lib  # noqa: E402

H, H_ref, B = 40, 60, 512
d, prob, obj, ro = bench.build_inputs(B, H, H_ref, seed=1234, perturb=0.05)
s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]),
                newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-4, max_iter=5))
for t in range(H_ref):
    s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
s.set_objective(obj.q, obj.u)
s.set_window(np.stack([w for (w, _, _, _) in ro]) + 1)
s.set_reference(*(np.stack([getattr(r, k) for (_, r, _, _) in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")))
q0 = torch.tensor(np.stack([r[2] for r in ro]), dtype=torch.float64, device="cuda")
q1 = torch.tensor(np.stack([r[3] for r in ro]), dtype=torch.float64, device="cuda")
lib = _lib.load()
f = lib.cimpc_debug_sweep_prof
f.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 16)()
s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
f(buf)
steps = 4
for _ in range(steps):
    s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
f(buf)
v = np.array(list(buf), dtype=np.float64)
names = ["lifetime", "barriers+pick", "stage", "end-of-solve", "pull", "ip trips", "sens trips"]
waves = v[7]
print("waves (all launches of %d steps): %d, pulls %d, ip trips %d, sens trips %d" % (steps, waves, v[8], v[9], v[10]))
for k, n in enumerate(names):
    print("%-14s %8.1f k-ticks per wave  %5.1f %%" % (n, v[k] / waves / 1e3, 100 * v[k] / v[0]))
print("ticks per ip trip %.0f, per sens trip %.0f, per pull %.0f" % (v[5] / max(v[9], 1), v[6] / max(v[10], 1), v[4] / max(v[8], 1) * 1.0))
