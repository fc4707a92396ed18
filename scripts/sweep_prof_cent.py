"""Diagnostic: clock accounting of the 32-lane lock-step sweep per wave on BASELINE configs[4]'s inputs (library built with
-DCIMPC_SWEEP_PROF, CIMPC_LIB=...).  usage: CIMPC_LIB=.../libcimpc_sprof.so python scripts/sweep_prof_cent.py [B]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions, _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = 60
I = bench.centroidal_payload_inputs(B, H)
m, P, kappa, ro = I["m"], I["P"], I["kappa"], I["rollouts"]
s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=kappa, r_tol=1e-4),
                newton_opts=NewtonOptions(kappa=kappa, r_tol=3e-5, max_iter=5))
for t in range(P.H):
    s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
s.set_objective(I["Q"], I["R"])
s.set_window(np.stack([r["window"] for r in ro]) + 1)
s.set_reference(*(np.stack([r[k] for r in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")))
q0 = torch.tensor(np.stack([r["q0"] for r in ro]), dtype=torch.float64, device="cuda")
q1 = torch.tensor(np.stack([r["q1"] for r in ro]), dtype=torch.float64, device="cuda")
lib = _lib.load()
f = lib.cimpc_debug_sweep_prof_centroidal
f.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 16)()
s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
f(buf)
steps = 3
for _ in range(steps):
    s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
f(buf)
v = np.array(list(buf), dtype=np.float64)
names = ["lifetime", "barriers+pick", "stage", "end-of-solve", "pull", "ip trips", "sens trips"]
waves = v[7]
print("centroidal H = 60, B = %d; waves (all launches of %d steps): %d, pulls %d, ip trips %d, sens trips %d" % (B, steps, waves, v[8], v[9], v[10]))
for k, n in enumerate(names):
    print("%-14s %8.1f k-ticks per wave  %5.1f %%" % (n, v[k] / waves / 1e3, 100 * v[k] / v[0]))
print("ticks per ip trip %.0f, per sens trip %.0f, per pull %.0f  (s_memtime: 100 MHz)" % (v[5] / max(v[9], 1), v[6] / max(v[10], 1), v[4] / max(v[8], 1) * 1.0))
