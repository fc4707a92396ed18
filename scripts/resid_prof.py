"""Diagnostic: clock accounting of the two launches of the decision stage (resid_slot_kernel, resid_decide_kernel); library built
with -DCIMPC_RESID_PROF, selected through CIMPC_LIB.
usage: make -C contactimplicitmpc/jl_amd/csrc EXTRA=-DCIMPC_RESID_PROF OUT=../libcimpc_prof.so BUILD=build_prof
       CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/libcimpc_prof.so python scripts/resid_prof.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions, _lib  # noqa: E402

H, H_ref, B = 40, 60, int(os.environ.get("B", "512"))
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
f = lib.cimpc_debug_resid_prof
f.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * 64)()
s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
f(buf)
s.profile_enable(True); s.profile_reset()
steps = 4
for _ in range(steps):
    s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
f(buf)
pr = s.profile_read()
v = np.array(list(buf), dtype=np.float64).reshape(4, 16)
print("HIP-event time of the stage per round: %.1f us (%d rounds)" % (1e3 * pr["resid_ms"] / max(pr["resid_launches"], 1), pr["resid_launches"]))
names = [["", "entry + effective-slot table", "residual rows", "norm"],
         ["", "entry checks", "decision", "statistics", "next candidates: apply_step", "next candidates: enqueue", "accept: apply_step", "res copy",
          "dz_good copy + scalars"]]
for k, kn in enumerate(("resid_slot_kernel", "resid_decide_kernel")):
    n = max(v[k, 0], 1)
    print("%s: %d workgroups (%.1f per round), mean lifetime %.2f us, longest %.2f us" % (kn, v[k, 0], v[k, 0] / max(pr["resid_launches"], 1), v[k, 13] / n / 100, v[k, 14] / 100))
    for j in range(1, len(names[k])):
        print("    %-34s %7.2f us per workgroup  %5.1f %%" % (names[k][j], v[k, j] / n / 100, 100 * v[k, j] / max(v[k, 13], 1)))
    ns = max(v[2 + k, 0], 1)
    print("  slow workgroups (> 30 us): %d (%.1f per round), mean lifetime %.2f us" % (v[2 + k, 0], v[2 + k, 0] / max(pr["resid_launches"], 1), v[2 + k, 13] / ns / 100))
    for j in range(1, len(names[k])):
        print("    %-34s %7.2f us per slow workgroup" % (names[k][j], v[2 + k, j] / ns / 100))
