"""Are the per-solve statistics sane at every step?  (round 4: a B = 128 knob run printed a garbage `sweeps` sum)
usage: CIMPC_LIB=<lib> python scripts/dbg/stats_check.py B [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions  # noqa: E402

B = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
H, H_ref = 40, 60
d, prob, obj, ro = bench.build_inputs(B, H, H_ref, 1234, 0.05)
s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]),
                newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-4, max_iter=5))
for t in range(H_ref):
    s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
s.set_objective(obj.q, obj.u)
s.set_window(np.stack([w for (w, _, _, _) in ro]) + 1)
s.set_reference(*(np.stack([getattr(r, k) for (_, r, _, _) in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")))
q0 = torch.tensor(np.stack([r[2] for r in ro]), dtype=torch.float64, device="cuda")
q1 = torch.tensor(np.stack([r[3] for r in ro]), dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
if os.environ.get("STATS_PROF"):
    s.profile_enable(True)
bad = 0
first = None
import ctypes as C


def dump():
    lib = s.lib
    if not hasattr(lib, "cimpc_debug_dump_stats"):
        return
    out = (C.c_longlong * (B * 4))(); addr = (C.c_ulonglong * 8)()
    lib.cimpc_debug_dump_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_ulonglong)]
    lib.cimpc_debug_dump_stats(s.h, out, addr)
    a = np.array(out).reshape(B, 4)
    badrows = [(b, [hex(int(x)) for x in a[b]]) for b in range(B) if (a[b] < 0).any() or (a[b] > 10 ** 7).any()]
    print("   addresses kkt_list slot_list counters stats ro_sweeps ro_ip_iters ro_ip_fail nlog:", [hex(int(x)) for x in addr])
    print("   corrupted records:", badrows[:12])

for k in range(steps):
    s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
    st = s.stats(); rc = s.rollout_counters()
    if first is None:
        first = st
    ok = st["newton_iters"] == first["newton_iters"] and 0 < st["sweeps"] < 100 * B and 0 < st["ip_solves"] < 10 ** 9 and 0 < st["ip_iters"] < 10 ** 10
    if not ok:
        bad += 1
        print("step", k, "BAD", st, "ro_sweeps sum", int(rc["sweeps"].sum()))
        if st["sweeps"] > 100 * B or st["ip_solves"] > 10 ** 9:
            dump()
print(os.environ.get("CIMPC_LIB", "default"), "B", B, "steps", steps, "bad", bad, "first", first)
