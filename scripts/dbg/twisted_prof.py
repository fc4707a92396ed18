"""Diagnostic: constant-rate clock stamps of the two chains of the twisted KKT solve (library built with -DCIMPC_KKT_TWPROF, selected
through CIMPC_LIB): where the two workgroups of a rollout spend their time and how long the hand-overs take.
usage: CIMPC_LIB=contactimplicitmpc/jl_amd/libcimpc_twprof.so python scripts/dbg/twisted_prof.py [model H H_ref]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import make_case, make_solver  # noqa: E402
from oracle import synth  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "quadruped"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 40
H_ref = int(sys.argv[3]) if len(sys.argv) > 3 else 60
d, prob, tabs, rollouts = make_case(model, 0, H_ref=H_ref, H=H, B=1, seed=3)
obj = synth.make_objective(d, H, kind=model if model in ("quadruped", "hopper") else "quadruped")
s = make_solver(d, prob, rollouts, H, obj=obj)
q = np.stack([r.q for (_, r, _, _) in rollouts]); th = np.stack([r.theta for (_, r, _, _) in rollouts])
s.implicit_dynamics(q, th)
r = np.random.default_rng(0).standard_normal((1, s.N))
NEWTON = os.environ.get("TW_NEWTON") == "1"      # the KKT launches of newton_solve! (finish = 1: the last chain starts the line search)
if NEWTON:
    from contactimplicitmpc.jl_amd import NewtonOptions
    s.close()
    s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-9, max_iter=3))
s.lib.cimpc_debug_read_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
rows = []
for rep in range(6):
    if NEWTON:
        s.newton_solve(np.stack([ro[2] for ro in rollouts]), np.stack([ro[3] for ro in rollouts]))
    else:
        s.kkt_solve(r, 10.0)
    out = (C.c_longlong * 64)()
    s.lib.cimpc_debug_read_stats(s.h, out, 64)
    rows.append(np.array(list(out), dtype=np.float64))
v = rows[-1]
TK = 100.0      # wall_clock64 ticks per microsecond
top, bot, w = v[8:16], v[16:24], v[24:28]
t0 = min(top[0], bot[0])
names = ["enter", "LDS clear done", "forward pass done", "(bottom) middle dnu received", "backward pass done", "recovery done", "finish counter", "line search started (last chain)"]
print("%s H = %d, twisted KKT solve (us since the first chain entered); twisted launches: %d" % (model, H, s.kkt_twisted()))
for j, n in enumerate(names):
    print("  %-32s top %8.2f   bottom %8.2f" % (n, (top[j] - t0) / TK, (bot[j] - t0) / TK))
print("  kernel entry: top %.2f, bottom %.2f us" % ((v[49] - t0) / TK, (v[48] - t0) / TK))
print("  top chain, wave 1 at row m: waits for the traces from %.2f to %.2f us" % ((w[0] - t0) / TK, (w[1] - t0) / TK))

def hw(x):
    x = int(x)
    return "wave %d simd %d cu %d sh %d se %d" % (x & 15, (x >> 4) & 3, (x >> 8) & 15, (x >> 12) & 1, (x >> 13) & 7)
print("  top chain on xcc %d, %s;  bottom chain on xcc %d, %s" % (int(v[29]) & 15, hw(v[28]), int(v[31]) & 15, hw(v[30])))
if v[32:48].any():      # (-DCIMPC_KKT_WPROF as well) per wave: shader-clock cycles inside its stage / at the tick barrier
    for c, nm in ((0, "top"), (1, "bottom")):
        w_ = v[32 + 8 * c: 40 + 8 * c]
        print("  %-6s stage A %7.1f + %6.1f us at the barrier | stage B %7.1f + %6.1f | stage C %7.1f + %6.1f  (2.4 GHz assumed)"
              % (nm, w_[0] / 2400, w_[1] / 2400, w_[2] / 2400, w_[3] / 2400, w_[4] / 2400, w_[5] / 2400))
