"""Diagnostic: clock stamps of the two chains of the twisted BANDED KKT solve (library built with -DCIMPC_KKT_TWPROF, CIMPC_LIB=...):
usage: CIMPC_LIB=contactimplicitmpc/jl_amd/libcimpc_twprof.so python scripts/dbg/twisted_banded_prof.py [model H H_ref]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from common import make_case, make_solver  # noqa: E402
from oracle import synth  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "centroidal"
H = int(sys.argv[2]) if len(sys.argv) > 2 else 60
H_ref = int(sys.argv[3]) if len(sys.argv) > 3 else 71
d, prob, tabs, rollouts = make_case(model, 0, H_ref=H_ref, H=H, B=1, seed=3)
obj = synth.make_objective(d, H, kind=model if model in ("quadruped", "hopper") else "quadruped", velocity=True)
s = make_solver(d, prob, rollouts, H, obj=obj)
q = np.stack([r.q for (_, r, _, _) in rollouts]); th = np.stack([r.theta for (_, r, _, _) in rollouts])
s.implicit_dynamics(q, th)
r = np.random.default_rng(0).standard_normal((1, s.N))
s.lib.cimpc_debug_read_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
for rep in range(5):
    s.kkt_solve(r, 10.0)
    out = (C.c_longlong * 64)()
    s.lib.cimpc_debug_read_stats(s.h, out, 64)
v = np.array(list(out), dtype=np.float64)
TK = 100.0
top, bot = v[8:16], v[16:24]
t0 = min(top[0], bot[0])
names = ["enter", "pre-pass, window fill, first diagonal block", "(top) reaches the trace block", "(top) trace received", "factorisation loop done",
         "back substitution starts", "(bottom) middle x received", "back substitution done"]
print("%s H = %d, twisted banded KKT solve (us since the first chain entered); twisted launches: %d" % (model, H, s.kkt_twisted()))
for j, n in enumerate(names):
    print("  %-46s top %9.2f   bottom %9.2f" % (n, (top[j] - t0) / TK, (bot[j] - t0) / TK))
