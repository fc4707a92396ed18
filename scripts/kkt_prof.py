"""Diagnostic: phase clocks of the condensed KKT recursion (library built with -DCIMPC_KKT_PROF, selected through CIMPC_LIB;
one-wave kernel: CIMPC_KKT_PIPE=0).  usage: CIMPC_KKT_PIPE=0 CIMPC_LIB=.../libcimpc_kprof.so python scripts/kkt_prof.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from common import make_case, make_solver  # noqa: E402
from oracle import synth  # noqa: E402
import torch  # noqa: E402,F401

H = 40
d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=60, H=H, B=1, seed=3)
obj = synth.make_objective(d, H, kind="quadruped")
from contactimplicitmpc.jl_amd import NewtonOptions  # noqa: E402
s = make_solver(d, prob, rollouts, H, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=1e-9, max_iter=3))
q = np.stack([r.q for (_, r, _, _) in rollouts]); th = np.stack([r.theta for (_, r, _, _) in rollouts])
s.implicit_dynamics(q, th)
r = np.random.default_rng(0).standard_normal((1, s.N))
if os.environ.get("CIMPC_KKT_PIPE", "0") != "0":      # the pipelined kernel runs inside the Newton loop only (B = 1: every KKT solve)
    for _ in range(2):
        u1, it, rn = s.newton_solve(np.stack([ro[2] for ro in rollouts]), np.stack([ro[3] for ro in rollouts]))
    print("newton_solve:", it, rn, s.stats())
else:
    for _ in range(3):
        s.kkt_solve(r, 10.0)
import torch
buf = torch.zeros(32, dtype=torch.int64)
# the clocks sit in the handle's statistics buffer (NewtonDev::stats[8..23]); read them through the debug accessor
lib = s.lib
lib.cimpc_debug_read_stats.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.c_int]
out = (C.c_longlong * 32)()
lib.cimpc_debug_read_stats(s.h, out, 32)
v = np.array(list(out), dtype=np.float64)[8:24]
names = {1: "A: operands -> LDS", 2: "A: T0, T1, T2 products", 3: "A: Y_ii, Y_i,i-1, L2, beta", 4: "B: Y1 update + L1", 5: "B: L0 L0^T input + rhs",
         6: "B: Cholesky + inverse", 7: "B: y + spill", 8: "backward pass + recovery"}
if os.environ.get("CIMPC_KKT_PIPE", "0") != "0":      # pipelined kernel: per wave, time in its stage / at the tick barrier
    TK = 2400.0
    for w, n in enumerate(("stage A (wave 0)", "stage B (wave 1)", "stage C (wave 2)")):
        print("  %-18s in stage %7.2f us   at the barrier %7.2f us   (per tick %.2f / %.2f us)" % (n, v[9 + 2 * w] / TK, v[10 + 2 * w] / TK, v[9 + 2 * w] / TK / (H + 2), v[10 + 2 * w] / TK / (H + 2)))
    print("  backward pass + recovery (wave 0) %7.2f us" % (v[8] / TK))
    sys.exit(0)
tot = v[1:9].sum()
TK = 2400.0      # s_memtime ticks per microsecond (shader clock)
print("one KKT solve, H = %d: %.1f us" % (H, tot / TK))
for k, n in names.items():
    print("  %-30s %8.2f us  %5.1f %%   (%.2f us per step)" % (n, v[k] / TK, 100 * v[k] / tot, v[k] / TK / H))
