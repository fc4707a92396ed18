"""One-off stress: lock-step rounds vs single launch vs hybrid (forced early hand-off) must agree bit for bit
(counters, trajectories) over several seeds / models / perturbation levels.
The claim holds between schedules that run the SAME KKT kernels: from H = 24 on the rounds take the two-ended forms (twisted / duo) where the
persistent kernel keeps the one-ended job, the two agree to 1e-12 and not to the bit, and a chaotic rollout (quadruped H = 40) then changes its
counters - so the script pins the one-ended kernels (CIMPC_KKT_TWISTED=0) unless STRESS_TWO_ENDED=1 asks for the default choice (round 6, final code:
0 mismatches pinned, the H = 40 case differs unpinned)."""
import os, sys
if os.environ.get("STRESS_TWO_ENDED", "0") != "1": os.environ["CIMPC_KKT_TWISTED"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from common import make_case, make_solver
from contactimplicitmpc.jl_amd import synthetic as synth
from contactimplicitmpc.jl_amd import NewtonOptions
bad = 0
for (model, H, H_ref, B, pert, seed) in [("quadruped", 20, 30, 200, 0.05, 1), ("quadruped", 12, 16, 300, 0.1, 2), ("hopper", 20, 24, 256, 0.05, 3),
                                          ("quadruped", 40, 60, 150, 0.08, 4), ("pushbot", 10, 12, 180, 0.05, 5)]:
    d, prob, tabs, ro = make_case(model, 0, H_ref=H_ref, H=H, B=B, seed=seed, perturb=pert)
    obj = synth.make_objective(d, H, kind=model)
    q0 = np.stack([r[2] for r in ro]); q1 = np.stack([r[3] for r in ro])
    outs = []
    for (mode, tail) in (("0", None), ("1", None), ("2", str(B // 2)), ("2", "40")):
        os.environ["CIMPC_ASYNC"] = mode
        if tail: os.environ["CIMPC_ASYNC_TAIL"] = tail
        else: os.environ.pop("CIMPC_ASYNC_TAIL", None)
        s = make_solver(d, prob, ro, H, obj=obj, newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-4, max_iter=5))
        u1, it, rn = s.newton_solve(q0, q1)
        outs.append((u1, it, s.trajectory(), s.rollout_counters(), s.stats()["rounds"]))
        s.close()
    ref = outs[0]
    for k, o in enumerate(outs[1:]):
        ok = (np.array_equal(o[1], ref[1]) and all(np.array_equal(o[3][c], ref[3][c]) for c in ("sweeps", "ip_iters", "ip_failures"))
              and np.allclose(o[0], ref[0], rtol=1e-12, atol=1e-14) and np.allclose(o[2]["q"], ref[2]["q"], rtol=1e-12, atol=1e-14))
        bad += not ok
        print(model, H, B, "variant", k + 1, "rounds", o[4], "vs", ref[4], "OK" if ok else "MISMATCH", "fails", int(ref[3]["ip_failures"].sum()), "iters hist", np.bincount(ref[1]))
print("mismatches:", bad)
