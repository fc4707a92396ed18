"""Experiment: the headline batch (512 rollouts) as k concurrent sub-batches, one handle + one host thread each.
The lock-step rounds are latency-bound (a round costs ~0.4 ms whatever the number of active rollouts), so independent
sub-batches can fill each other's gaps.  usage: python scripts/exp_subbatch.py [k ...]"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions  # noqa: E402

H, H_ref, B = 40, 60, 512
d, prob, obj, ro = bench.build_inputs(B, H, H_ref, seed=1234, perturb=0.05)


def make(ro_):
    n = len(ro_)
    s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=n, mode=0, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]),
                    newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-4, max_iter=5))
    for t in range(H_ref):
        s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
    s.set_objective(obj.q, obj.u)
    s.set_window(np.stack([w for (w, _, _, _) in ro_]) + 1)
    s.set_reference(*(np.stack([getattr(r, k) for (_, r, _, _) in ro_]) for k in ("q", "u", "w", "gamma", "b", "theta")))
    q0 = torch.tensor(np.stack([r[2] for r in ro_]), dtype=torch.float64, device="cuda")
    q1 = torch.tensor(np.stack([r[3] for r in ro_]), dtype=torch.float64, device="cuda")
    return s, q0, q1


for k in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    n = B // k
    hs = [make(ro[j * n:(j + 1) * n]) for j in range(k)]
    torch.cuda.synchronize()
    steps = 10

    def work(j, reps):
        s, q0, q1 = hs[j]
        for _ in range(reps):
            s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)

    for reps in (3, steps):
        th = [threading.Thread(target=work, args=(j, reps)) for j in range(k)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
    its = sum(int(h[0].newton_info()[1].sum()) for h in hs)
    print("sub-batches %d x %d rollouts: %.2f ms per 512-rollout step, %.0f MPC steps/s, newton iters %d" % (k, n, 1e3 * dt / steps, B * steps / dt, its))
    for h in hs:
        h[0].close()
