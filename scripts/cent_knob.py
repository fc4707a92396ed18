"""centroidal payload configuration (BASELINE configs[4] inputs) at batch size B under the current environment:
python scripts/cent_knob.py <B> [backend]   -> ms per batch step (fp64 condensed KKT by default)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions  # noqa: E402

B = int(sys.argv[1]); backend = int(sys.argv[2]) if len(sys.argv) > 2 else 0
H = 60
I = bench.centroidal_payload_inputs(B, H)
m, P, kappa, ro = I["m"], I["P"], I["kappa"], I["rollouts"]
s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=kappa, r_tol=1e-4),
                newton_opts=NewtonOptions(kappa=kappa, r_tol=3e-5, max_iter=5, kkt_backend=backend))
for t in range(P.H):
    s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
s.set_objective(I["Q"], I["R"])
s.set_window(np.stack([r["window"] for r in ro]) + 1)
s.set_reference(*(np.stack([r[k] for r in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")))
q0 = torch.tensor(np.stack([r["q0"] for r in ro]), dtype=torch.float64, device="cuda")
q1 = torch.tensor(np.stack([r["q1"] for r in ro]), dtype=torch.float64, device="cuda")
s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
s.profile_enable(True); s.profile_reset()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 3
for _ in range(n):
    s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
pr = s.profile_read()
print("[%s] B=%d backend %d: %.2f ms/step (%.0f steps/s) sweep %.2f kkt %.2f resid %.2f async %.2f" % (os.environ.get("TAG", ""), B, backend, 1e3 * dt, B / dt,
      pr["ip_sweep_ms"] / n, pr["kkt_ms"] / n, pr["resid_ms"] / n, pr["async_ms"] / n))
