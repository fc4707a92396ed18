import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
np.seterr(all="ignore")
import bench
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
H, H_ref, B = 40, 60, 512
d, prob, obj, ro = bench.build_inputs(B, H, H_ref, seed=1234, perturb=0.05)
s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]),
                newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-4, max_iter=5))
for t in range(H_ref):
    s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
s.set_objective(obj.q, obj.u)
s.set_window(np.stack([w for (w, _, _, _) in ro]) + 1)
s.set_reference(np.stack([r.q for (_, r, _, _) in ro]), np.stack([r.u for (_, r, _, _) in ro]), np.stack([r.w for (_, r, _, _) in ro]),
                np.stack([r.gamma for (_, r, _, _) in ro]), np.stack([r.b for (_, r, _, _) in ro]), np.stack([r.theta for (_, r, _, _) in ro]))
q = np.stack([r.q for (_, r, _, _) in ro]).copy(); th = np.stack([r.theta for (_, r, _, _) in ro]).copy()
rng = np.random.default_rng(0)
for scale in (0.0, 0.02, 0.1):
    qq = q + rng.uniform(-scale, scale, q.shape)
    tt = th.copy(); tt[:, :, :22] += rng.uniform(-scale, scale, (B, H, 22))
    s.profile_enable(True); s.profile_reset()
    out = s.implicit_dynamics(qq, tt)
    p = s.profile_read()
    it = out["iters"].ravel(); st = out["status"].ravel()
    print("scale", scale, "sweep ms %.3f" % p["ip_sweep_ms"], "fail", int((st == 0).sum()), "iters mean %.2f max %d" % (it.mean(), it.max()),
          "hist>=10:", np.bincount(it, minlength=101)[10:].nonzero()[0][:20] + 10, "counts", np.bincount(it, minlength=101)[10:][np.bincount(it, minlength=101)[10:].nonzero()[0][:20]])
