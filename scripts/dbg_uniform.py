import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
np.seterr(all="ignore")
import bench
from oracle import synth
from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
H, H_ref = 40, 60
CASES = ((512, True), (512, False), (2048, True), (2048, False)) if len(sys.argv) < 2 else ((int(sys.argv[1]), True),)
for B, uniform in CASES:
    d, prob, obj, ro = bench.build_inputs(B, H, H_ref, seed=1234, perturb=0.02)
    if uniform:
        ro = [synth.make_rollout(d, prob, H, phase=0, seed=1, perturb=0.02)] * B
    s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]))
    for t in range(H_ref):
        s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
    s.set_window(np.stack([w for (w, _, _, _) in ro]) + 1)
    q = np.stack([r.q for (_, r, _, _) in ro]).copy(); th = np.stack([r.theta for (_, r, _, _) in ro]).copy()
    for (w, r, q0, q1), qq in zip(ro, q):
        qq[0] = q0; qq[1] = q1
    th[:, 0, :11] = q[:, 0]; th[:, 0, 11:22] = q[:, 1]; th[:, 1, :11] = q[:, 1]
    s.implicit_dynamics(q, th)
    s.profile_enable(True); s.profile_reset()
    for _ in range(3):
        out = s.implicit_dynamics(q, th)
    p = s.profile_read()
    it = out["iters"].ravel()
    print("B", B, "uniform", uniform, "sweep ms %.3f" % (p["ip_sweep_ms"] / p["ip_sweep_launches"]), "launches", p["ip_sweep_launches"],
          "ns/solve %.1f" % (1e6 * p["ip_sweep_ms"] / 3 / (B * H)), "iters mean %.2f max %d" % (it.mean(), it.max()))
