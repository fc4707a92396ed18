"""How often the interior-point line search of the spec runs out of its `max_ls` evaluations - the one
place where the carried violations describe a point the iterate has already left.  Measured on the bench's own inputs and
on deliberately hard instances: it never happens inside a solve that converges."""
import numpy as np

from common import make_case
from oracle import ip as oip


def _sweep(d, tabs, rollouts, opts):
    n = exhausted_solves = failed = exhausted_in_converged = 0
    for (window, ref, q0, q1) in rollouts:
        tr = ref.copy()
        tr.q[0], tr.q[1] = q0, q1
        tr.update_theta(d, 0); tr.update_theta(d, 1)
        for i in range(len(window) - 2):
            z = oip.z_initialize(d, tr.q[i + 2])
            c = {}
            status, it, dz = oip.interior_point_solve(tabs[int(window[i])], z, tr.theta[i], opts, counters=c)
            n += 1
            failed += (not status)
            if c.get("ls_exhausted", 0):
                exhausted_solves += 1
                exhausted_in_converged += bool(status)
    return n, exhausted_solves, failed, exhausted_in_converged


def test_line_search_exhaustion_is_confined_to_failing_solves():
    import bench
    d, prob, obj, ro = bench.build_inputs(24, 40, 60, seed=1234, perturb=0.05)       # the headline batch's first rollouts
    from oracle import lcp
    tabs = [lcp.LinTable(d, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t]) for t in range(60)]
    n, ex, failed, ex_conv = _sweep(d, tabs, ro, oip.IPOptions(kappa_tol=prob["kappa"]))
    assert n == 24 * 40 and ex_conv == 0
    # hard instances: perturbations ten times larger (some solves fail)
    d2, prob2, tabs2, hard = make_case("quadruped", 0, H_ref=10, H=8, B=24, seed=21, perturb=0.5)
    n2, ex2, failed2, ex_conv2 = _sweep(d2, tabs2, hard, oip.IPOptions(kappa_tol=prob2["kappa"]))
    assert ex_conv2 == 0, (n2, ex2, failed2, ex_conv2)
    print("bench inputs: %d solves, %d with an exhausted search, %d failed; hard: %d / %d / %d" % (n, ex, failed, n2, ex2, failed2))
