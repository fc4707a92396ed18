"""Round 6 device tests.

* `cimpc_mpc_solve` (one C call per MPC step: window, reference, altitude, q0, q1 in - u1 / iterations / residual and the
  trajectory out) against the five-call sequence it replaces (cimpc_set_altitude, cimpc_set_window, cimpc_set_reference,
  cimpc_newton_solve, cimpc_get_trajectory): bit-identical results over a warm-started loop whose window moves every step - the
  call sequence of the drop-in under the reference's unchanged policy (/root/reference/src/controller/policy.jl:113-142).
"""
import numpy as np
import pytest

from oracle import synth

from common import make_case, make_solver

pytestmark = pytest.mark.gpu


def _shifted(d, prob, rollouts, H, k):
    """Rollout inputs k MPC steps later: phase + k windows / references of the same problem (new objects, same seeds)."""
    out = []
    H_ref = prob["z0"].shape[0]
    for b, (window, ref, q0, q1) in enumerate(rollouts):
        ph = (int(window[0]) + k) % H_ref
        w2, r2, _, _ = synth.make_rollout(d, prob, H, phase=ph, seed=1000 + b, perturb=0.0)
        out.append((w2, r2))
    return out


@pytest.mark.parametrize("model,mode,B", [("quadruped", 0, 3), ("hopper", 0, 1), ("pushbot", 1, 2)])
def test_mpc_solve_equals_the_five_call_sequence(model, mode, B):
    H, H_ref = 8, 12
    d, prob, tabs, rollouts = make_case(model, mode, H_ref=H_ref, H=H, B=B, seed=3, perturb=1e-2)
    obj = synth.make_objective(d, H)
    sa = make_solver(d, prob, rollouts, H, obj=obj)
    sb = make_solver(d, prob, rollouts, H, obj=obj)
    q0 = np.stack([r[2] for r in rollouts]); q1 = np.stack([r[3] for r in rollouts])
    rng = np.random.default_rng(0)
    for k in range(4):
        sh = _shifted(d, prob, rollouts, H, k)
        win = np.stack([w for (w, _) in sh]) + 1
        ref = {f: np.stack([getattr(r, f) for (_, r) in sh]) for f in ("q", "u", "w", "gamma", "b", "theta")}
        alt = 1e-3 * rng.standard_normal((B, d.nc)) if k >= 2 else None
        # (a) the five calls
        if alt is not None:
            sa.set_altitude(alt)
        sa.set_window(win)
        sa.set_reference(ref["q"], ref["u"], ref["w"], ref["gamma"], ref["b"], ref["theta"])
        u1a, ita, rna = sa.newton_solve(q0, q1, warm_start=k > 0)
        ta = sa.trajectory()
        # (b) one call
        u1b, itb, rnb, tb = sb.mpc_solve(q0, q1, window=win, reference=ref, alt=alt, warm_start=k > 0, which=("q", "u", "gamma", "b", "nu"))
        assert np.array_equal(ita, itb) and np.array_equal(u1a, u1b) and np.array_equal(rna, rnb)
        for f in ("q", "u", "gamma", "b", "nu"):
            assert np.array_equal(ta[f], tb[f]), f
        q0, q1 = q1, ta["q"][:, 2].copy()
    # unchanged window / reference: NULL inputs keep what is resident
    u1a, ita, rna = sa.newton_solve(q0, q1, warm_start=True)
    u1b, itb, rnb, _ = sb.mpc_solve(q0, q1, warm_start=True, which=())
    assert np.array_equal(ita, itb) and np.array_equal(u1a, u1b) and np.array_equal(rna, rnb)
    sa.close(); sb.close()


def test_mpc_solve_validates_its_window():
    from contactimplicitmpc.jl_amd import CimpcError
    H, H_ref = 6, 8
    d, prob, tabs, rollouts = make_case("quadruped", 0, H_ref=H_ref, H=H, B=2, seed=1)
    s = make_solver(d, prob, rollouts, H, obj=synth.make_objective(d, H))
    q0 = np.stack([r[2] for r in rollouts]); q1 = np.stack([r[3] for r in rollouts])
    bad = np.stack([w for (w, _, _, _) in rollouts]) + 1
    bad[1, 3] = H_ref + 1
    with pytest.raises(CimpcError):
        s.mpc_solve(q0, q1, window=bad)
    s.close()
