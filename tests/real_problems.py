"""Real reference problems for the tests: the reference's gait files (tests/golden/gaits/*.jld2 - data files of the
reference: src/dynamics/quadruped/gaits/gait2.jld2 is the input of test/controller/implicit_dynamics.jl and
test/controller/mpc_quadruped.jl; examples/centroidal_quadruped/reference/inplace_trot_v7.jld2 of
examples/centroidal_quadruped/continuous_trot.jl) through the model restatements of
contactimplicitmpc/jl_amd/lcp_models.py."""
import functools
import os

import numpy as np

from oracle import lcp
from oracle.dims import Dims
from oracle.newton import Traj

HERE = os.path.dirname(os.path.abspath(__file__))
GAITS = {
    "quadruped": ("quadruped", os.path.join(HERE, "golden", "gaits", "quadruped_gait2.jld2")),
    "centroidal": ("centroidal_quadruped", os.path.join(HERE, "golden", "gaits", "centroidal_inplace_trot_v7.jld2")),
    "flamingo": ("flamingo", os.path.join(HERE, "golden", "gaits", "flamingo_gait_forward_36_4.jld2")),
    # src/dynamics/hopper_2D/gaits/gait_forward.jld2 (examples/hopper/flat.jl:15-18): a `:joint_traj` file - a serialized ContactTraj
    # whose z / θ the reference linearizes AS THEY ARE (implicit_dynamics.jl:37-50).  Caveat (tests/test_real_models.py): the file
    # predates the shipped hopper model - its leg-length row is off by 4e-3 and the slack part of z has an older ordering - so it
    # is the reference's PROBLEM (what examples/hopper/flat.jl solves), not a pin of the hopper dynamics.
    "hopper": ("hopper_2D", os.path.join(HERE, "golden", "gaits", "hopper_gait_forward.jld2")),
}
JOINT_TRAJ = {"hopper"}


@functools.lru_cache(maxsize=None)
def real_problem(which: str, kappa: float, update_friction: bool = False, mode: int = 0):
    """-> (Dims, ReferenceProblem, prob dict in the layout of oracle.synth.make_problem, LinTables)."""
    from contactimplicitmpc.jl_amd import gait_io, lcp_models
    name, path = GAITS[which]
    model = lcp_models.MODELS[name]()
    if which in JOINT_TRAJ:
        P = lcp_models.reference_problem_from_traj(model, gait_io.load_joint_traj(path), kappa)
    else:
        P = lcp_models.reference_problem(model, gait_io.load_gait(path), kappa, update_friction)
    d = Dims(nq=model.nq, nu=model.nu, nw=model.nw, nc=model.nc, nb=model.nb, mode=mode)
    prob = dict(z0=P.z, th0=P.theta, r0=P.r0, rz0=P.rz0, rth0=P.rth0, kappa=kappa, q_ref=P.q, u_ref=P.u, w_ref=P.w,
                gamma_ref=P.gamma, b_ref=P.b, stride=lcp_models.get_stride(model, P.q), P=P)
    tabs = [lcp.LinTable(d, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t]) for t in range(P.H)]
    return d, P, prob, tabs


def real_rollout(d: Dims, prob, H: int, phase: int, seed: int, perturb: float = 0.0, vel_perturb: float = 0.05):
    """`lcp_models.make_rollout` in the (window, ref, q0, q1) form of oracle.synth.make_rollout."""
    from contactimplicitmpc.jl_amd import lcp_models
    r = lcp_models.make_rollout(prob["P"], H, phase, seed, perturb, vel_perturb)
    ref = Traj(q=r["q"], u=r["u"], w=r["w"], gamma=r["gamma"], b=r["b"], theta=r["theta"])
    chk = ref.copy()
    chk.update_theta(d)
    assert np.array_equal(chk.theta, ref.theta)
    return r["window"], ref, r["q0"], r["q1"]
