"""CPU oracle for the ContactImplicitMPC.jl per-MPC-step solver path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path
(`contactimplicitmpc/jl_amd`, `include/`, the HIP library) may import, link or
execute anything under `oracle/`.  Allowed importers: `tests/`,
`__graft_entry__.smoke()`, `bench.py`'s `cpu_baseline` leg and the validation
scripts under `scripts/` (closed-loop runs: CPU plant, tracking_error) - and
there only as the checker / the timed CPU baseline, never as the thing shipped.

Modules: lcp.py, ip.py (incl. knot_store: the per-knot sensitivity memory of
im_traj.ip[t]), newton.py, mpc.py (the path), cimpc_ref.c / cref.py (C port,
CPU baseline), dims.py (the checker's OWN index layouts; newton.py holds its own
trajectory / objective containers), synth.py (re-export of the seeded synthetic
INPUT generator - it produces the inputs both sides of a parity test are fed
with and holds no solver arithmetic), plant.py (simulator step + closed loop, CPU: planar
chains, hopper_2D, 3-D centroidal_quadruped), banded.py (the banded KKT kernel's
algorithm in numpy).

Parity status (see DESIGN.md section "Oracle"):
  * reference-side callbacks (rlin!, rzlin!, Schur/MGS-QR, linear_solve!,
    sensitivities) and the Newton layer (residual!, jacobian!, update_traj!,
    newton_solve!) follow `/root/reference/src` line by line and are pinned by
    re-running the reference's own test constructions
    (tests/test_oracle_reference_constructions.py).
  * the interior-point iteration itself lives in RoboDojo.jl 0.1.3, which is
    NOT present under /root/reference and cannot be executed here (no Julia):
    **IP iterate-level parity: unpinned** - the loop in `ip.py` is
    build-defined, modelled on RoboDojo 0.1.3, pinned only at the level the
    reference's tests pin it (converged answer to tolerance).  That level IS
    exercised on the reference's own data: test/controller/implicit_dynamics.jl
    (quadruped gait2.jld2, |dq2|_inf < 1e-2 at every knot) is reproduced by
    tests/test_real_models.py through the true model linearization
    (contactimplicitmpc/jl_amd/lcp_models.py + gait_io.py), and the closed
    loop of test/controller/mpc_quadruped.jl (plant = oracle/plant.py) lands
    within 2 % of the tracking errors that test records (tests/test_closed_loop.py,
    scripts/closed_loop.py).
"""
from .dims import Dims  # noqa: F401
