"""Oracle (test infrastructure only): MPC-loop glue between two newton_solve! calls -
rot_n_stride!, update_window! of /root/reference/src/controller/mpc_utils.jl:1-101 and
/root/reference/src/controller/policy.jl:136-141,154-171 (policy: shift the controller's copy of the
reference trajectory by one step, continue it with a constant stride, advance the window)."""
import numpy as np

from .dims import Dims
from .newton import Traj


def rotate(traj: Traj):
    """rotate!, mpc_utils.jl:7-41: every array moves one step to the left, the first entry goes last."""
    for a in (traj.q, traj.u, traj.w, traj.gamma, traj.b, traj.theta):
        first = a[0].copy()
        a[:-1] = a[1:]
        a[-1] = first


def mpc_stride(dims: Dims, traj: Traj, stride):
    """mpc_stride!, mpc_utils.jl:79-101: the last two configurations repeat the first two, shifted by
    `stride`; theta of the steps that read them is refreshed (q0 / q1 slices only)."""
    H = traj.H
    for t in (H + 1, H + 2):                      # 1-based as in the reference
        traj.q[t - 1] = traj.q[t - H - 1] + stride
        for tau in (t - 2, t - 3):                # update_theta!(traj, t-2), update_theta!(traj, t-3)
            traj.theta[tau - 1, dims.iq0] = traj.q[tau - 1]
            traj.theta[tau - 1, dims.iq1] = traj.q[tau]


def rot_n_stride(dims: Dims, traj: Traj, stride):
    """rot_n_stride!, mpc_utils.jl:1-5."""
    rotate(traj)
    mpc_stride(dims, traj, stride)


def update_window(window, H_ref):
    """update_window!, policy.jl:162-171, on 0-based knot indices."""
    return (np.asarray(window) + 1) % H_ref
