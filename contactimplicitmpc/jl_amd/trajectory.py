"""Host-side data containers of the path: problem dimensions / index layouts, the contact
trajectory and the tracking objective - mirrors of the reference's host types
(file:line relative to /root/reference):

  index layouts            src/simulation/index.jl:13-107, 117-178, 187-327, 371-384
  ContactTraj              src/controller/trajectory.jl:1-49, update_theta! :67-82
  copy_traj!               src/controller/newton.jl:105-128
  TrackingObjective / TrackingVelocityObjective   src/controller/objective.jl:3-47

z order  [q2; g1; b1; psi1; s1; eta1; s2], th order [q0; q1; u1; w1; mu; h],
x=q2, y1=[g1;b1;psi1], y2=[s1;eta1;s2]; Newton block sizes: newton_residual.jl:18-54, 69-98.
Pure numpy; no device code.  (The CPU checker re-exports these types so that both sides of a
parity test are built from the same inputs.)
"""
from dataclasses import dataclass

import numpy as np

MODE_CONFIGURATION = 0        # :configuration
MODE_CONFIGURATIONFORCE = 1   # :configurationforce


@dataclass(frozen=True)
class Dims:
    nq: int
    nu: int
    nw: int
    nc: int
    nb: int          # nc * friction_dim(env)
    mode: int = MODE_CONFIGURATION

    # ---- interior-point (per-knot LCP) sizes -------------------------------
    @property
    def nx(self):
        return self.nq

    @property
    def ny(self):
        return 2 * self.nc + self.nb

    @property
    def nz(self):   # index.jl:371-377
        return self.nq + 4 * self.nc + 2 * self.nb

    @property
    def nth(self):  # index.jl:379-384
        return 2 * self.nq + self.nu + self.nw + 2

    # ---- Newton (horizon) sizes ---------------------------------------------
    @property
    def nd(self):
        return self.nq if self.mode == MODE_CONFIGURATION else self.nq + self.nc + self.nb

    @property
    def nr(self):
        return self.nq + self.nu if self.mode == MODE_CONFIGURATION \
            else self.nq + self.nu + self.nc + self.nb

    # ---- index vectors (0-based) ---------------------------------------------
    @property
    def ix(self):
        return np.arange(0, self.nq)

    @property
    def iy1(self):
        return np.arange(self.nq, self.nq + self.ny)

    @property
    def iy2(self):
        return np.arange(self.nq + self.ny, self.nq + 2 * self.ny)

    idyn = ix
    irst = iy1   # the residual uses the same 3-group partition sizes
    ibil = iy2

    # z sub-blocks
    @property
    def ig1(self):
        return np.arange(self.nq, self.nq + self.nc)

    @property
    def ib1(self):
        return np.arange(self.nq + self.nc, self.nq + self.nc + self.nb)

    # theta sub-blocks
    @property
    def iq0(self):
        return np.arange(0, self.nq)

    @property
    def iq1(self):
        return np.arange(self.nq, 2 * self.nq)

    @property
    def iu1(self):
        return np.arange(2 * self.nq, 2 * self.nq + self.nu)

    @property
    def iw1(self):
        return np.arange(2 * self.nq + self.nu, 2 * self.nq + self.nu + self.nw)


# dimension constants of the models named by BASELINE.json configs
# (reference model files cited there)
PUSHBOT = dict(nq=2, nu=2, nw=2, nc=2, nb=4)        # pushbot/model.jl:126-130
HOPPER_2D = dict(nq=4, nu=2, nw=2, nc=1, nb=2)      # hopper_2D/model.jl:100-104
QUADRUPED = dict(nq=11, nu=8, nw=2, nc=4, nb=8)     # quadruped/model.jl:500-504
CENTROIDAL = dict(nq=18, nu=12, nw=3, nc=4, nb=16)  # centroidal_quadruped/model.jl:187-190 (= point_foot_quadruped, centroidal_quadruped_box)
FLAMINGO = dict(nq=9, nu=6, nw=2, nc=4, nb=8)       # flamingo/model.jl
HOPPER_3D = dict(nq=7, nu=3, nw=3, nc=1, nb=4)      # hopper_3D/model.jl:96-99
WALLEDCARTPOLE = dict(nq=4, nu=1, nw=4, nc=2, nb=4)  # walledcartpole/model.jl:143-147
PARTICLE = dict(nq=3, nu=3, nw=3, nc=1, nb=4)       # particle/model.jl:114-117
PARTICLE_2D = dict(nq=2, nu=2, nw=2, nc=1, nb=2)    # particle_2D/model.jl
CENTROIDAL_WALL = dict(nq=18, nu=12, nw=3, nc=8, nb=32)   # centroidal_quadruped_wall (nz = 114, ny = 48: runtime-dimension kernel)


@dataclass
class Traj:
    """ContactTraj (trajectory.jl:1-49) restricted to what the path reads."""
    q: np.ndarray       # (H+2, nq)
    u: np.ndarray       # (H, nu)
    w: np.ndarray       # (H, nw)
    gamma: np.ndarray   # (H, nc)
    b: np.ndarray       # (H, nb)
    theta: np.ndarray   # (H, nth)

    @property
    def H(self):
        return self.u.shape[0]

    def copy(self):
        return Traj(*(a.copy() for a in (self.q, self.u, self.w, self.gamma, self.b, self.theta)))

    def update_theta(self, dims: Dims, t=None):
        """update_theta!, trajectory.jl:67-82 (0-based t; None = all)."""
        ts = range(self.H) if t is None else [t]
        for k in ts:
            if 0 <= k < self.H:
                self.theta[k, dims.iq0] = self.q[k]
                self.theta[k, dims.iq1] = self.q[k + 1]
                self.theta[k, dims.iu1] = self.u[k]
                self.theta[k, dims.iw1] = self.w[k]


def copy_traj(dst: Traj, src: Traj, H):
    """copy_traj!, newton.jl:105-128 (first H steps)."""
    dst.q[:H + 2] = src.q[:H + 2]
    dst.u[:H] = src.u[:H]
    dst.w[:H] = src.w[:H]
    dst.gamma[:H] = src.gamma[:H]
    dst.b[:H] = src.b[:H]
    dst.theta[:H] = src.theta[:H]


@dataclass
class Objective:
    """objective.jl:3-47.  q/u/gamma/b: per-step square matrices (dense allowed).
    velocity objective iff v is not None."""
    q: np.ndarray                      # (H, nq, nq)
    u: np.ndarray                      # (H, nu, nu)
    gamma: np.ndarray = None           # (H, nc, nc)
    b: np.ndarray = None               # (H, nb, nb)
    v: np.ndarray = None               # (H, nq, nq) or None
    v_target: np.ndarray = None        # (H, nq)
    q_target: np.ndarray = None        # (H, nq)

    @classmethod
    def tracking(cls, dims, H, q=None, u=None, gamma=None, b=None):
        """TrackingObjective(model, env, H; q, u, γ, b) (objective.jl:9-22): per-step weight matrices, zeros where not given."""
        z = lambda n, M: np.zeros((H, n, n)) if M is None else np.asarray(M, dtype=np.float64).reshape(H, n, n)
        return cls(q=z(dims.nq, q), u=z(dims.nu, u), gamma=z(dims.nc, gamma), b=z(dims.nb, b))

    def __post_init__(self):
        if self.v is not None:
            H, nq = self.q.shape[0], self.q.shape[1]
            if self.v_target is None:
                self.v_target = np.zeros((H, nq))
            if self.q_target is None:     # objective.jl:36-45
                if np.any(self.v_target != 0.0):
                    qt = np.zeros((H, nq))
                    for t in range(1, H):
                        qt[t] = qt[t - 1] + self.v_target[t - 1]
                    self.q_target = qt
                else:
                    self.q_target = np.zeros((H, nq))
