"""Linearization producer for the reference's closed-form models.

In the reference the per-knot linearization `(r0, rz0, rθ0)` that `cimpc_set_linearization` consumes is produced
by code-generated residual functions (`LinearizedStep`, src/controller/linearized_step.jl:10-31, calls
`s.res.r!/rz!/rθ!`; the generated blobs are not shipped).  This module restates the residual itself from the model
definitions and differentiates it with torch (fp64, CPU, `torch.func` - exact derivatives, no symbolic codegen):

  residual          src/simulation/simulation.jl:133-158   (LinearizedCone)
  dynamics          src/dynamics/model.jl:11-36            (variational midpoint integrator)
  z / θ packing     src/simulation/index.jl:413-415, 437-451; simulation.jl:108-124
  hopper_2D         src/dynamics/hopper_2D/model.jl:31-110
  quadruped         src/dynamics/quadruped/model.jl:75-590  (planar kinematic tree, Lagrangian)
  flamingo          src/dynamics/flamingo/model.jl:62-503   (planar biped with toe / heel contacts)
  centroidal_quad.  src/dynamics/centroidal_quadruped/model.jl:61-229, src/dynamics/euler.jl:3-11
  reference traj.   src/controller/trajectory.jl:152-184    (`get_trajectory`, :split_traj_alt)

It is host-side input preparation (the reference does it once per knot at policy build) - not
part of the timed path.  Environments: flat ground only (`flat_2D_lc`, `flat_3D_lc`: surface rotation = identity).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch
from torch.func import grad, jacfwd, jacrev, jvp, vmap

F64 = torch.float64


def _t(x):
    return torch.as_tensor(np.asarray(x, dtype=np.float64), dtype=F64)


class ContactModel:
    """A model in the reference's sense: dimensions + M, C, B, A, J, ϕ (flat ground, linearized friction cone)."""
    name = ""
    nq = nu = nw = nc = 0
    space = 2                     # dim(env): 2 (x, z) or 3 (x, y, z)
    mu_world = 1.0
    g = 9.81

    @property
    def nf(self):                 # friction_dim(env), environment.jl:126-127
        return 2 if self.space == 2 else 4

    @property
    def nb(self):
        return self.nc * self.nf

    @property
    def nz(self):
        return self.nq + 4 * self.nc + 2 * self.nb

    @property
    def nth(self):
        return 2 * self.nq + self.nu + self.nw + 2

    def friction_mapping(self):   # environment.jl:105-112
        if self.space == 2:
            return _t([[1.0, -1.0]])
        return _t([[1.0, 0.0, -1.0, 0.0], [0.0, 1.0, 0.0, -1.0]])

    # -- to be provided by the model ---------------------------------------------------------------
    def joint_friction(self):
        return torch.zeros(self.nq, dtype=F64)

    def lagrangian_derivatives(self, q, v):       # dynamics/model.jl:11-15: D1L = -C(q, v), D2L = M(q) v
        return -self.C(q, v), self.M(q) @ v

    def kinematics(self, q):                      # stacked contact-point positions, `space` entries per contact
        raise NotImplementedError

    def J(self, q):
        return jacfwd(self.kinematics)(q)

    def phi(self, q):                             # signed distances: the last coordinate of every contact point
        return self.kinematics(q)[self.space - 1::self.space]

    def B(self, q):
        raise NotImplementedError

    def A(self, q):                               # disturbance enters the first nw coordinates (all three models)
        return torch.eye(self.nw, self.nq, dtype=F64)

    # -- shared ------------------------------------------------------------------------------------
    def contact_forces(self, gamma, b):           # per contact [m b_i ; γ_i]  (e.g. quadruped/model.jl:487-495)
        m = self.friction_mapping()
        parts = []
        for i in range(self.nc):
            parts.append(m @ b[i * self.nf:(i + 1) * self.nf])
            parts.append(gamma[i:i + 1])
        return torch.cat(parts)

    def velocity_stack(self, q1, q2, h):          # per contact mᵀ v_tangential  (e.g. quadruped/model.jl:497-510)
        v = self.J(q2) @ (q2 - q1) / h
        m = self.friction_mapping()
        return torch.cat([m.T @ v[i * self.space:i * self.space + self.space - 1] for i in range(self.nc)])

    def E(self):                                  # simulation.jl:126-130
        return torch.kron(torch.eye(self.nc, dtype=F64), torch.ones(1, self.nf, dtype=F64))

    def dynamics(self, h, q0, q1, u1, w1, lam, q2):
        qm1, vm1 = 0.5 * (q0 + q1), (q1 - q0) / h
        qm2, vm2 = 0.5 * (q1 + q2), (q2 - q1) / h
        d1l1, d2l1 = self.lagrangian_derivatives(qm1, vm1)
        d1l2, d2l2 = self.lagrangian_derivatives(qm2, vm2)
        return (0.5 * h * d1l1 + d2l1 + 0.5 * h * d1l2 - d2l2 + self.B(qm2).T @ u1 + self.A(qm2).T @ w1 + lam
                - h * self.joint_friction() * vm2)

    def unpack_z(self, z):
        nq, nc, nb = self.nq, self.nc, self.nb
        o = np.cumsum([0, nq, nc, nb, nc, nc, nb, nc])
        return tuple(z[o[i]:o[i + 1]] for i in range(7))      # q2, γ1, b1, ψ1, s1, η1, s2

    def unpack_theta(self, th):
        nq, nu, nw = self.nq, self.nu, self.nw
        o = np.cumsum([0, nq, nq, nu, nw, 1, 1])
        return tuple(th[o[i]:o[i + 1]] for i in range(6))     # q0, q1, u1, w1, μ, h

    def residual(self, z, th, kappa):
        q0, q1, u1, w1, mu, h = self.unpack_theta(th)
        q2, gam, b, psi, s1, eta, s2 = self.unpack_z(z)
        h = h[0]
        lam = self.J(q2).T @ self.contact_forces(gam, b)
        E = self.E()
        return torch.cat([
            self.dynamics(h, q0, q1, u1, w1, lam, q2),
            s1 - self.phi(q2),
            eta - self.velocity_stack(q1, q2, h) - E.T @ psi,
            s2 - (mu[0] * gam - E @ b),
            gam * s1 - kappa,
            b * eta - kappa,
            psi * s2 - kappa,
        ])

    def pack_z(self, q2, gam, b, psi, eta):       # index.jl:437-441: s1 = ϕ(q2), s2 = μ_world γ - E b
        q2, gam, b, psi, eta = map(_t, (q2, gam, b, psi, eta))
        s1 = self.phi(q2)
        s2 = self.mu_world * gam - self.E() @ b
        return torch.cat([q2, gam, b, psi, s1, eta, s2]).numpy()

    def pack_theta(self, q0, q1, u1, w1, mu, h):
        return np.concatenate([q0, q1, u1, w1, [mu], [h]])

    def linearize(self, z, th, kappa):
        """(r0, rz0, rθ0) at (z, θ, κ): what `LinearizedStep(s, z, θ, κ)` holds (linearized_step.jl:10-31)."""
        zt, tt = _t(z), _t(th)
        k = torch.tensor(float(kappa), dtype=F64)
        r0 = self.residual(zt, tt, k)
        rz0, rth0 = jacrev(self.residual, argnums=(0, 1))(zt, tt, k)
        return r0.numpy(), rz0.numpy(), rth0.numpy()

    def linearize_batch(self, z, th, kappa):
        """`linearize` for a stack of knots (one vmapped pass)."""
        zt, tt = _t(z), _t(th)
        k = torch.tensor(float(kappa), dtype=F64)
        r0 = vmap(self.residual, in_dims=(0, 0, None))(zt, tt, k)
        rz0, rth0 = vmap(jacrev(self.residual, argnums=(0, 1)), in_dims=(0, 0, None))(zt, tt, k)
        return r0.numpy(), rz0.numpy(), rth0.numpy()


class Hopper2D(ContactModel):
    name, nq, nu, nw, nc, space = "hopper_2D", 4, 2, 2, 1, 2
    mu_world = 0.8
    mb, ml, Jb, Jl = 3.0, 0.3, 0.75, 0.075

    def M(self, q):
        return torch.diag(_t([self.mb + self.ml, self.mb + self.ml, self.Jb + self.Jl, self.ml]))

    def C(self, q, v):
        return _t([0.0, (self.mb + self.ml) * self.g, 0.0, 0.0])

    def kinematics(self, q):
        return torch.stack([q[0] + q[3] * torch.sin(q[2]), q[1] - q[3] * torch.cos(q[2])])

    def B(self, q):
        z, o = torch.zeros((), dtype=F64), torch.ones((), dtype=F64)
        return torch.stack([torch.stack([z, z, o, z]), torch.stack([-torch.sin(q[2]), torch.cos(q[2]), z, o])])


class Particle(ContactModel):
    """src/dynamics/particle/model.jl: 3-D unit point mass, its own contact point (flat_3D_lc: four friction directions)."""
    name, nq, nu, nw, nc, space = "particle", 3, 3, 3, 1, 3
    mu_world = 1.0
    m = 1.0

    def M(self, q):
        return self.m * torch.eye(3, dtype=F64)

    def C(self, q, v):
        return _t([0.0, 0.0, self.m * self.g])

    def kinematics(self, q):
        return q

    def B(self, q):
        return torch.eye(3, dtype=F64)


class PlanarChain(ContactModel):
    """Planar articulated model with ABSOLUTE link angles: q = (x, z, angles...).  Every body / contact point is a chain
    of (signed length, angle index) segments from the hip at (x, z): a segment adds r (sin θ, -cos θ).  The Lagrangian
    is the sum over bodies of translational + rotational kinetic energy minus m g z (quadruped/model.jl:261-359,
    flamingo/model.jl:257-325)."""
    space = 2

    def bodies(self):       # (mass, inertia, own angle index, chain to the centre of mass)
        raise NotImplementedError

    def contacts(self):     # chains to the contact points
        raise NotImplementedError

    def torque_pairs(self):  # (parent angle, child angle) per actuator: the torque acts on the relative angle
        raise NotImplementedError

    @staticmethod
    def _point(q, chain):
        x, z = q[0], q[1]
        for r, k in chain:
            x = x + r * torch.sin(q[k])
            z = z - r * torch.cos(q[k])
        return x, z

    def lagrangian(self, q, v):
        L = torch.zeros((), dtype=F64)
        for m, J, own, chain in self.bodies():
            vx, vz = v[0], v[1]
            for r, k in chain:
                vx = vx + r * torch.cos(q[k]) * v[k]
                vz = vz + r * torch.sin(q[k]) * v[k]
            _, pz = self._point(q, chain)
            L = L + 0.5 * m * (vx * vx + vz * vz) + 0.5 * J * v[own] ** 2 - m * self.g * pz
        return L

    def lagrangian_derivatives(self, q, v):
        # C = d/dq(dL/dq̇) q̇ - dL/dq (quadruped/model.jl:479-484); D1L = -C, D2L = dL/dq̇ = M(q) q̇
        dLdv = grad(self.lagrangian, argnums=1)
        d2l, cor = jvp(lambda qq: dLdv(qq, v), (q,), (v,))
        dLdq = grad(self.lagrangian, argnums=0)(q, v)
        return dLdq - cor, d2l

    def kinematics(self, q):
        pts = []
        for chain in self.contacts():
            pts.extend(self._point(q, chain))
        return torch.stack(pts)

    def B(self, q):
        pairs = self.torque_pairs()
        Bm = np.zeros((len(pairs), self.nq))
        for i, (a, b) in enumerate(pairs):
            Bm[i, a], Bm[i, b] = -1.0, 1.0
        return _t(Bm)


class Quadruped(PlanarChain):
    """Planar quadruped (~Unitree A1): q = (x, z, torso, thigh1, calf1, thigh2, calf2, thigh3, calf3, thigh4, calf4)."""
    name, nq, nu, nw, nc = "quadruped", 11, 8, 2, 4
    mu_world = 1.0
    mu_joint = 0.1
    m_torso, m_thigh, m_leg = 4.713 + 4 * 0.696, 1.013, 0.166
    J_torso, J_thigh, J_leg = 0.01683 + 4 * 0.696 * 0.183 ** 2, 0.00552, 0.00299
    l_torso, l_thigh, l_leg = 0.183 * 2, 0.2, 0.2
    d_torso, d_thigh, d_leg = 0.5 * 0.183 * 2 + 0.0127, 0.5 * 0.2 - 0.00323, 0.5 * 0.2 - 0.006435

    def bodies(self):
        lt, lh = self.l_torso, self.l_thigh
        out = [(self.m_torso, self.J_torso, 2, [(self.d_torso, 2)])]
        for thigh, calf, front in ((3, 4, False), (5, 6, False), (7, 8, True), (9, 10, True)):
            root = [(lt, 2)] if front else []         # legs 3, 4 hang from the far end of the torso
            out.append((self.m_thigh, self.J_thigh, thigh, root + [(self.d_thigh, thigh)]))
            out.append((self.m_leg, self.J_leg, calf, root + [(lh, thigh), (self.d_leg, calf)]))
        return out

    def contacts(self):
        lt, lh, ll = self.l_torso, self.l_thigh, self.l_leg
        return [[(lh, 3), (ll, 4)], [(lh, 5), (ll, 6)], [(lt, 2), (lh, 7), (ll, 8)], [(lt, 2), (lh, 9), (ll, 10)]]

    def torque_pairs(self):
        return ((2, 3), (3, 4), (2, 5), (5, 6), (2, 7), (7, 8), (2, 9), (9, 10))

    def joint_friction(self):
        return _t([0.0] * 3 + [self.mu_joint] * 8)


class Flamingo(PlanarChain):
    """Planar biped with feet (src/dynamics/flamingo/model.jl): q = (x, z, torso, thigh1, calf1, thigh2, calf2, foot1,
    foot2); the torso points UP from the hip (negative segment), each foot has a toe and a heel contact."""
    name, nq, nu, nw, nc = "flamingo", 9, 6, 2, 4
    mu_world = 0.9
    m_torso, m_thigh, m_calf, m_foot = 12.0, 0.4598, 0.306, 0.3466
    J_torso, J_thigh, J_calf, J_foot = 0.10, 0.01256, 0.00952, 0.0015
    l_torso, l_thigh, l_calf, l_foot = 0.385, 0.42, 0.45, 0.1725
    d_torso, d_thigh, d_calf, d_foot = 0.20, 0.42 / 2, 0.45 / 2, 0.0525

    def bodies(self):
        cb = 0.5 * (self.l_foot - self.d_foot)
        out = [(self.m_torso, self.J_torso, 2, [(-self.d_torso, 2)])]
        for thigh, calf, foot in ((3, 4, 7), (5, 6, 8)):
            out.append((self.m_thigh, self.J_thigh, thigh, [(self.d_thigh, thigh)]))
            out.append((self.m_calf, self.J_calf, calf, [(self.l_thigh, thigh), (self.d_calf, calf)]))
            out.append((self.m_foot, self.J_foot, foot, [(self.l_thigh, thigh), (self.l_calf, calf), (cb, foot)]))
        return out

    def contacts(self):     # toe 1, heel 1, toe 2, heel 2 (flamingo/model.jl:343-350)
        out = []
        for thigh, calf, foot in ((3, 4, 7), (5, 6, 8)):
            leg = [(self.l_thigh, thigh), (self.l_calf, calf)]
            out += [leg + [(self.l_foot, foot)], leg + [(-self.d_foot, foot)]]
        return out

    def torque_pairs(self):
        return ((2, 3), (3, 4), (2, 5), (5, 6), (4, 7), (6, 8))


def _skew(x):
    z = torch.zeros((), dtype=F64)
    return torch.stack([torch.stack([z, -x[2], x[1]]), torch.stack([x[2], z, -x[0]]), torch.stack([-x[1], x[0], z])])


def euler_rotation_matrix(e):                         # dynamics/euler.jl:3-11
    a, b, c = e[0], e[1], e[2]
    ca, sa, cb, sb, cc, sc = torch.cos(a), torch.sin(a), torch.cos(b), torch.sin(b), torch.cos(c), torch.sin(c)
    return torch.stack([torch.stack([ca * cb, ca * sb * sc - sa * cc, ca * sb * cc + sa * sc]),
                        torch.stack([sa * cb, sa * sb * sc + ca * cc, sa * sb * cc - ca * sc]),
                        torch.stack([-sb, cb * sc, cb * cc])])


class CentroidalQuadruped(ContactModel):
    """q = (body position, body orientation, foot 1..4 positions); point feet, single rigid body."""
    name, nq, nu, nw, nc, space = "centroidal_quadruped", 18, 12, 3, 4, 3
    mu_world = 0.3
    mu_joint = 1.0
    mass_body, mass_foot = 13.5, 0.2
    inertia = (0.0178533 * 10.0, 0.0377999 * 10.0, 0.0456542 * 10.0)

    def M(self, q):
        return torch.diag(_t([self.mass_body] * 3 + list(self.inertia) + [self.mass_foot] * 12))

    def C(self, q, v):
        w = v[3:6]
        grav = lambda m: _t([0.0, 0.0, m * self.g])
        return torch.cat([grav(self.mass_body), _skew(w) @ (_t(self.inertia) * w)] + [grav(self.mass_foot)] * 4)

    def kinematics(self, q):
        return q[6:18]

    def B(self, q):
        R = euler_rotation_matrix(q[3:6])
        I3, Z3 = torch.eye(3, dtype=F64), torch.zeros(3, 3, dtype=F64)
        rows = [torch.cat([I3] * 4, dim=1),
                torch.cat([R.T @ _skew(q[6 + 3 * i:9 + 3 * i] - q[0:3]) for i in range(4)], dim=1)]
        for i in range(4):
            rows.append(torch.cat([-I3 if j == i else Z3 for j in range(4)], dim=1))
        return torch.cat(rows, dim=0).T                # (nu, nq)

    def joint_friction(self):
        return self.mu_joint * _t([10.0] * 3 + [30.0] * 3 + [10.0] * 12)


class CentroidalQuadrupedUndamped(CentroidalQuadruped):
    """`centroidal_quadruped_undamped` (centroidal_quadruped/model.jl:218-228): the variant the shipped in-place
    trot references satisfy to round-off (tests/test_real_models.py)."""
    name = "centroidal_quadruped_undamped"

    def joint_friction(self):
        return torch.zeros(self.nq, dtype=F64)


MODELS = {"hopper_2D": Hopper2D, "particle": Particle, "quadruped": Quadruped, "flamingo": Flamingo, "centroidal_quadruped": CentroidalQuadruped,
          "centroidal_quadruped_undamped": CentroidalQuadrupedUndamped}


@dataclass
class ReferenceProblem:
    """A reference trajectory (ContactTraj) with its per-knot linearization table (ImplicitTrajectory.lin)."""
    model: ContactModel
    H: int
    h: float
    kappa: float
    q: np.ndarray          # (H + 2, nq)
    u: np.ndarray
    w: np.ndarray
    gamma: np.ndarray
    b: np.ndarray
    z: np.ndarray          # (H, nz)
    theta: np.ndarray      # (H, nθ)
    r0: np.ndarray         # (H, nz)
    rz0: np.ndarray        # (H, nz, nz)
    rth0: np.ndarray       # (H, nz, nθ)


def reference_problem(model: ContactModel, gait, kappa: float, update_friction: bool = False) -> ReferenceProblem:
    """`get_trajectory(...; load_type = :split_traj_alt)` (trajectory.jl:169-180) followed by the `LinearizedStep`
    of every knot at κ (`ImplicitTrajectory`, implicit_dynamics.jl:37-50).  `update_friction`:
    `update_friction_coefficient!` (trajectory.jl:133-141) - θ carries the model's μ_world instead of the file's μ."""
    H = gait.H
    if update_friction:
        import dataclasses
        gait = dataclasses.replace(gait, mu=float(model.mu_world))
    w = np.zeros((H, model.nw))
    z = np.stack([model.pack_z(gait.q[t + 2], gait.gamma[t], gait.b[t], gait.psi[t], gait.eta[t]) for t in range(H)])
    th = np.stack([model.pack_theta(gait.q[t], gait.q[t + 1], gait.u[t], w[t], gait.mu, gait.h) for t in range(H)])
    r0, rz0, rth0 = model.linearize_batch(z, th, kappa)
    return ReferenceProblem(model, H, gait.h, kappa, gait.q.copy(), gait.u.copy(), w, gait.gamma.copy(), gait.b.copy(), z, th,
                            r0, rz0, rth0)


def reference_problem_from_traj(model: ContactModel, traj, kappa: float) -> ReferenceProblem:
    """The same from a serialized ContactTraj (`load_type = :joint_traj`: z and θ come from the file as they are)."""
    r0, rz0, rth0 = model.linearize_batch(traj.z, traj.theta, kappa)
    return ReferenceProblem(model, traj.H, traj.h, kappa, traj.q.copy(), traj.u.copy(), traj.w.copy(), traj.gamma.copy(),
                            traj.b.copy(), traj.z.copy(), traj.theta.copy(), r0, rz0, rth0)


def get_stride(model: ContactModel, q_ref: np.ndarray) -> np.ndarray:
    """`get_stride` (mpc_utils.jl:103-107): forward progress of one gait period, first coordinate only."""
    stride = np.zeros(model.nq)
    stride[0] = q_ref[-2][0] - q_ref[0][0]
    return stride


def make_rollout(P: ReferenceProblem, H: int, phase: int, seed: int = 0, perturb: float = 0.0, vel_perturb: float = 0.05):
    """One MPC problem on the reference gait: window = knots phase .. phase+H+1 (0-based, periodic), the reference
    restricted to it with the gait's stride added on wrap-around (`rot_n_stride!`, mpc_utils.jl:48-101) and θ
    rebuilt from the shifted configurations (`update_θ!`, trajectory.jl:67-82); (q0, q1) = reference + one shared
    U(-perturb, perturb) offset (examples/quadruped/monte_carlo.jl:79-91 perturbs the initial configuration)."""
    m, Hr = P.model, P.H
    stride = get_stride(m, P.q)
    window = (phase + np.arange(H + 2)) % Hr
    # what `phase` applications of rot_n_stride! leave in p.traj.q: mpc_stride! re-derives q_{H+1}, q_{H+2} from
    # q_1, q_2 (+ stride) after every rotation, the file's own q_{H+1} survives one lap (cimpc_set_gait does the same)
    def qa(a):
        if a == 0 or (phase == 0 and a == Hr + 1):
            return P.q[a]
        lap, r = divmod(a - 1, Hr)
        return P.q[r + 1] + lap * stride
    q = np.stack([qa(phase + i) for i in range(H + 2)])
    kn = window[:H]
    th = P.theta[kn].copy()
    th[:, :m.nq] = q[:H]
    th[:, m.nq:2 * m.nq] = q[1:H + 1]
    rng = np.random.default_rng(seed)
    dq = rng.uniform(-perturb, perturb, m.nq)
    q0 = q[0] + dq
    q1 = q[1] + dq + vel_perturb * rng.uniform(-perturb, perturb, m.nq)
    return dict(window=window, q=q, u=P.u[kn].copy(), w=P.w[kn].copy(), gamma=P.gamma[kn].copy(), b=P.b[kn].copy(),
                theta=th, q0=q0, q1=q1)


def relative_state_cost(qbody, qorientation, qfoot):
    """Dense 18 x 18 state cost of the centroidal examples (centroidal_quadruped/model.jl:170-185): body, orientation
    and foot-relative-to-body terms."""
    Q = np.zeros((18, 18))
    Q[0:3, 0:3] = np.diag(qbody)
    Q[3:6, 3:6] = np.diag(qorientation)
    F = np.diag(qfoot)
    for i in range(4):
        s = slice(6 + 3 * i, 9 + 3 * i)
        Q[0:3, 0:3] += F
        Q[s, s] += F
        Q[0:3, s] -= F
        Q[s, 0:3] -= F
    return Q
