"""Seeded synthetic reference-shaped problems: the input generator of bench.py and of the parity tests.

The reference's code-generated residual blobs are missing from the mount
(.MISSING_LARGE_BLOBS) and Julia is absent, so real `rz0/rth0` cannot be
produced here.  This generator builds per-knot linearizations with the block
structure of the true `residual` (src/simulation/simulation.jl:133-158) and of
the variational dynamics (src/dynamics/model.jl:18-36):

  rdyn = M(q1-q0)/h - M(q2-q1)/h + B^T u + A^T w + Jn^T g + Jt^T (b+ - b-)      [+ O(h) stiffness]
  rimp = s1 - phi(q2)             rmdp = eta - vT(q1,q2,h) - E^T psi
  rfri = s2 - (mu g - E b)        rbil = y1 .* y2 - kappa

so Dx ~ -M/h, Dy1 = [Jn^T, Jt^T P, 0], Rx = [-Jn; -P^T Jt/h; 0], Ry1 = the fixed
(-E^T, -mu, E) pattern, Ry2 = 1, rth blocks from the same terms.  z0 sits on the
central path (y10.*y20 = kappa) with r0 = 0, i.e. the reference trajectory is an
exact solution of the linearized model.
"""
import numpy as np

from .trajectory import Dims, Objective, Traj


def make_problem(dims: Dims, H_ref: int, seed: int = 0, h: float = 0.015625, mu: float = 0.5,
                 kappa: float = 2.0e-4):
    """Returns dict(z0 (H_ref,nz), th0 (H_ref,nth), r0 (H_ref,nz), rz0 (H_ref,nz,nz),
    rth0 (H_ref,nz,nth), q_ref (H_ref+2,nq), u_ref, w_ref, gamma_ref, b_ref, h, mu, kappa)."""
    d = dims
    rng = np.random.default_rng(seed)
    nq, nu, nw, nc, nb = d.nq, d.nu, d.nw, d.nc, d.nb
    nx, ny, nz, nth = d.nx, d.ny, d.nz, d.nth
    fd = nb // nc                      # friction_dim
    nt = fd // 2                       # tangential directions per contact
    # mass-like SPD matrix
    G = rng.standard_normal((nq, nq)) * 0.15
    M = np.diag(rng.uniform(0.5, 3.0, nq)) + G @ G.T
    Kst = rng.standard_normal((nq, nq)) * 0.3
    Kst = Kst @ Kst.T
    Bact = np.zeros((nu, nq))
    for k in range(nu):
        Bact[k, nq - nu + k if nq >= nu else k % nq] = 1.0
    Bact += 0.05 * rng.standard_normal((nu, nq))
    Aw = np.zeros((nw, nq))
    for k in range(min(nw, nq)):
        Aw[k, k] = 1.0
    E = np.kron(np.eye(nc), np.ones((1, fd)))            # E_func, simulation.jl:127-131
    P = np.vstack([np.eye(nt), -np.eye(nt)])             # (fd, nt): b = (b+, b-)
    # smooth periodic reference gait
    tt = np.arange(H_ref + 2)
    amp = rng.uniform(0.02, 0.1, nq)
    ph = rng.uniform(0, 2 * np.pi, nq)
    q_ref = amp[None, :] * np.sin(2 * np.pi * tt[:, None] / H_ref + ph[None, :]) \
        + rng.uniform(-0.3, 0.3, nq)[None, :]
    u_ref = 0.5 * np.sin(2 * np.pi * tt[:H_ref, None] / H_ref + rng.uniform(0, 6.28, nu)[None, :])
    w_ref = np.zeros((H_ref, nw))
    out = dict(z0=np.zeros((H_ref, nz)), th0=np.zeros((H_ref, nth)), r0=np.zeros((H_ref, nz)),
               rz0=np.zeros((H_ref, nz, nz)), rth0=np.zeros((H_ref, nz, nth)),
               q_ref=q_ref, u_ref=u_ref, w_ref=w_ref,
               gamma_ref=np.zeros((H_ref, nc)), b_ref=np.zeros((H_ref, nb)),
               h=h, mu=mu, kappa=kappa)
    # contact Jacobians drift slowly along the gait; like a legged robot each contact
    # point depends on the floating base (first 3 coordinates) and on its own leg joints
    # Models with more contact rows than coordinates (centroidal_quadruped_wall: 8 contacts = 4 feet x {floor, wall}) pair the
    # contacts up per foot: the two contacts of a foot share its leg columns and are in stance on opposite halves of the gait,
    # so that the active set never over-constrains q (all contacts of such a model active at once has no solution).
    paired = nc * (1 + nt) > nq
    nfeet = (nc + 1) // 2 if paired else max(nc, 1)
    def _sparsify(J, rows_per_contact):
        mask = np.zeros_like(J)
        nleg = max((nq - 3) // nfeet, 1)
        for c in range(nc):
            f = c % nfeet
            cols = list(range(min(3, nq))) + [min(3 + f * nleg + k, nq - 1) for k in range(nleg)]
            mask[c * rows_per_contact:(c + 1) * rows_per_contact, cols] = 1.0
        return J * mask
    Jn0 = _sparsify(rng.standard_normal((nc, nq)) * 0.6, 1)
    Jt0 = _sparsify(rng.standard_normal((nc * nt, nq)) * 0.6, nt)
    Jn1 = _sparsify(rng.standard_normal((nc, nq)) * 0.2, 1)
    Jt1 = _sparsify(rng.standard_normal((nc * nt, nq)) * 0.2, nt)
    # stance pattern: each contact is active on a contiguous half of the gait
    stance_phase = rng.uniform(0, 1, nc)
    if paired:
        stance_phase = np.array([stance_phase[c % nfeet] + 0.5 * (c // nfeet) for c in range(nc)])
    for t in range(H_ref):
        s = np.sin(2 * np.pi * t / H_ref)
        Jn = Jn0 + s * Jn1
        Jt = Jt0 + s * Jt1
        JtP = np.zeros((nq, nb))                       # d rdyn / d b
        vst = np.zeros((nb, nq))                       # tangential velocity stack rows
        for c in range(nc):
            Jc = Jt[c * nt:(c + 1) * nt]               # (nt, nq)
            JtP[:, c * fd:(c + 1) * fd] = Jc.T @ P.T
            vst[c * fd:(c + 1) * fd] = P @ Jc
        Dx = -(M / h + 0.5 * h * Kst) + 0.01 * rng.standard_normal((nq, nq))
        Dy1 = np.hstack([Jn.T, JtP, np.zeros((nq, nc))])
        Rx = np.vstack([-Jn, -vst / h, np.zeros((nc, nq))])
        Ry1 = np.zeros((ny, ny))
        # rows: imp (nc) | mdp (nb) | fri (nc);  cols: gamma (nc) | b (nb) | psi (nc)
        Ry1[nc:nc + nb, nc + nb:] = -E.T
        Ry1[nc + nb:, :nc] = -mu * np.eye(nc)
        Ry1[nc + nb:, nc:nc + nb] = E
        Ry2 = np.ones(ny)
        # central-path reference point, consistent with the friction-cone row
        # s2 = mu*gamma - E*b (simulation.jl:154): y10 .* y20 = kappa on every pair
        y10 = np.zeros(ny)
        y20 = np.zeros(ny)
        for c in range(nc):
            active = ((t / H_ref + stance_phase[c]) % 1.0) < 0.5
            ib = [nc + c * fd + k for k in range(fd)]
            if active:
                g = rng.uniform(0.3, 1.5)                  # normal force
                bb = rng.uniform(0.02, 0.9, fd)
                bb *= rng.uniform(0.2, 0.9) * mu * g / bb.sum()   # inside the cone
            else:
                s1 = rng.uniform(0.05, 0.3)                # contact height
                g = kappa / s1
                bb = rng.uniform(0.5, 1.0, fd)
                bb *= rng.uniform(0.3, 0.7) * mu * g / bb.sum()
            s2 = mu * g - bb.sum()
            y10[c], y20[c] = g, kappa / g                  # gamma . s1
            y10[ib], y20[ib] = bb, kappa / bb              # b . eta
            y10[nc + nb + c], y20[nc + nb + c] = kappa / s2, s2   # psi . s2
        z0 = np.concatenate([q_ref[t + 2], y10, y20])
        th0 = np.concatenate([q_ref[t], q_ref[t + 1], u_ref[t], w_ref[t], [mu], [h]])
        rz = np.zeros((nz, nz))
        rz[:nx, :nx] = Dx
        rz[:nx, nx:nx + ny] = Dy1
        rz[nx:nx + ny, :nx] = Rx
        rz[nx:nx + ny, nx:nx + ny] = Ry1
        rz[nx:nx + ny, nx + ny:] = np.diag(Ry2)
        rz[nx + ny:, nx:nx + ny] = np.diag(y20)
        rz[nx + ny:, nx + ny:] = np.diag(y10)
        rth = np.zeros((nz, nth))
        rth[:nx, d.iq0] = -M / h + 0.25 * h * Kst
        rth[:nx, d.iq1] = 2 * M / h - 0.25 * h * Kst
        rth[:nx, d.iu1] = Bact.T
        rth[:nx, d.iw1] = Aw.T
        rth[:nx, -1] = 0.1 * rng.standard_normal(nq)                  # d/dh
        rth[nx + nc:nx + nc + nb, d.iq1] = vst / h
        rth[nx + nc:nx + nc + nb, -1] = 0.1 * rng.standard_normal(nb)  # d/dh
        rth[nx + nc + nb:nx + ny, -2] = -y10[:nc]                      # d/dmu of s2-(mu*g-Eb)
        out["z0"][t], out["th0"][t] = z0, th0
        out["rz0"][t], out["rth0"][t] = rz, rth
        out["gamma_ref"][t] = y10[:nc]
        out["b_ref"][t] = y10[nc:nc + nb]
    return out


def make_objective(dims: Dims, H: int, kind: str = "quadruped", dense_q: bool = False,
                   velocity: bool = False, seed: int = 0):
    """Tracking objective weights.  quadruped: test/controller/mpc_quadruped.jl:23-27;
    hopper: examples/hopper/flat.jl:30-34."""
    d = dims
    if kind == "hopper" and d.nq == 4:
        qd = 0.1 * np.array([0.1, 3.0, 1.0, 3.0])
        ud = np.array([1e-3, 1.0])
    else:
        qd = 1e-2 * np.concatenate([[1.0, 0.02, 0.25], 0.25 * np.ones(max(d.nq - 3, 0))])[:d.nq]
        ud = 3e-2 * np.ones(d.nu)
    Q = np.tile(np.diag(qd)[None], (H, 1, 1))
    if dense_q:
        rng = np.random.default_rng(seed + 17)
        G = rng.standard_normal((d.nq, d.nq)) * 0.02
        Q = Q + (G @ G.T)[None]
    R = np.tile(np.diag(ud)[None], (H, 1, 1))
    Cg = np.tile((1e-100 * np.eye(d.nc))[None], (H, 1, 1))
    Cb = np.tile((1e-100 * np.eye(d.nb))[None], (H, 1, 1))
    V = np.tile((1e-5 * np.eye(d.nq))[None], (H, 1, 1)) if velocity else None
    return Objective(q=Q, u=R, gamma=Cg, b=Cb, v=V)


def make_rollout(dims: Dims, prob, H: int, phase: int, seed: int, perturb: float = 1e-2,
                 vel_perturb: float = 0.05):
    """One Monte-Carlo rollout (examples/quadruped/monte_carlo.jl:76-92 analogue):
    window = gait knots phase..phase+H+1 (mod H_ref), ref = gait restricted to the
    window, (q0,q1) = reference + shared U(-perturb, perturb) offset."""
    d = dims
    H_ref = prob["u_ref"].shape[0]
    rng = np.random.default_rng(seed)
    window = (phase + np.arange(H + 2)) % H_ref
    q = np.stack([prob["q_ref"][(phase + i) % H_ref] for i in range(H + 2)])
    kn = window[:H]
    theta = prob["th0"][kn].copy()
    ref = Traj(q=q, u=prob["u_ref"][kn].copy(), w=prob["w_ref"][kn].copy(),
               gamma=prob["gamma_ref"][kn].copy(), b=prob["b_ref"][kn].copy(), theta=theta)
    ref.update_theta(d)
    # configuration offset shared by (q0, q1) (the reference Monte-Carlo perturbs the initial
    # configuration and keeps the reference velocity) + a small velocity perturbation
    dq = rng.uniform(-perturb, perturb, d.nq)
    q0 = q[0] + dq
    q1 = q[1] + dq + vel_perturb * rng.uniform(-perturb, perturb, d.nq)
    return window, ref, q0, q1
