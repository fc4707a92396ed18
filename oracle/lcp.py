"""Linearized contact-LCP operators: CPU restatement (numpy, fp64).

Test infrastructure only (see oracle/__init__.py).

Follows, function for function:
  src/solver/qr.jl:18-22, 113-158      modified Gram-Schmidt QR + back-subst.
  src/solver/schur.jl:33-49, 80-110    Schur complement about the top-left block
  src/controller/linearized_solver.jl  RLin/RZLin/RthLin ctors :67-161,224-304,325-359
                                       rlin! :364-373, rzlin! :378-399,
                                       violations :401-409, correction :411-418,
                                       linear_solve! (vector) :424-444,
                                       linear_solve! (matrix / sensitivities) :451-479
"""
import numpy as np

from .dims import Dims


# ----------------------------------------------------------------------------
# qr.jl
# ----------------------------------------------------------------------------
def triu_perm(k, j):
    """Packed index of R[k, j] (1-based k<=j), qr.jl:18-22.  Returns 0-based."""
    return (j - 1) * j // 2 + k - 1


def mgs_factorize(A):
    """Modified Gram-Schmidt, left-looking, exactly the loop of qr.jl:113-137.

    Returns (qs, rs): qs[j] = j-th orthonormal column, rs = packed upper R
    (column-major packed, qr.jl:18-22)."""
    n = A.shape[0]
    qs = [None] * n
    rs = np.zeros(n * (n + 1) // 2)
    off = 0
    for j in range(n):
        q = A[:, j].copy()
        for k in range(j):
            rs[off] = float(np.dot(q, qs[k]))
            q = q - qs[k] * rs[off]
            off += 1
        rs[off] = float(np.sqrt(np.dot(q, q)))
        q = q / rs[off]
        qs[j] = q
        off += 1
    return qs, rs


def qr_solve(qs, rs, b):
    """qr.jl:142-158: x = R^-1 Q^T b with row-oriented back-substitution."""
    n = len(qs)
    x = np.zeros(n)
    for j in range(n):
        x[j] = float(np.dot(qs[j], b))
    for j in range(n - 1, -1, -1):
        for k in range(j + 1, n):
            x[j] -= rs[triu_perm(j + 1, k + 1)] * x[k]
        x[j] /= rs[triu_perm(j + 1, j + 1)]
    return x


def qr_matrix_solve(qs, rs, B):
    """qr.jl:46-60 (matrix right-hand side)."""
    n, m = B.shape
    X = np.zeros((n, m))
    for j in range(n):
        X[j, :] = qs[j] @ B
    for j in range(n - 1, -1, -1):
        for k in range(j + 1, n):
            X[j, :] -= rs[triu_perm(j + 1, k + 1)] * X[k, :]
        X[j, :] /= rs[triu_perm(j + 1, j + 1)]
    return X


# ----------------------------------------------------------------------------
# schur.jl
# ----------------------------------------------------------------------------
class Schur:
    """M = [A B; C D], complement about A.  schur.jl:13-49."""

    def __init__(self, A, B, C, D):
        self.A, self.B, self.C = A.copy(), B.copy(), C.copy()
        self.Ai = np.linalg.inv(A)              # schur.jl:39
        self.CAi = C @ self.Ai                  # schur.jl:40
        self.CAiB = C @ self.Ai @ B             # schur.jl:41
        self.qs, self.rs = mgs_factorize(D - self.CAiB)   # schur.jl:43

    @classmethod
    def from_matrix(cls, M, n):
        return cls(M[:n, :n], M[:n, n:], M[n:, :n], M[n:, n:])

    def factorize(self, D):
        """schur_factorize!, schur.jl:80-88."""
        self.qs, self.rs = mgs_factorize(D - self.CAiB)

    def solve(self, u, v):
        """schur_solve!, schur.jl:93-110.  Returns (x, y)."""
        temp = qr_solve(self.qs, self.rs, self.CAi @ u - v)
        x = self.Ai @ (u + self.B @ temp)
        y = -temp
        return x, y


# ----------------------------------------------------------------------------
# linearized_solver.jl : per-knot linearization table (RLin + RZLin + RthLin)
# ----------------------------------------------------------------------------
class LinTable:
    """Constant blocks of one reference knot (A1 of SURVEY section 8a).

    Built from the dense (z0, th0, r0, rz0, rth0) of a LinearizedStep
    (linearized_step.jl:1-31) exactly as the RLin / RZLin / RthLin constructors
    slice them (linearized_solver.jl:67-161, 224-304, 325-359)."""

    def __init__(self, dims: Dims, z0, th0, r0, rz0, rth0):
        d = dims
        self.dims = d
        ix, iy1, iy2 = d.ix, d.iy1, d.iy2
        idyn, irst, ibil = d.ix, d.iy1, d.iy2
        self.rdyn0 = r0[idyn].copy()
        self.rrst0 = r0[irst].copy()
        self.rbil0 = r0[ibil].copy()
        self.Dx = rz0[np.ix_(idyn, ix)].copy()
        self.Dy1 = rz0[np.ix_(idyn, iy1)].copy()
        self.Rx = rz0[np.ix_(irst, ix)].copy()
        self.Ry1 = rz0[np.ix_(irst, iy1)].copy()
        self.Ry2 = np.diag(rz0[np.ix_(irst, iy2)]).copy()
        self.rthdyn = rth0[idyn, :].copy()
        self.rthrst = rth0[irst, :].copy()
        self.rthbil = rth0[ibil, :].copy()
        self.x0 = z0[ix].copy()
        self.y10 = z0[iy1].copy()
        self.y20 = z0[iy2].copy()
        self.th0 = th0.copy()
        self.alt = np.zeros(d.nc)        # RLin.alt, linearized_solver.jl:64-65
        # RZLin ctor: y1 = diag(rz0[ibil,iy2]), y2 = diag(rz0[ibil,iy1])  (:247-248)
        y1 = np.diag(rz0[np.ix_(ibil, iy2)]).copy()
        y2 = np.diag(rz0[np.ix_(ibil, iy1)]).copy()
        with np.errstate(divide="ignore", invalid="ignore"):
            D = self.Ry1 - np.diag(self.Ry2 * y2 / y1)            # :251
        if not np.all(np.isfinite(D)):
            D = self.Ry1.copy()
        self.S = Schur(self.Dx, self.Dy1, self.Rx, D)             # :252-254
        # live state of RZLin
        self.y1 = y1
        self.y2 = y2


def rlin(tab: LinTable, z, th, kappa):
    """rlin!, linearized_solver.jl:364-373.  Returns (rdyn, rrst, rbil)."""
    d = tab.dims
    x, y1, y2 = z[d.ix], z[d.iy1], z[d.iy2]
    rdyn = tab.rdyn0 + tab.Dx @ (x - tab.x0) + tab.Dy1 @ (y1 - tab.y10) \
        + tab.rthdyn @ (th - tab.th0)
    alt = np.concatenate([tab.alt, np.zeros(d.nc + d.nb)])
    rrst = tab.rrst0 + tab.Rx @ (x - tab.x0) + tab.Ry1 @ (y1 - tab.y10) \
        + tab.Ry2 * (y2 - tab.y20) + tab.rthrst @ (th - tab.th0) + alt
    rbil = y1 * y2 - kappa
    return rdyn, rrst, rbil


def rzlin(tab: LinTable, z, reg=0.0):
    """rzlin!, linearized_solver.jl:378-399 (+ schur_factorize!)."""
    d = tab.dims
    tab.y1 = z[d.iy1].copy()
    tab.y2 = z[d.iy2].copy()
    y1_reg = np.maximum(tab.y1, reg)
    y2_reg = np.maximum(tab.y2, reg)
    D = tab.Ry1 - np.diag(tab.Ry2 * y2_reg / y1_reg)
    tab.S.factorize(D)


def residual_violation(rdyn, rrst):
    """linearized_solver.jl:401-405."""
    return max(np.max(np.abs(rdyn)), np.max(np.abs(rrst)))


def bilinear_violation(rbil):
    """linearized_solver.jl:407-409."""
    return np.max(np.abs(rbil))


def linear_solve_vec(tab: LinTable, rdyn, rrst, rbil, reg=0.0):
    """linear_solve!(D, rz, r), linearized_solver.jl:424-444.  Returns D[nz]."""
    d = tab.dims
    y1_reg = np.maximum(reg, tab.y1)
    y2_reg = np.maximum(reg, tab.y2)
    u = rdyn
    v = rrst - tab.Ry2 * rbil / y1_reg
    x, y = tab.S.solve(u, v)
    Delta = np.zeros(d.nz)
    Delta[d.ix] = x
    Delta[d.iy1] = y
    Delta[d.iy2] = (rbil - y2_reg * Delta[d.iy1]) / y1_reg
    return Delta


def linear_solve_mat(tab: LinTable, reg=0.0):
    """linear_solve!(dz, rz, rth), linearized_solver.jl:451-479.

    Returns +rz^-1 rth (nz x nth) - sign as stored by the reference callback
    (the reference test linearized_solver.jl:67 compares with rz0 \\ rth0);
    the minus of dz/dth = -rz^-1 rth is applied by the IP driver (ip.py)."""
    d = tab.dims
    y1_reg = np.maximum(reg, tab.y1)
    y2_reg = np.maximum(reg, tab.y2)
    dz = np.zeros((d.nz, d.nth))
    for i in range(d.nth):
        u = tab.rthdyn[:, i]
        v = tab.rthrst[:, i]
        x, y = tab.S.solve(u, v)
        dz[d.ix, i] = x
        dz[d.iy1, i] = y
        dz[d.iy2, i] = (tab.rthbil[:, i] - y2_reg * dz[d.iy1, i]) / y1_reg
    return dz


def linear_solve_mat_x_adjoint(tab: LinTable, reg=0.0, ncols=None):
    """The x rows of linear_solve!(dz, rz, rth) (linearized_solver.jl:451-479) in the ADJOINT form the device's sensitivity
    pass runs in :configuration mode (ip_kernel_impl.h: sensitivities, Model::ADJ) - a model of that schedule, not of a
    reference routine.  Column c of the x rows is, by schur_solve! (schur.jl:93-110),

        x_c = Ai (u_c + B M^-1 (CAi u_c - v_c)) = K0[:, c] + A2 Gs[:, c],   K0 = Ai rthdyn, Gs = CAi rthdyn - rthrst,

    with A2 = (Ai B) M^-1 formed row by row from nx solves with M^T (its own MGS factorization) instead of one solve with M
    per column.  Returns +x rows (nx x ncols), the sign of linear_solve_mat."""
    d = tab.dims
    n = d.nth if ncols is None else ncols
    y1_reg = np.maximum(reg, tab.y1)
    y2_reg = np.maximum(reg, tab.y2)
    S = tab.S
    M = (tab.Ry1 - np.diag(tab.Ry2 * y2_reg / y1_reg)) - S.CAiB
    qs, rs = mgs_factorize(M.T.copy())
    AiB = S.Ai @ S.B
    A2 = np.stack([qr_solve(qs, rs, AiB[i]) for i in range(AiB.shape[0])])
    K0 = S.Ai @ tab.rthdyn[:, :n]
    Gs = S.CAi @ tab.rthdyn[:, :n] - tab.rthrst[:, :n]
    return K0 + A2 @ Gs


def dense_rz(tab: LinTable, z):
    """Dense rz at z with the structure of linearized_solver.jl:167-169."""
    d = tab.dims
    nx, ny = d.nx, d.ny
    M = np.zeros((d.nz, d.nz))
    M[:nx, :nx] = tab.Dx
    M[:nx, nx:nx + ny] = tab.Dy1
    M[nx:nx + ny, :nx] = tab.Rx
    M[nx:nx + ny, nx:nx + ny] = tab.Ry1
    M[nx:nx + ny, nx + ny:] = np.diag(tab.Ry2)
    M[nx + ny:, nx:nx + ny] = np.diag(z[d.iy2])
    M[nx + ny:, nx + ny:] = np.diag(z[d.iy1])
    return M
