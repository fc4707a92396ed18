"""CPU restatement of the banded KKT backend (TEST INFRASTRUCTURE ONLY - see oracle/__init__.py).

`kkt_banded_kernel` (contactimplicitmpc/jl_amd/csrc/kkt_dense.hip) factors the KKT matrix of `jacobian!`
(src/controller/newton_jacobian.jl:148-248) in the INTERLEAVED ordering [u_i, q_{i+2}, nu_i] per step
as L D L^T without pivoting, with a sliding (w + R) x (w + R) window in circular slots and R pivots per window update.
This module states the same algorithm in numpy - same slot logic, same order of operations per entry - so that the
structure (bandwidth, quasi-definiteness, blocked update) is checked on the CPU; the device kernel is compared with
numpy's dense solve in tests/test_gpu_parity.py."""
import numpy as np


def interleave_perm(lay, d):
    """reference layout (primal segment step-major, then the duals) -> interleaved [u_i, q_{i+2}, nu_i] per step"""
    H, nr, nd = lay.H, d.nr, d.nd
    s = nr + nd
    p = np.zeros(lay.N, dtype=int)
    for t in range(H):
        p[t * s:t * s + nr] = np.arange(t * nr, (t + 1) * nr)
        p[t * s + nr:(t + 1) * s] = H * nr + np.arange(t * nd, (t + 1) * nd)
    return p


def half_bandwidth(d):
    """row nu_i reaches back to q_i of step i - 2"""
    return 3 * (d.nr + d.nd) - 1 - d.nu


def blocked_ldl_solve(A, b, w, R=4):
    """Right-looking L D L^T of the symmetric band matrix A (half-bandwidth w), R pivots per window update, circular
    window slots, right-hand side riding along; back substitution in axpy form.  Returns (x, max |L|)."""
    N = len(b); M = w + R; MS = M + 1
    W = np.zeros((MS, MS)); yw = np.zeros(M)
    Lr = np.zeros((N, w + 1)); yg = np.zeros(N)

    def commit(i):
        if i >= N:
            return
        si = i % M
        for c in range(w + 1):
            j = i - w + c
            if j >= 0:
                sj = j % M
                W[si, sj] = A[i, j]; W[sj, si] = A[i, j]
        yw[si] = b[i]

    for i in range(min(M, N)):
        commit(i)
    lmax = 0.0
    k = 0
    while k < N:
        Rb = min(R, N - k)
        PL = np.zeros((Rb, MS)); dv = np.zeros(Rb)
        for t in range(Rb):                      # panel: left-looking inside the block
            p = k + t; sp = p % M; m = min(w, N - 1 - p)
            col = np.zeros(m + 1)
            for r in range(m + 1):
                sr = (p + r) % M
                v = W[sr, sp]
                for u in range(t):
                    v -= PL[u, sr] * dv[u] * PL[u, sp]
                col[r] = v
            d = col[0]; dv[t] = d
            yp = yw[sp]; yg[p] = yp
            Lr[p, w] = 1.0 / d
            for r in range(1, m + 1):
                sr = (p + r) % M
                l = col[r] / d
                PL[t, sr] = l; lmax = max(lmax, abs(l))
                Lr[p + r, w - r] = l
                yw[sr] -= l * yp
        lo, hi = k + Rb, min(N - 1, k + Rb - 1 + w)
        for a in range(lo, hi + 1):              # rank-Rb update of the trailing window
            sa = a % M
            for c in range(lo, hi + 1):
                sc = c % M
                W[sa, sc] -= sum(PL[t, sa] * dv[t] * PL[t, sc] for t in range(Rb))
        for t in range(Rb):
            commit(k + M + t)
        k += Rb
    x = np.zeros(N); acc = np.zeros(N)
    for i in range(N - 1, -1, -1):
        x[i] = yg[i] * Lr[i, w] - acc[i]
        for c in range(w):
            j = i - w + c
            if j >= 0:
                acc[j] += Lr[i, c] * x[i]
    return x, lmax
