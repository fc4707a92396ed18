"""CPU restatement of the banded KKT backend (TEST INFRASTRUCTURE ONLY - see oracle/__init__.py).

`kkt_banded_kernel` (contactimplicitmpc/jl_amd/csrc/kkt_dense.hip) factors the KKT matrix of `jacobian!`
(src/controller/newton_jacobian.jl:148-248) in the INTERLEAVED ordering [u_i, q_{i+2}, nu_i] per step
as L D L^T without pivoting, with a sliding (w + R) x (w + R) window in circular slots and R pivots per window update.
This module states the same algorithm in numpy - same slot logic, same order of operations per entry - so that the
structure (bandwidth, quasi-definiteness, blocked update) is checked on the CPU; the device kernel is compared with
numpy's dense solve in tests/test_gpu_parity.py.  Round 4 (second half of the file): the control elimination the kernel starts with
(`eliminate_controls`, `half_bandwidth_reduced`) and the organisation of a block as diagonal block / rows below / lower-triangle update in
a power-of-two window (`chain_bulk_ldl_solve`), and the kernel's on-the-fly row generation (`reduced_entry`), checked against the dense solve,
the one-panel form and the assembled matrix in
tests/test_oracle_reference_constructions.py::test_banded_ldl_control_elimination_and_chain_form."""
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


# ---- round 4: the forms the device kernel takes today -----------------------------------------------------------

def reduced_perm(lay, d):
    """reference layout -> (kept, u): kept = interleaved [q_{i+2}, nu_i] per step (cf mode: the whole z_i block but u_i), u = the controls"""
    H, nr, nd, nu = lay.H, d.nr, d.nd, d.nu
    kept, us = [], []
    for t in range(H):
        us += list(range(t * nr, t * nr + nu))
        kept += list(range(t * nr + nu, (t + 1) * nr)) + list(range(H * nr + t * nd, H * nr + (t + 1) * nd))
    return np.array(kept), np.array(us)


def half_bandwidth_reduced(d):
    """s = nr - nu + nd rows per step, row nu_i still reaches back to q_i of step i - 2"""
    return 3 * (d.nr - d.nu + d.nd) - 1


def eliminate_controls(R, r, lay, d):
    """u_i appears in its own block only (R_i, and du1_i in the row of nu_i): the Schur complement on the controls changes the
    nu_i x nu_i blocks and nothing else.  Returns (A_red, r_red, recover) with recover(x_kept) -> the solution in the reference layout."""
    kept, us = reduced_perm(lay, d)
    Auu = R[np.ix_(us, us)]; Aku = R[np.ix_(kept, us)]
    Ai = np.linalg.inv(Auu)
    A_red = R[np.ix_(kept, kept)] - Aku @ Ai @ Aku.T
    r_red = r[kept] - Aku @ (Ai @ r[us])

    def recover(xk):
        x = np.zeros(lay.N)
        x[kept] = xk
        x[us] = Ai @ (r[us] - Aku.T @ xk)
        return x
    return A_red, r_red, recover


def chain_bulk_ldl_solve(A, b, w, RB=8):
    """The factorisation as kkt_banded_kernel organises it since round 4, in numpy: per block of RB pivots the RB x RB diagonal block
    first (pivots, the multipliers among the pivot rows, the forward substitution among them - the device's P1), then for every row
    below the block its RB multipliers by a triangular solve against the diagonal block (P2), then the rank-RB update of the LOWER
    triangle of the trailing window (the device's MFMA tiles); window of the next power of two >= w + RB slots.  Every entry
    receives the same operations in the same order as in `blocked_ldl_solve`.  Returns (x, max |L|)."""
    N = len(b); Mw = w + RB
    M = 1
    while M < Mw:
        M <<= 1
    W = np.zeros((M, M)); yw = np.zeros(M)
    Lr = np.zeros((N, w + 1)); yg = np.zeros(N)

    def commit(i):
        if i < N:
            for c in range(w + 1):
                j = i - w + c
                if j >= 0:
                    W[i % M, j % M] = A[i, j]                    # lower triangle only: i >= j
            yw[i % M] = b[i]

    for i in range(min(Mw, N)):
        commit(i)
    lmax = 0.0
    for k in range(0, N, RB):
        nb = min(RB, N - k)
        # P1: diagonal block, right-looking
        a = np.array([[W[(k + r) % M, (k + t) % M] if t <= r else 0.0 for t in range(nb)] for r in range(nb)])
        yr = np.array([yw[(k + r) % M] for r in range(nb)])
        L11 = np.zeros((nb, nb)); dv = np.zeros(nb); yp = np.zeros(nb)
        for t in range(nb):
            dv[t] = a[t, t]; yp[t] = yr[t]
            for r in range(t + 1, nb):
                L11[r, t] = a[r, t] / dv[t]
                yr[r] -= L11[r, t] * yp[t]
            for t2 in range(t + 1, nb):
                for r in range(t2, nb):
                    a[r, t2] -= (L11[r, t] * dv[t]) * L11[t2, t]
            yg[k + t] = yp[t]; Lr[k + t, w] = 1.0 / dv[t]
            for r in range(t + 1, nb):
                Lr[k + r, w - (r - t)] = L11[r, t]; lmax = max(lmax, abs(L11[r, t]))
        # P2: rows below the block
        rows = range(k + nb, min(N, k + nb + w))
        PL = {}
        for i in rows:
            q = i - k
            l = np.zeros(nb)
            for t in range(nb):
                if q - t <= w:
                    v = W[i % M, (k + t) % M]
                    for u in range(t):
                        v -= (l[u] * dv[u]) * L11[t, u]
                    l[t] = v / dv[t]
                    Lr[i, w - (q - t)] = l[t]; lmax = max(lmax, abs(l[t]))
            PL[i] = l
            for t in range(nb):
                yw[i % M] -= l[t] * yp[t]
        # update of the lower triangle of the trailing window, entering rows
        for i in rows:
            for j in rows:
                if j <= i:
                    W[i % M, j % M] -= sum((PL[i][t] * dv[t]) * PL[j][t] for t in range(nb))
        for t in range(nb):
            commit(k + Mw + t)
    x = np.zeros(N); acc = np.zeros(N)
    for i in range(N - 1, -1, -1):
        x[i] = yg[i] * Lr[i, w] - acc[i]
        for c in range(w):
            j = i - w + c
            if j >= 0:
                acc[j] += Lr[i, c] * x[i]
    return x, lmax


def reduced_entry(d, H, obj, im, rho, i, j):
    """Entry (i, j), j <= i, of the reduced interleaved KKT matrix [q_{t+2}, nu_t] per step, as kkt_banded_kernel GENERATES it when the
    row enters the window (kkt_dense.hip: BandRows::desc / decode - nothing is assembled on the device): from the objective blocks
    (hessian!, newton_jacobian.jl:200-248), the sensitivities (update_jacobian!, :166-189), G_t = du1_t R_t^-1 du1_t^T and
    rho = H beta kappa (:169-186).  :configuration mode."""
    nq, nd = d.nq, d.nd
    s = nq + nd
    ti, ki, tj, kj = i // s, i % s, j // s, j % s
    dt = ti - tj
    if ki < nq:                                                  # q row: P block
        if kj >= nq:
            return 0.0
        if dt == 0:
            v = obj.q[ti][ki, kj]
            if obj.v is not None:
                v += obj.v[ti][ki, kj]
                if ti + 1 < H:
                    v += obj.v[ti + 1][ki, kj]
            return v
        if dt == 1 and obj.v is not None:
            return -obj.v[ti][ki, kj]
        return 0.0
    kd = ki - nq                                                 # dual row nu_t
    if dt == 0:
        if kj < nq:
            return -1.0 if kj == kd else 0.0                     # -I at q_{t+2}
        G = im["du1"][ti] @ np.linalg.solve(obj.u[ti], im["du1"][ti].T)
        return -rho - G[kd, kj - nq] if kj == ki else -G[kd, kj - nq]
    if kj >= nq:
        return 0.0
    if dt == 1:
        return im["dq1"][ti][kd, kj]                             # dq1 at q_{t+1}
    if dt == 2:
        return im["dq0"][ti][kd, kj]                             # dq0 at q_t
    return 0.0


# ---- round 5: the TWISTED (two-ended) form of the banded factorisation (kkt_dense.hip: kkt_banded_kernel<..., TW>) ----------------------

def _chain_eliminate(get, rhs, w, RB, NP, NR, import_at=None, import_fn=None):
    """`chain_bulk_ldl_solve`'s elimination on an ABSTRACT band matrix: get(i, j) (j <= i), rhs(i) generate the rows as they enter the
    window, NP pivots are eliminated (a multiple of RB unless NP = NR), NR >= NP rows exist; import_fn(W, yw, M) is called at the start
    of the block `import_at` (the top chain receives the bottom chain's trace there).  Returns (Lr, yg, W, yw, M): the rows of L
    (rows >= NP hold the multipliers w.r.t. the pivots only), the pivots' right-hand sides, and the window as it stands after the last pivot."""
    assert NP == NR or NP % RB == 0
    Mw = w + RB
    M = 1
    while M < Mw:
        M <<= 1
    W = np.zeros((M, M)); yw = np.zeros(M)
    Lr = np.zeros((NR, w + 1)); yg = np.zeros(NR)

    def commit(i):
        if i < NR:
            for c in range(w + 1):
                j = i - w + c
                if j >= 0:
                    W[i % M, j % M] = get(i, j)
            yw[i % M] = rhs(i)

    for i in range(min(Mw, NR)):
        commit(i)
    for k in range(0, NP, RB):
        if import_at is not None and k == import_at:
            import_fn(W, yw, M)
        nb = min(RB, NP - k)
        a = np.array([[W[(k + r) % M, (k + t) % M] if t <= r else 0.0 for t in range(nb)] for r in range(nb)])
        yr = np.array([yw[(k + r) % M] for r in range(nb)])
        L11 = np.zeros((nb, nb)); dv = np.zeros(nb); yp = np.zeros(nb)
        for t in range(nb):
            dv[t] = a[t, t]; yp[t] = yr[t]
            for r in range(t + 1, nb):
                L11[r, t] = a[r, t] / dv[t]
                yr[r] -= L11[r, t] * yp[t]
            for t2 in range(t + 1, nb):
                for r in range(t2, nb):
                    a[r, t2] -= (L11[r, t] * dv[t]) * L11[t2, t]
            yg[k + t] = yp[t]; Lr[k + t, w] = 1.0 / dv[t]
            for r in range(t + 1, nb):
                Lr[k + r, w - (r - t)] = L11[r, t]
        rows = range(k + nb, min(NR, k + nb + w))
        PL = {}
        for i in rows:
            q = i - k
            l = np.zeros(nb)
            for t in range(nb):
                if q - t <= w:
                    v = W[i % M, (k + t) % M]
                    for u in range(t):
                        v -= (l[u] * dv[u]) * L11[t, u]
                    l[t] = v / dv[t]
                    Lr[i, w - (q - t)] = l[t]
            PL[i] = l
            for t in range(nb):
                yw[i % M] -= l[t] * yp[t]
        for i in rows:
            for j in rows:
                if j <= i:
                    W[i % M, j % M] -= sum((PL[i][t] * dv[t]) * PL[j][t] for t in range(nb))
        for t in range(nb):
            commit(k + Mw + t)
    return Lr, yg, W, yw, M


def _chain_back_substitute(Lr, yg, w, NP, NR, x_given=None):
    """Back substitution L^T x = D^-1 y over NR rows in axpy form; rows >= NP (the middle rows of the bottom chain) take their x from
    `x_given` and contribute through their multipliers w.r.t. the pivots only (columns >= NP of those rows are not part of the factor)."""
    x = np.zeros(NR); acc = np.zeros(NR)
    for i in range(NR - 1, -1, -1):
        x[i] = x_given[i - NP] if i >= NP else yg[i] * Lr[i, w] - acc[i]
        for c in range(w):
            j = i - w + c
            if j >= 0 and (i < NP or j < NP):
                acc[j] += Lr[i, c] * x[i]
    return x


def twisted_split(N, w, RB=8):
    """(pad, Nb, m2): the bottom chain eliminates Nb pivots of the reversed matrix (a multiple of RB) of which the first `pad` are decoupled
    dummy rows appended behind the matrix so that the trace lands on a block boundary of the top chain (m2 = N + pad - Nb - w is a
    multiple of RB); the top chain eliminates rows 0 .. m2 + w - 1 and receives the trace (rows m2 .. m2 + w - 1) at the start of block m2 - RB.
    The split balances the two chains: the top chain has m2 - RB pivots behind it when it needs the trace, the bottom chain Nb
    (15 / 32 of the rows outside the middle)."""
    pad = (w - N) % RB
    Np = N + pad
    Nb = ((Np - w + RB) * 15 // 32) // RB * RB               # (the device's measured balance: the bottom chain's blocks are the slower ones)
    Nb = max(RB, min(Nb, Np - w - 2 * RB))
    return pad, Nb, Np - Nb - w


def twisted_chain_bulk_ldl_solve(A, b, w, RB=8):
    """The banded L D L^T as TWO chains (the device: two workgroups per rollout).  BOTTOM chain: the reversed matrix
    A'[i', j'] = A[N-1-(i'-pad), N-1-(j'-pad)] (pad decoupled dummy rows first), Nb pivots; the rows behind its last pivot - matrix rows
    m2 .. m2 + w - 1 in reversed order - enter its window with their couplings to the pivots but a ZERO middle block and right-hand side,
    so that after the last pivot the window holds exactly what the eliminated rows contribute there (the trace).  TOP chain: rows
    0 .. m2 + w - 1 of the matrix; at the start of block m2 - RB (all trace rows are in its window, none is a pivot yet) it adds the
    trace, then runs on to its last row.  Back substitution of the top chain first (its first w values are the middle rows'), then the
    bottom chain's with those values given.  Returns x."""
    N = len(b)
    pad, Nb, m2 = twisted_split(N, w, RB)
    Nt = m2 + w
    assert m2 % RB == 0 and Nb % RB == 0 and m2 >= RB and Nb - pad >= 1
    nbr = Nb - pad                                               # real rows the bottom chain eliminates: matrix rows N-1 .. N-nbr = Nt

    def orig(ip):                                                # reversed index -> matrix index (dummy rows: -1)
        return N - 1 - (ip - pad) if ip >= pad else -1

    def get_b(ip, jp):
        if ip < pad or jp < pad:
            return 1.0 if ip == jp else 0.0
        if ip - pad >= nbr and jp - pad >= nbr:
            return 0.0                                           # middle block: the top chain's
        return A[orig(jp), orig(ip)]                             # (jp <= ip: matrix column >= row - the symmetric entry)

    def rhs_b(ip):
        return 0.0 if (ip < pad or ip - pad >= nbr) else b[orig(ip)]
    Lb, yb, Wb, ywb, M = _chain_eliminate(get_b, rhs_b, w, RB, Nb, Nb + w)
    S = np.zeros((w, w)); c = np.zeros(w)                        # trace in MATRIX order: row a <-> matrix row m2 + a = reversed row Nb + w - 1 - a
    for a in range(w):
        ia = Nb + w - 1 - a
        c[a] = ywb[ia % M]
        for a2 in range(a + 1):                                  # matrix (m2 + a, m2 + a2), a2 <= a  <->  reversed (ia2 >= ia): lower entry W[ia2, ia]
            ia2 = Nb + w - 1 - a2
            S[a, a2] = Wb[ia2 % M, ia % M]

    def import_fn(W, yw, Mt):
        for a in range(w):
            yw[(m2 + a) % Mt] += c[a]
            for a2 in range(a + 1):
                W[(m2 + a) % Mt, (m2 + a2) % Mt] += S[a, a2]
    Lt, yt, _, _, _ = _chain_eliminate(lambda i, j: A[i, j], lambda i: b[i], w, RB, Nt, Nt, import_at=m2 - RB, import_fn=import_fn)
    xt = _chain_back_substitute(Lt, yt, w, Nt, Nt)
    xb = _chain_back_substitute(Lb, yb, w, Nb, Nb + w, x_given=np.array([xt[m2 + w - 1 - a] for a in range(w)]))
    x = np.zeros(N)
    x[:Nt] = xt
    for ip in range(pad, Nb):
        x[orig(ip)] = xb[ip]
    return x
