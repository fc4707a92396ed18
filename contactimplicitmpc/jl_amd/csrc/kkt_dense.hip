// Reference-default KKT path for ANY mode / objective: jacobian! assembled densely
// (/root/reference/src/controller/newton_jacobian.jl:148-248: hessian! incl. the TrackingVelocityObjective
// off-diagonal blocks :218-248, IV / ITV :157-161, update_jacobian! :166-189 with the reg_du quirk
// rho = H*beta*kappa) and solved by LU with partial pivoting, the reference's default
// `opts.solver = :lu_solver` (newton.jl:10,86,218; lu.jl:4-12 -> RoboDojo LUSolver).
//
// One workgroup per rollout, the N x N matrix (N = H (nr + nd)) column-major in a global workspace.
// This is the general fallback: it covers :configurationforce (gamma / b weights of 1e-100 make the
// primal Hessian unusable for the condensed solve) and the velocity objective; the MFMA condensed
// kernel (newton_impl.h) stays the fast path of :configuration + TrackingObjective.  Work is
// (2/3) N^3 flops per solve (pushbot cf H = 10: N = 180; quadruped: N = 1200).
#include "newton_impl.h"

namespace cimpc {

namespace {

struct DenseLayout {      // newton_residual.jl:18-54 / 69-98 (0-based offsets inside one primal block)
    int nq, nu, nc, nb, nr, nd, H, N, oq;
    bool cf;
    __device__ DenseLayout(const NewtonDev& S)
        : nq(S.dm.nq), nu(S.dm.nu), nc(S.dm.nc), nb(S.dm.nb), nr(S.nr), nd(S.nd), H(S.dm.H), N(S.N),
          cf(S.dm.mode == CIMPC_MODE_CONFIGURATIONFORCE) { oq = cf ? nu + nc + nb : nu; }
    __device__ int pu(int t, int k) const { return t * nr + k; }
    __device__ int pq(int t, int k) const { return t * nr + oq + k; }
    __device__ int pz(int t, int k) const {        // iz = [iq; ig; ib]
        if (k < nq) return t * nr + oq + k;
        return t * nr + nu + (k - nq);              // gamma then b follow u inside the block
    }
    __device__ int dual(int t, int k) const { return H * nr + t * nd + k; }
};

}  // namespace

// LU with partial pivoting of the N x N column-major matrix A (in place) and solve of A x = b with b given in x
// (getrf-style, right-looking; the right-hand side is permuted and forward-substituted on the fly), one workgroup.
// `singular` (optional): set to 1 when a pivot column is exactly zero or not finite - the situation in which the reference's
// lu_solver throws SingularException at this seam (lu.jl:4-12 -> LinearAlgebra.lu!)
__device__ void dense_lu_solve(double* A, double* x, int* piv, int N, int tid, int nt, double* s_val, int* s_idx, int* singular = nullptr) {
    for (int k = 0; k < N; ++k) {
        double best = -1.0; int bi = k;
        for (int i = k + tid; i < N; i += nt) {
            const double v = fabs(A[(size_t)k * N + i]);
            if (v > best) { best = v; bi = i; }
        }
        s_val[tid] = best; s_idx[tid] = bi;
        __syncthreads();
        for (int s = nt / 2; s > 0; s >>= 1) {
            if (tid < s) {
                const double o = s_val[tid + s];
                if (o > s_val[tid] || (o == s_val[tid] && s_idx[tid + s] < s_idx[tid])) { s_val[tid] = o; s_idx[tid] = s_idx[tid + s]; }
            }
            __syncthreads();
        }
        const int p = s_idx[0];
        const double pmag = s_val[0];
        __syncthreads();
        if (tid == 0 && singular != nullptr && !(pmag > 0.0 && pmag < HUGE_VAL)) *singular = 1;
        if (p != k) {
            for (int j = tid; j < N; j += nt) {
                const double t0 = A[(size_t)j * N + k];
                A[(size_t)j * N + k] = A[(size_t)j * N + p];
                A[(size_t)j * N + p] = t0;
            }
            if (tid == 0) { const double t0 = x[k]; x[k] = x[p]; x[p] = t0; }
        }
        if (tid == 0) piv[k] = p;
        __syncthreads();
        const double inv = 1.0 / A[(size_t)k * N + k];
        for (int i = k + 1 + tid; i < N; i += nt) A[(size_t)k * N + i] *= inv;
        __syncthreads();
        const int m = N - k - 1;
        // trailing update, columns j > k: A[i,j] -= l[i] * u[j]; rows are the fast (coalesced) index
        for (size_t e = tid; e < (size_t)m * m; e += nt) {
            const int i = k + 1 + (int)(e % m), j = k + 1 + (int)(e / m);
            A[(size_t)j * N + i] = fma(-A[(size_t)k * N + i], A[(size_t)j * N + k], A[(size_t)j * N + i]);
        }
        // forward substitution of the right-hand side rides along: x[i] -= l[i] * x[k]
        for (int i = k + 1 + tid; i < N; i += nt) x[i] = fma(-A[(size_t)k * N + i], x[k], x[i]);
        __syncthreads();
    }
    // ---- back substitution with U ------------------------------------------------------------------
    for (int k = N - 1; k >= 0; --k) {
        if (tid == 0) x[k] /= A[(size_t)k * N + k];
        __syncthreads();
        const double xk = x[k];
        for (int i = tid; i < k; i += nt) x[i] = fma(-A[(size_t)k * N + i], xk, x[i]);
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void kkt_dense_kernel(NewtonDev S, KktArgs K, double* ws_all) {
    const int b = blockIdx.x + S.b0, tid = threadIdx.x, nt = blockDim.x;
    if (K.stage != nullptr && K.stage[b] != STAGE_KKT) return;
    const DenseLayout L(S);
    const int N = L.N, H = L.H, nq = L.nq, nu = L.nu, nd = L.nd, nths = S.nths;
    double* A = ws_all + (size_t)b * ((size_t)N * N + 2 * N);
    double* x = A + (size_t)N * N;                 // right-hand side / solution
    int* piv = reinterpret_cast<int*>(x + N);
    const double beta = K.beta ? K.beta[b] : K.beta_scalar;
    const double rho = (double)H * beta * S.kappa;  // newton_jacobian.jl:169-186 quirk
    const double* dzb = kkt_dz(S, K, b, H, nths, nd);
    __shared__ double s_val[256];
    __shared__ int s_idx[256];

    // ---- jacobian!: dense assembly ---------------------------------------------------------------
    for (size_t e = tid; e < (size_t)N * N; e += nt) A[e] = 0.0;
    __syncthreads();
    auto at = [&](int r, int c) -> double& { return A[(size_t)c * N + r]; };
    for (int e = tid; e < H * nq * nq; e += nt) {               // hessian!: q blocks (+ velocity terms)
        const int t = e / (nq * nq), k = e - t * nq * nq, r = k % nq, c = k / nq;
        double v = S.Q[(size_t)t * nq * nq + k];
        if (S.V != nullptr) {
            v += S.V[(size_t)t * nq * nq + k];
            if (t + 1 < H) v += S.V[(size_t)(t + 1) * nq * nq + k];
            if (t >= 1) {
                const double o = S.V[(size_t)t * nq * nq + k];
                at(L.pq(t - 1, r), L.pq(t, c)) = -o;
                at(L.pq(t, r), L.pq(t - 1, c)) = -o;
            }
        }
        at(L.pq(t, r), L.pq(t, c)) = v;
    }
    for (int e = tid; e < H * nu * nu; e += nt) {
        const int t = e / (nu * nu), k = e - t * nu * nu;
        at(L.pu(t, k % nu), L.pu(t, k / nu)) = S.R[(size_t)t * nu * nu + k];
    }
    if (L.cf) {
        const int nc = L.nc, nb = L.nb;
        for (int e = tid; e < H * nc * nc; e += nt) {
            const int t = e / (nc * nc), k = e - t * nc * nc;
            at(t * L.nr + nu + k % nc, t * L.nr + nu + k / nc) = S.Cg[(size_t)t * nc * nc + k];
        }
        for (int e = tid; e < H * nb * nb; e += nt) {
            const int t = e / (nb * nb), k = e - t * nb * nb;
            at(t * L.nr + nu + nc + k % nb, t * L.nr + nu + nc + k / nb) = S.Cb[(size_t)t * nb * nb + k];
        }
    }
    for (int e = tid; e < H * nd; e += nt) {                    // IV / ITV and the dual regularisation
        const int t = e / nd, k = e - t * nd;
        at(L.pz(t, k), L.dual(t, k)) = -1.0;
        at(L.dual(t, k), L.pz(t, k)) = -1.0;
        at(L.dual(t, k), L.dual(t, k)) = -rho;
    }
    __syncthreads();
    for (int e = tid; e < H * nths * nd; e += nt) {             // update_jacobian!: sensitivities
        const int i = e / (nths * nd), k = e - i * nths * nd, r = k % nd, c = k / nd;
        const double v = dzb[e];
        int col;
        if (c < nq) { if (i < 2) continue; col = L.pq(i - 2, c); }
        else if (c < 2 * nq) { if (i < 1) continue; col = L.pq(i - 1, c - nq); }
        else col = L.pu(i, c - 2 * nq);
        at(L.dual(i, r), col) += v;        // the -1 of IV sits in another column block for q_{i}, so += is safe
        at(col, L.dual(i, r)) += v;
    }
    for (int e = tid; e < N; e += nt) x[e] = K.r[(size_t)b * S.N + e];
    __syncthreads();

    dense_lu_solve(A, x, piv, N, tid, nt, s_val, s_idx);
    double* D = K.delta + (size_t)b * S.N;
    for (int e = tid; e < N; e += nt) D[e] = x[e];
    __syncthreads();
    if (K.finish) {
        __threadfence_block();
        start_line_search<BlockSync>(S, b, 1, tid, nt);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Banded backend for :configuration + TrackingVelocityObjective (block-tridiagonal P: the condensed dual Schur
// complement would be dense).  The same KKT matrix in the INTERLEAVED ordering [u_i, q_{i+2}, nu_i] per step
// ("the form the HIP KKT kernel should factor") is symmetric, banded with half-bandwidth
// w = 3 (nr + nd) - 1 - nu (row nu_i reaches back to q_i of step i-2) and quasi-definite (P > 0, -rho I < 0), so the
// LDL^T factorization needs no pivoting in this order (measured on assembled jacobian! matrices: |L| <= 1e2..1e4,
// backward error 3e-15..3e-14).  One workgroup per rollout; the active (w+1) x (w+1) lower-triangular window of
// the right-looking elimination lives in LDS (circular row / column slots, no data movement), every matrix row
// is GENERATED when it enters the window (no assembled matrix in memory), the right-hand side rides along, the
// rows of L go to a global workspace for the back substitution (one wavefront, axpy form, sums in registers, rows staged through LDS).
// Round 4 rebuilt the kernel around its sequential part - see the comments at the block loop and DESIGN.md 5.2c.
// Work: N w^2 / 2 multiply-adds (centroidal H = 60: 25 M, against (2/3) N^3 = 16 G of the dense LU).
// ---------------------------------------------------------------------------------------------------------
namespace {
struct BandRows {
    const NewtonDev& S; const DenseLayout& L; const double* dzb; double rho; int s, nths;
    const double* G;      // reduced form (controls eliminated): [H][nd x nd] du1_t R_t^-1 du1_t^T, column-major; nullptr = full form
    // An entry is a constant or a combination of at most three values in memory.  The two halves are separate so that a caller can
    // ISSUE the loads of an entry long before it needs the value (the rows entering the window are fetched a block ahead: with the
    // arithmetic next to the loads the fetching wavefronts sat out a memory latency per block - 3.7 k of 14 k cycles).
    enum : int { CONST = 0, PLAIN = 1, NEG = 2, SUM = 3, C_MINUS = 4, DIFF = 5 };
    //   CONST c   PLAIN x0   NEG -x0   SUM (x0 + x1) + x2   C_MINUS c - x0   DIFF x0 - x1      (absent operands: null pointer, read as 0)
    // One expression for all of them, ((c + s0 x0) + s1 x1) + x2 with s = -1 / +1 (absent operands and c read as 0: every mode's own
    // roundings - adding zero is exact) - no branches where sixty-four lanes hold sixty-four different modes.
    static __device__ __forceinline__ double combine(int mode, double c, double x0, double x1, double x2) {
        const double a0 = (mode == NEG || mode == C_MINUS) ? -x0 : x0, a1 = mode == DIFF ? -x1 : x1;
        return ((c + a0) + a1) + x2;
    }
    // What an entry is depends on (row index in its step, column index in its step, steps between them) only - not on the step:
    // a 32-bit DESCRIPTOR  mode | array << 3 | constant << 6 | offset inside the step's block << 8.  The kernel tabulates the
    // descriptors of one period (s rows x w + 1 columns) once per launch; the rows entering the window then cost a table look-up
    // and a decode instead of this case analysis per entry (round 4: the analysis, divergent across the lanes, was 3 k of 13 k
    // cycles per block on fourteen of the sixteen wavefronts).
    enum : int { AQ = 0, AV = 1, AG = 2, ADZ = 3, AR = 4 };      // arrays: Q, V, G (reduced form), dz (this rollout), R
    enum : int { C0 = 0, CM1 = 1, CMRHO = 2 };                   // constants: 0, -1, -rho
    static __device__ __forceinline__ unsigned pack(int mode, int arr, int cid, int off) { return (unsigned)mode | (unsigned)arr << 3 | (unsigned)cid << 6 | (unsigned)off << 8; }
    __device__ unsigned desc(int dt, int ki, int kj) const {
        const int nq = L.nq, nu = L.nu, nr = L.nr, nd = L.nd;
        auto qrow = [&](int kq, int cq) {                        // P block: Q_t (+ V_t + V_{t+1}) on the diagonal step, -V_t one step back
            const int e = cq * nq + kq;
            if (dt == 0) return pack(SUM, AQ, C0, e);
            if (dt == 1 && S.V != nullptr) return pack(NEG, AV, C0, e);
            return pack(CONST, 0, C0, 0);
        };
        if (G != nullptr) {      // ordering [q_{t+2}, nu_t] per step, s = nq + nd
            if (ki < nq) {                                       // q row
                if (kj >= nq) return pack(CONST, 0, C0, 0);
                return qrow(ki, kj);
            }
            const int kd = ki - nq;                              // dual row nu_t
            if (dt == 0) {
                if (kj < nq) return pack(CONST, 0, kj == kd ? CM1 : C0, 0);      // -I at q_{t+2}
                const int off = (kj - nq) * nd + kd;
                return kj == ki ? pack(C_MINUS, AG, CMRHO, off) : pack(NEG, AG, C0, off);      // -(rho I + du1 R^-1 du1^T)
            }
            if (kj >= nq) return pack(CONST, 0, C0, 0);
            if (dt == 1) return pack(PLAIN, ADZ, C0, (nq + kj) * nd + kd);      // dq1 at q_{t+1}
            if (dt == 2) return pack(PLAIN, ADZ, C0, kj * nd + kd);             // dq0 at q_t
            return pack(CONST, 0, C0, 0);
        }
        if (ki < nu) {                                           // u row: R_t (same block only)
            if (dt == 0 && kj < nu) return pack(PLAIN, AR, C0, kj * nu + ki);
            return pack(CONST, 0, C0, 0);
        }
        if (ki < nr) {                                           // q row
            if (kj < nu || kj >= nr) return pack(CONST, 0, C0, 0);      // (lower part: no dual columns before a q row)
            return qrow(ki - nu, kj - nu);
        }
        const int kd = ki - nr;                                  // dual row nu_t
        if (dt == 0) {
            if (kj < nu) return pack(PLAIN, ADZ, C0, (2 * nq + kj) * nd + kd);          // du1
            if (kj < nr) return pack(CONST, 0, kj - nu == kd ? CM1 : C0, 0);            // -I at q_{t+2}
            return pack(CONST, 0, kj == ki ? CMRHO : C0, 0);
        }
        if (kj < nu || kj >= nr) return pack(CONST, 0, C0, 0);
        if (dt == 1) return pack(PLAIN, ADZ, C0, (nq + kj - nu) * nd + kd);             // dq1 at q_{t+1}
        if (dt == 2) return pack(PLAIN, ADZ, C0, (kj - nu) * nd + kd);                  // dq0 at q_t
        return pack(CONST, 0, C0, 0);
    }
    // descriptor + step of the row -> mode, constant, operand addresses (selects, one branch for the three-operand sum)
    __device__ __forceinline__ int decode(unsigned d, int ti, const double*& p0, const double*& p1, const double*& p2, double& c) const {
        const int mode = d & 7, arr = (d >> 3) & 7, cid = (d >> 6) & 3, off = (int)(d >> 8);
        c = cid == C0 ? 0.0 : (cid == CM1 ? -1.0 : -rho);
        const int nq2 = L.nq * L.nq;
        const double* base = arr == AQ ? S.Q : arr == AV ? S.V : arr == AG ? G : arr == ADZ ? dzb : S.R;
        const int stride = arr <= AV ? nq2 : arr == AG ? L.nd * L.nd : arr == ADZ ? nths * L.nd : L.nu * L.nu;
        p0 = mode == CONST ? nullptr : base + (ti * stride + off);
        p1 = p2 = nullptr;
        if (mode == SUM && S.V != nullptr) {
            p1 = S.V + (ti * nq2 + off);
            if (ti + 1 < L.H) p2 = p1 + nq2;
        }
        return mode;
    }
    // entry of row (ti, ki), column (tj, kj) - (step, index in the step), j <= i - of the interleaved KKT matrix: mode, constant and operand addresses
    __device__ int terms4(int ti, int ki, int tj, int kj, const double*& p0, const double*& p1, const double*& p2, double& c) const {
        return decode(desc(ti - tj, ki, kj), ti, p0, p1, p2, c);
    }
};
}  // namespace

// Workgroup barrier that waits for the wave's LDS traffic only: the pivots hand data over through LDS, while
// the global stores of the factor rows and the prefetch loads of the next matrix row stay in flight across
// them (a __syncthreads() waits for vmcnt(0) too - measured 6.4 us per pivot, almost all of it store / load
// latency at the two barriers).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// doubles of workspace per rollout: rows of L (N x (w + 1)), y (N), then G and g of the reduced form, then the descriptors of one period of rows.  Sized for the FULL form
// (the larger one) so that the host's allocation does not depend on which form a launch takes.
__host__ __device__ inline size_t banded_ws_per_rollout(const NewtonDev& S) {
    const size_t sF = (size_t)S.nr + S.nd, NF = (size_t)S.dm.H * sF;
    size_t wF = 3 * sF - 1 - S.dm.nu; if (wF > NF - 1) wF = NF - 1;
    const size_t one = NF * (wF + 1) + NF + (size_t)S.dm.H * S.nd * (S.nd + 1) + (sF * (wF + 2) + 1) / 2;      // ... and the descriptor table (32-bit words)
    // twisted form (two chains per rollout, kkt_banded_twisted_kernel): the chains' rows of L and y together are w + 8 more than
    // N (the bottom chain keeps the multiplier rows of the w middle rows, <= 7 dummy rows pad its matrix), each chain has its own
    // G / g / descriptor table, and the bottom chain hands its trace (w x w + w) to the top chain
    const size_t tw = (wF + 8) * (wF + 2) + (size_t)S.dm.H * S.nd * (S.nd + 1) + (sF * (wF + 2) + 1) / 2 + wF * (wF + 1);
    return one + tw;
}
// Split of the twisted banded factorisation (CPU statement: the test oracle's banded.py: twisted_split / twisted_chain_bulk_ldl_solve):
// `pad` decoupled dummy rows behind the matrix make N + pad - w a multiple of RB; the bottom chain eliminates Nb pivots (a multiple of
// RB, the dummies first) of the reversed matrix, the top chain rows 0 .. m2 + w - 1 with m2 = N + pad - Nb - w (a multiple of RB): it
// receives the bottom chain's trace on rows m2 .. m2 + w - 1 at the start of block m2 - RB.
struct BandSplit { int pad, Nb, m2; };
__host__ __device__ inline BandSplit banded_twisted_split(int N, int w, int RB) {
    BandSplit sp;
    sp.pad = ((w - N) % RB + RB) % RB;
    const int Np = N + sp.pad;
    // (15 / 32 of the rows outside the middle: the bottom chain's block takes 3.4 us against the top chain's 3.2, and its trace has to
    //  be written and read - profiles/r05/twisted_banded_prof_a.log: with an even split the top chain waited 57 us of 790)
    int Nb = ((Np - w + RB) * 15 / 32) / RB * RB;
    const int hi = Np - w - 2 * RB;
    if (Nb > hi) Nb = hi / RB * RB;
    if (Nb < RB) Nb = RB;
    sp.Nb = Nb; sp.m2 = Np - Nb - w;
    return sp;
}
constexpr int KKT_BAND_TW_FLAG0 = 4;      // words of NewtonDev::kkt_tw_flags[b]: 4 trace ready, 5 middle x ready, 6 chains finished

// -DCIMPC_BANDED_PROF (diagnostic builds, with -DCIMPC_KKT_PROF for the accessor): shader-clock accounting of the banded kernel, rollout 0,
// per wavefront: [0] pre-pass, window fill, first diagonal block  [1] P2  [2] barrier  [3] wavefront 0: tile (0, 0) / the others: entering rows, rhs, stores
// [4] wavefront 0: next diagonal block / the others: tiles  [5] fetch  [6] barrier  [7] -  [8] back substitution  [9] control
// recovery; wavefront v -> stats[8 + 10 v ..] (the handle needs B >= 42 rollouts for the room)
#ifdef CIMPC_BANDED_PROF
#define BPROF(j) { const long long tn_ = clock64(); bp[j] += tn_ - bt; bt = tn_; }
#else
#define BPROF(j)
#endif
#ifndef CIMPC_BANDED_THREADS
#define CIMPC_BANDED_THREADS 1024      // (256 / 512 / 1024 measured on the round-3 form)
#endif
static_assert(CIMPC_BANDED_THREADS == 1024, "the block phases are laid out for sixteen wavefronts: P2 covers 1024 / 16 x (16 / RB) rows below a block, "
                                            "wavefront 0 is the chain, wavefronts 1 .. 15 the bulk");
// RB: pivots per window update (4, or 8 where the window of w + 8 slots fits the LDS: the reduced form of every compiled model).  The
// per-entry cost of an update pass is index arithmetic and LDS traffic, not its RB multiply-adds (profiles/r04/banded_prof_before.log:
// update + its barrier 43 % of a solve, 7-8 k cycles per pass at RB = 4), so half as many passes are worth the deeper panel.  Every entry
// still receives its pivots' contributions one after the other in pivot order: the factors do not depend on RB.
// a += (c of lane K0 of this lane's DPP row of 16) * l0 + (c of lane K1) * l1, one instruction per multiply-add (lane_group.h:
// v_fmac_f64_dpp row_newbcast and its hazard rule - one s_nop 1 after c is written).  (A bank mask on the destination instead of a zero
// factor was tried and gives wrong results: the 64-bit DPP forms do not honour it.)
template <int K0, int K1>
__device__ __forceinline__ void fmac_bcast2(double& a, double c, double l0, double l1) {
    asm("s_nop 1\n\t"
        "v_fmac_f64_dpp %[a], %[c], %[l0] row_newbcast:%[k0] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %[a], %[c], %[l1] row_newbcast:%[k1] row_mask:0xf bank_mask:0xf\n\t"
        : [a] "+v"(a) : [c] "v"(c), [l0] "v"(l0), [l1] "v"(l1), [k0] "n"(K0), [k1] "n"(K1));
}
// POW2: the window has a power-of-two number of slots (where that fits the LDS): a slot index wraps with one AND instead of two
// compare-subtract pairs - the wraps are a sixth of the instructions of a block, and every phase but P1 is issue-bound.
__host__ __device__ inline int banded_pow2_slots(int extent) { int p = 1; while (p < extent) p <<= 1; return p; }
// SLOTS: that power of two as a compile-time constant (32 / 64 / 128: every LDS offset and the row stride SLOTS + 1 fold into the
// instructions - fewer scalar registers to spill, shift-adds for the addresses), -1: power of two known at run time, 0: w + RB slots.
// TW = 0: the whole matrix on one workgroup.  TW = 1 / 2 (round 5): the TOP / BOTTOM chain of the twisted form, two workgroups per rollout
// (kkt_banded_twisted_kernel below; needs the reduced form, RB = 8 and a compile-time power-of-two window).
template <int RB, int SLOTS, int TW>
__device__ __forceinline__ void kkt_banded_body(const NewtonDev& S, const KktArgs& K, double* ws_all, double* sm, int b) {
    constexpr bool POW2 = SLOTS != 0;
    static_assert(TW == 0 || (SLOTS > 0 && RB == 8), "twisted chains: compile-time power-of-two window, blocks of eight");
    constexpr bool REV = TW == 2;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (K.stage != nullptr && K.stage[b] != STAGE_KKT) return;
#ifdef CIMPC_KKT_TWPROF
    // (diagnostic builds) constant-rate clock stamps of the twisted banded chains -> statistics words 8 + 8 (TW - 1) + j
#define BTWSTAMP(j) { if (TW != 0 && tid == 0 && b == 0) ((long long*)S.stats)[8 + 8 * (TW - 1) + (j)] = (long long)wall_clock64(); }
#else
#define BTWSTAMP(j) {}
#endif
    BTWSTAMP(0)
    const DenseLayout L(S);
    // Round 4: the controls are eliminated first.  u_t appears in its own block only (R_t, and du1_t in the row of nu_t), so
    //   Du_t = R_t^-1 (r_u,t - du1_t^T Dnu_t)   and the nu_t row becomes   ... - (rho I + du1_t R_t^-1 du1_t^T) Dnu_t = r_d,t - du1_t R_t^-1 r_u,t
    // exactly: the factored matrix is the interleaved KKT matrix of [q_{t+2}, nu_t] - still symmetric quasi-definite - with
    // s = nq + nd rows per step instead of nr + nd and half-bandwidth 3 s - 1 (centroidal H = 60: N 2880 -> 2160, w 131 -> 107,
    // N w^2 49 M -> 25 M, a quarter fewer pivots on the sequential chain).  S.band_reduce = 0 (an R_t the host could not
    // invert) keeps the full form.
    const bool reduced = S.band_reduce != 0 && L.nu > 0;
    const int H = L.H, nr = L.nr, nd = L.nd, nq = L.nq, nu = L.nu;
    const int s = reduced ? nq + nd : nr + nd, N = H * s;      // N: rows of the whole matrix
    const int w = min(reduced ? 3 * s - 1 : 3 * s - 1 - nu, N - 1), LW = w + 1;
    // chain geometry: NP pivots are eliminated, NR >= NP rows exist in this chain's matrix
    const BandSplit sp = TW != 0 ? banded_twisted_split(N, w, RB) : BandSplit{0, 0, 0};
    const int NP = TW == 1 ? sp.m2 + w : TW == 2 ? sp.Nb : N;
    const int NR = TW == 2 ? sp.Nb + w : NP;
    [[maybe_unused]] const int pad = sp.pad, nbr = sp.Nb - sp.pad, m2 = sp.m2;      // nbr: matrix rows the bottom chain eliminates (N-1 .. N-nbr)
    [[maybe_unused]] int* const xfl = TW != 0 ? S.kkt_tw_flags + (size_t)b * KKT_TW_FLAGS + KKT_BAND_TW_FLAG0 : nullptr;
    [[maybe_unused]] bool tw_ok = true;
    const int Mw = w + RB;                                       // rows in the window: pivot k+RB-1 reaches row k+RB-1+w
    const int M = SLOTS > 0 ? SLOTS : (POW2 ? banded_pow2_slots(Mw) : Mw);      // slots (index mod M)
    double* wsb = ws_all + (size_t)b * banded_ws_per_rollout(S);
    // (twisted: rows of L / y of the top chain first, the bottom chain's behind them; G / g / descriptors per chain; then the trace)
    const int rows_all = TW != 0 ? N + w + 8 : N;
    double* Lr = wsb + (TW == 2 ? (size_t)(m2 + w) * LW : 0);    // row i: L[i][i-w .. i-1], slot w: 1 / d_i
    double* yg = wsb + (size_t)rows_all * LW + (TW == 2 ? m2 + w : 0);
    const size_t g_doubles = (size_t)H * nd * (nd + 1) + ((size_t)s * (w + 2) + 1) / 2;
    double* Gm = wsb + (size_t)rows_all * (LW + 1) + (TW == 2 ? g_doubles : 0);      // reduced form: [H][nd x nd] du1 R^-1 du1^T, then [H][nd] du1 R^-1 r_u
    [[maybe_unused]] double* const xtr = wsb + (size_t)rows_all * (LW + 1) + 2 * g_doubles;      // trace [w x w] (matrix order, lower triangle), then [w] right-hand side
    const int MS = M + 1;                                        // row stride: slot M is a dummy row / column
    double* W = sm;                                              // [M+1][M+1] window, slot = index mod M; BOTH triangles kept
    double* yw = W + (size_t)MS * MS;                            // [M]   right-hand side of the rows in the window
    double* PL = yw + MS;                                        // [RB][M+1] multipliers of the block's pivots, by row SLOT
    double* dv = PL + (size_t)RB * MS;                           // [2][RB]  the block's pivots d            (by block parity, like the next three)
    double* dinv = dv + 2 * RB;                                  // [2][RB]  their reciprocals
    double* ypv = dinv + 2 * RB;                                 // [2][RB]  the pivot rows' right-hand sides (after the forward substitution among them)
    double* L11 = ypv + 2 * RB;                                  // [2][RB][RB] multipliers among the block's pivot rows (row-major, strictly lower)
    double* nL11 = L11 + 2 * RB * RB;                            // [2][RB + 1][RB] the same negated, and a row of zeros (P2's operand: no selects)
    int* tile_tab = (int*)(nL11 + 2 * (RB + 1) * RB);            // [<= 96] update tiles (I | J << 8), tile column 0 first
    // parameters of the entering rows' decode, read from LDS by array index (as scalar registers they were spilled: the fetch was 250
    // instructions, a third of them v_readlane reloads):
    unsigned long long* f_base = (unsigned long long*)(tile_tab + 96);      // [8] Q, V, G, dz, R (BandRows::AQ ..), then r, g
    int* f_stride = (int*)(f_base + 8);                          // [8] doubles per step of those arrays
    double* f_const = (double*)(f_stride + 8);                   // [4] 0, -1, -rho (BandRows::C0 ..)
    const double beta = K.beta ? K.beta[b] : K.beta_scalar;
    const double rho = (double)H * beta * S.kappa;               // newton_jacobian.jl:169-186 quirk
    const double* dzb = kkt_dz(S, K, b, H, S.nths, nd);
    const double* rb = K.r + (size_t)b * S.N;
    double* gv = Gm + (size_t)H * nd * nd;
    unsigned* dtab = (unsigned*)(gv + (size_t)H * nd);           // [s][w + 2] descriptors of the rows of one step (BandRows::desc)
#ifdef CIMPC_BANDED_PROF
    long long bp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, bt = clock64();
#endif
    if (reduced) {      // G_t = du1_t R_t^-1 du1_t^T, g_t = du1_t R_t^-1 r_u,t  (R_t^-1: NewtonDev::Rinv, symmetric); T = du1 R^-1 staged in LDS
        double* Tm = sm;                                         // [H][nd x nu] (the window is not in use yet)
        // (twisted: a chain forms G / g of the steps its rows touch only - the top chain rows < m2 + w, the bottom chain rows >= m2)
        const int tg0 = TW == 2 ? m2 / s : 0, tg1 = TW == 1 ? min(H, (m2 + w + s - 1) / s) : H, ntg = tg1 - tg0;
        for (int e = tid + tg0 * nd * nu; e < tg1 * nd * nu; e += nt) {
            const int t = e / (nd * nu), k = e - t * nd * nu, r = k % nd, c = k / nd;
            const double* du = dzb + ((size_t)t * S.nths + 2 * nq) * nd;      // du1_t, nd x nu column-major
            const double* Ri = S.Rinv + (size_t)t * nu * nu + (size_t)c * nu;
            double a = 0.0;
            for (int m = 0; m < nu; ++m) a = fma(du[(size_t)m * nd + r], Ri[m], a);
            Tm[e] = a;
        }
        __syncthreads();
        (void)ntg;
        for (int e = tid + tg0 * nd * (nd + 1); e < tg1 * nd * (nd + 1); e += nt) {
            const int t = e / (nd * (nd + 1)), k = e - t * nd * (nd + 1), r = k % nd, c = k / nd;
            const double* Tr = Tm + (size_t)t * nd * nu;
            double a = 0.0;
            if (c < nd) {
                const double* du = dzb + ((size_t)t * S.nths + 2 * nq) * nd;
                for (int m = 0; m < nu; ++m) a = fma(Tr[(size_t)m * nd + r], du[(size_t)m * nd + c], a);
                Gm[(size_t)t * nd * nd + (size_t)c * nd + r] = a;
            } else {
                for (int m = 0; m < nu; ++m) a = fma(Tr[(size_t)m * nd + r], rb[t * nr + m], a);
                gv[t * nd + r] = a;
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    const BandRows row{S, L, dzb, rho, s, S.nths, reduced ? Gm : nullptr};
    // (interleaved index i = t s + k -> index in the reference's layout: primal segment step-major, then the duals)
    // A row entering the window has w + 2 entries: columns i-w .. i and the right-hand side.  Only the LOWER triangle of the window is
    // kept (row >= column: nothing ever reads the other one).  The entries of the RB rows that enter per block belong to the threads from
    // 128 on, at most NE each, and are FETCHED one block ahead (global loads of the sensitivities / weights stay off the critical path).
    constexpr int NE = 2;
    const int EW = w + 2;
    const bool need2 = RB * EW > (int)blockDim.x - 128;          // (uniform: the second entry per thread exists only for the widest bands)
    struct Pending { double x0, x1, x2, c; int mode; };         // an entry whose loads are in flight
    // entry e of row i = ti s + ki; (tj, kj): its column i - w + e likewise (unused for the right-hand side, e = w + 1).  Loads only.
    auto entry_issue = [&](int mode, int e, int ti, int ki, const double* p0, const double* p1, const double* p2, Pending& P) {
        if (e > w) {                                             // right-hand side (reduced form: the dual rows carry r_d - du1 R^-1 r_u)
            p0 = rb + (reduced ? (ki < nq ? ti * nr + nu + ki : H * nr + ti * nd + (ki - nq)) : (ki < nr ? ti * nr + ki : H * nr + ti * nd + (ki - nr)));
            p1 = (reduced && ki >= nq) ? gv + ti * nd + (ki - nq) : nullptr;
            p2 = nullptr;
            mode = p1 ? BandRows::DIFF : BandRows::PLAIN;
        }
        P.mode = mode;
        if (p0) P.x0 = *p0;
        if (p1) P.x1 = *p1;
        if (p2) P.x2 = *p2;
    };
    // entry e of row i = ti s + ki; (tj, kj): its column i - w + e likewise (unused for the right-hand side, e = w + 1).  Loads only.
    auto entry_fetch4 = [&](int i, int e, int ti, int ki, int tj, int kj) -> Pending {
        Pending P{0.0, 0.0, 0.0, 0.0, BandRows::CONST};
        if (i >= NR) return P;
        const double *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;
        int mode = BandRows::CONST;
        if (e <= w && i - w + e >= 0) mode = row.terms4(ti, ki, tj, kj, p0, p1, p2, P.c);
        entry_issue(mode, e, ti, ki, p0, p1, p2, P);
        return P;
    };
    // ... of a row past the first w (every column exists), from its tabulated descriptor and the parameter block in LDS
    auto entry_fetch_d = [&](int i, int e, int ti, int ki, unsigned d) -> Pending {
        Pending P{0.0, 0.0, 0.0, 0.0, BandRows::CONST};
        if (i >= NR) return P;
        if constexpr (REV) {      // (ti, ki arrive as the REVERSED step / index of the row: matrix step H-1-ti, index s-1-ki)
            // middle rows (behind the last pivot): their block among themselves and their right-hand side belong to the top chain
            if (i - pad >= nbr && (e > w || i - w + e - pad >= nbr)) return P;
            ti = H - 1 - ti; ki = s - 1 - ki;
        }
        const double *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;
        int mode;
        if (e <= w) {
            mode = d & 7;
            const int arr = (d >> 3) & 7, st = f_stride[arr], o = __mul24(ti, st) + (int)(d >> 8);
            P.c = f_const[(d >> 6) & 3];
            p0 = mode == BandRows::CONST ? nullptr : (const double*)f_base[arr] + o;
            if (mode == BandRows::SUM) {
                const double* Vp = (const double*)f_base[BandRows::AV];
                if (Vp != nullptr) { p1 = Vp + o; if (ti + 1 < H) p2 = p1 + st; }
            }
        } else {                                                 // right-hand side (reduced form: the dual rows carry r_d - du1 R^-1 r_u)
            p0 = (const double*)f_base[5] + (reduced ? (ki < nq ? ti * nr + nu + ki : H * nr + ti * nd + (ki - nq)) : (ki < nr ? ti * nr + ki : H * nr + ti * nd + (ki - nr)));
            p1 = (reduced && ki >= nq) ? (const double*)f_base[6] + (ti * nd + (ki - nq)) : nullptr;
            mode = p1 ? BandRows::DIFF : BandRows::PLAIN;
        }
        P.mode = mode;
        if (p0) P.x0 = *p0;
        if (p1) P.x1 = *p1;
        if (p2) P.x2 = *p2;
        return P;
    };
    // Bottom chain: row i, entry e of the REVERSED matrix A'[i][j] = A[N-1-(i-pad)][N-1-(j-pad)] (pad decoupled dummy rows first), j = i-w+e.
    // The matrix entry is an UPPER one (column c >= row r): by symmetry the lower entry (c, r).  Middle rows: see entry_fetch_d.
    [[maybe_unused]] auto entry_fetch_rev = [&](int i, int e) -> Pending {
        Pending P{0.0, 0.0, 0.0, 0.0, BandRows::CONST};
        if (i >= NR) return P;
        const int j = i - w + e;
        if (e <= w && j < 0) return P;
        if (i < pad || (e <= w && j < pad)) { P.c = (e <= w && i == j) ? 1.0 : 0.0; return P; }      // dummy rows: identity, zero right-hand side
        if (i - pad >= nbr && (e > w || j - pad >= nbr)) return P;
        const int r = N - 1 - (i - pad), tr = r / s, kr = r - tr * s;
        const double *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;
        int mode = BandRows::CONST;
        if (e <= w) {
            const int c = N - 1 - (j - pad), tc = c / s, kc = c - tc * s;
            mode = row.decode(row.desc(tc - tr, kc, kr), tc, p0, p1, p2, P.c);
        }
        entry_issue(mode, e, tr, kr, p0, p1, p2, P);
        return P;
    };
    auto entry_fetch = [&](int i, int e) -> Pending {
        if constexpr (REV) return entry_fetch_rev(i, e);
        const int j = max(i - w + min(e, w), 0), ti = i / s, tj = j / s;
        return entry_fetch4(i, e, ti, i - ti * s, tj, j - tj * s);
    };
    auto entry_done = [&](const Pending& P) { return BandRows::combine(P.mode, P.c, P.x0, P.x1, P.x2); };
    auto entry_value = [&](int i, int e) -> double { return entry_done(entry_fetch(i, e)); };
    auto entry_commit = [&](int i, int e, double v) {
        if (i >= NR) return;
        const int si = i % M;
        if (e <= w) { const int j = i - w + e; if (j >= 0) W[si * MS + j % M] = v; }
        else yw[si] = v;
    };
    for (int idx = tid; idx < s * EW; idx += nt) {               // descriptors of the rows of one step: entry e of row index ki sits in column
        const int ki = idx / EW, e = idx - ki * EW;              // ki - w + e of the row's own step, counted in steps of s downwards
        unsigned d = 0u;
        if (e <= w) {
            if constexpr (REV) {
                // reversed rows: table index ki = reversed in-step index -> matrix index kr = s-1-ki; the entry is the lower entry
                // (c, r) with c = r + (w - e): dt steps ahead.  decode() is handed the ROW's step tr, so the offset takes dt blocks along.
                const int kr = s - 1 - ki, crel = kr + (w - e), dt = crel / s, kc = crel - dt * s;
                d = row.desc(dt, kc, kr);
                const int mode = d & 7, arr = (d >> 3) & 7;
                if (mode != BandRows::CONST) {
                    const int stride = arr <= BandRows::AV ? nq * nq : arr == BandRows::AG ? nd * nd : arr == BandRows::ADZ ? S.nths * nd : nu * nu;
                    d += (unsigned)(dt * stride) << 8;
                }
            } else {
                const int rel = ki - w + e, st = (rel + 3 * s) / s - 3;      // floor(rel / s), rel >= -(3 s - 1)
                d = row.desc(-st, ki, rel - st * s);
            }
        }
        dtab[idx] = d;
    }
    if (tid == 0) {
        f_base[BandRows::AQ] = (unsigned long long)S.Q; f_base[BandRows::AV] = (unsigned long long)S.V;
        f_base[BandRows::AG] = (unsigned long long)(reduced ? Gm : nullptr); f_base[BandRows::ADZ] = (unsigned long long)dzb;
        f_base[BandRows::AR] = (unsigned long long)S.R; f_base[5] = (unsigned long long)rb; f_base[6] = (unsigned long long)gv; f_base[7] = 0ull;
        f_stride[BandRows::AQ] = f_stride[BandRows::AV] = nq * nq; f_stride[BandRows::AG] = nd * nd; f_stride[BandRows::ADZ] = S.nths * nd;
        f_stride[BandRows::AR] = nu * nu; f_stride[5] = f_stride[6] = f_stride[7] = 0;
        f_const[BandRows::C0] = 0.0; f_const[BandRows::CM1] = -1.0; f_const[BandRows::CMRHO] = -rho; f_const[3] = 0.0;
    }
    __threadfence_block();
    __syncthreads();
    {
        const int n0 = min(Mw, NR) * EW;                         // the first Mw rows, all threads
        for (int idx = tid; idx < n0; idx += nt) { const int i = idx / EW, e = idx - i * EW; entry_commit(i, e, entry_value(i, e)); }
    }
    for (int e = tid; e < RB * MS; e += nt) PL[e] = 0.0;
    for (int e = tid; e < 2 * (RB + 1) * RB; e += nt) nL11[e] = 0.0;
    int ent_t[NE], ent_e[NE];                                    // (row of the block, entry) of this thread's entering entries; -1: none
#pragma unroll
    for (int n = 0; n < NE; ++n) {
        const int idx = tid - 128 + n * ((int)nt - 128);
        const bool on = tid >= 128 && idx < RB * EW;
        ent_t[n] = on ? idx / EW : -1;
        ent_e[n] = on ? idx - (idx / EW) * EW : 0;
    }
    Pending nxt[NE];
    int eti[NE], eki[NE];                                        // (step, index in the step) of the row of the NEXT fetch, advanced RB per block
    unsigned dsc[NE];                                            // ... and its descriptor, loaded a block ahead
#pragma unroll
    for (int n = 0; n < NE; ++n) {                               // rows entering after the first block
        const int i = ent_t[n] >= 0 ? Mw + ent_t[n] : NR, j = max(i - w + min(ent_e[n], w), 0);
        const int irel = REV ? max(i - pad, 0) : i;              // (bottom chain: reversed step / index, counted behind the dummy rows)
        eti[n] = irel / s; eki[n] = irel - eti[n] * s;
        const int tj = j / s;
        if constexpr (REV) nxt[n] = entry_fetch_rev(i, ent_e[n]);
        else nxt[n] = entry_fetch4(i, ent_e[n], eti[n], eki[n], tj, j - tj * s);
        eki[n] += RB; while (eki[n] >= s) { eki[n] -= s; ++eti[n]; }
        dsc[n] = ent_t[n] >= 0 ? dtab[eki[n] * EW + ent_e[n]] : 0u;
    }
    __syncthreads();
    const int tx = tid & 63, ty = tid >> 6, nty = nt >> 6;
    auto rl = [](double v, int lane) {                           // v of another lane of the wavefront (lane: uniform)
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
        return __hiloint2double(hi, lo);
    };
    auto wrap = [&](int x) {                                     // slot of an index < 3 M past a slot
        if constexpr (POW2) return x & (M - 1);
        else { if (x >= M) x -= M; if (x >= M) x -= M; return x; }
    };
    // A block of RB pivots (round 4; DESIGN.md 5.2c has the path here with the phase clocks of every step):
    //   P2   all sixteen wavefronts, RB lanes per row below the block (lane t owns the row's entry in pivot column t): the triangular solve
    //        against the diagonal block runs ACROSS the lanes - per pivot u two multiplies and one fused multiply-add whose DPP source is the
    //        row broadcast of l_u d_u; the lanes store their multipliers into LDS (PL) and into the row of L in global memory;
    //   barrier
    //   chain (wavefront 0):  tile (0, 0) of the update - the next diagonal block -, then P1 of the NEXT block on lanes 0 .. RB-1: pivots,
    //        reciprocals, the multipliers among the pivot rows, the forward substitution among them (lane r owns pivot row r, right-looking,
    //        hand-overs by v_readlane, no branches, no global stores) - the only sequential part of the factorisation;
    //   bulk (wavefronts 1 .. 15), beside the chain:  right-hand side of the rows below, the other tiles of the rank-RB update of the lower
    //        triangle (16 x 16 tiles on v_mfma_f64_16x16x4: A = -(l d) of the tile's rows, B = l of its columns, K = RB), the entering rows
    //        committed, the next ones fetched;
    //   barrier
    // pivots / reciprocals / diagonal block are double-buffered by block parity (the bulk of block k reads d of block k while the chain writes
    // the next).  Every entry receives the same operations in the same order as in the one-pivot-at-a-time form of round 3: the factors are
    // bit-identical to it (scripts/dbg/banded_cmp.py).  Cycles per block of 8 pivots (w = 107), as the forms came: one-wave panel of all rows
    // 42 k; P1 / P2 / P3 split 27 k; MFMA tiles, DPP solve, slots by counting 13.4 k; look-ahead in three phases 10.8 k; chain beside bulk,
    // compile-time window, stores from P2 8.3 k - of which the P2 phase 2.1 k, the chain 5.0 k beside an issue-bound bulk of 6.2 k.
    int sk = 0, pb = 0, mt_tab = -1;                             // k % M by counting, parity of the block, the window size the tile table is for
    const int wv = __builtin_amdgcn_readfirstlane(ty);
    // (the chain brings the right-hand side of the next block's eight pivot rows up to date in registers: `pending`)
    auto m24 = [](int a_, int b_) { return __mul24(a_, b_); };  // (v_mul_lo_u32 is a quarter-rate instruction; every product here fits 24 x 24 bits)
    auto P1 = [&](int kk, int skk, int par, bool pending, const double* ypv_cur) {      // lanes 0 .. RB-1 of wavefront 0
        const int nbk = min(RB, NP - kk);
        double* dvp = dv + par * RB; double* dinvp = dinv + par * RB; double* ypvp = ypv + par * RB; double* L11p = L11 + par * RB * RB;
        double* nL11p = nL11 + par * (RB + 1) * RB;
        const int r = tid;
        const bool vr = r < nbk;
        const int sr = wrap(skk + r);
        double a[RB];
        double yr = vr ? yw[sr] : 0.0;
        if (pending) {                                           // the block before this one still owes these rows its right-hand-side update
#pragma unroll
            for (int t = 0; t < RB; ++t) yr = fma(-PL[m24(t, MS) + sr], ypv_cur[t], yr);
        }
#pragma unroll
        for (int t = 0; t < RB; ++t) a[t] = (vr && t <= r && t < nbk) ? W[m24(sr, MS) + wrap(skk + t)] : 0.0;
#pragma unroll
        for (int t = 0; t < RB; ++t) {
            if (t < nbk) {
                const double d = rl(a[t], t), inv = 1.0 / d, yp = rl(yr, t);
                const double lv = r > t ? a[t] * inv : 0.0;      // (rows past the end carry zeros)
                yr = fma(-lv, yp, yr);                           // forward substitution rides along
                const double c = lv * d;
#pragma unroll
                for (int t2 = t + 1; t2 < RB; ++t2) a[t2] = fma(-c, rl(lv, t2), a[t2]);      // right-looking inside the block
                L11p[r * RB + t] = lv; nL11p[r * RB + t] = -lv;  // (zero on and above the diagonal)
                dvp[t] = d; dinvp[t] = inv; ypvp[t] = yp;        // (the same value from every lane)
            }
        }
    };
    if (tid < RB) P1(0, 0, 0, false, ypv);
    lds_barrier();
    BPROF(0)
    BTWSTAMP(1)
    const int tb = tid - 64;             // the bulk threads: wavefronts 1 .. 15
    for (int k = 0; k < NP; k += RB) {
        if constexpr (TW == 1) {
            // The trace of the bottom chain: what the rows it eliminated contribute to rows m2 .. m2 + w - 1.  All of them are in the
            // window now and none has been a pivot (the next diagonal block - pivots m2 .. - is formed later in this iteration).
            if (k == m2 - RB) {
                BTWSTAMP(2)
                tw_ok = kkt_tw_wait(xfl + 0, S.kkt_tw_epoch, S.kkt_tw_spins); if (!tw_ok) kkt_tw_give_up(xfl, S.kkt_tw_epoch, tid == 0);
                BTWSTAMP(3)
                for (int e = tid; e < w * (w + 1); e += nt) {
                    const int a = e / (w + 1), a2 = e - a * (w + 1);
                    if (a2 <= a || a2 == w) {
                        const double v = tw_ok ? xtr[e] : __builtin_nan("");
                        if (a2 == w) yw[(m2 + a) & (M - 1)] += v;
                        else W[m24(((m2 + a) & (M - 1)), MS) + ((m2 + a2) & (M - 1))] += v;
                    }
                }
                lds_barrier();
            }
        }
        const int nb_ = min(RB, NP - k);                         // pivots of this block
        const double* dvp = dv + pb * RB; const double* dinvp = dinv + pb * RB; const double* ypvp = ypv + pb * RB; const double* L11p = L11 + pb * RB * RB;
        const int base = k + nb_, mt = max(min(w, NR - base), 0); // mt rows / columns of the trailing window present
        const int nT = (mt + 15) >> 4, ntiles = nT * (nT + 1) / 2;
        if (mt != mt_tab) {                                      // tile table (steady state: written once): tile (0, 0) first
            if (tid < ntiles) {
                int I = tid, J = 0;
                if (tid >= nT) {
                    const int t2 = tid - nT;
                    I = 0; while ((I + 1) * (I + 2) / 2 <= t2) ++I;
                    J = t2 - I * (I + 1) / 2 + 1; I += 1;
                }
                tile_tab[tid] = I | (J << 8);
            }
            mt_tab = mt;
        }
        {                                                        // ---- P2: row k + q, q = RB .. RB + w - 1, entry t
            constexpr int NG = 16 / RB;                          // rows per DPP row of 16 lanes
            const int l16 = tx & 15, g = l16 / RB, t = l16 - g * RB;
            const int q = RB + (tid >> 4) * NG + g, i = k + q;
            const bool on = q < RB + w && i < NR && t < nb_;
            const int sr = wrap(sk + q);
            double a = (on && q - t <= w) ? W[m24(sr, MS) + wrap(sk + t)] : 0.0;
            const double inv = dinvp[t], d = dvp[t];
            double lm[NG][RB];                                   // minus the pivot row t's multipliers for the earlier pivots (zero from the
            const double* nL11p = nL11 + pb * (RB + 1) * RB;     // diagonal on) in the column of this lane's row of the DPP row, zero in the others
#pragma unroll                                                   // (reading a row of zeros instead of selecting doubled the LDS reads: P2 600 k cycles slower)
            for (int u = 0; u < RB; ++u) {
                const double v = nL11p[t * RB + u];
#pragma unroll
                for (int g2 = 0; g2 < NG; ++g2) lm[g2][u] = g == g2 ? v : 0.0;
            }
            // a -= (l_u d_u) L11[t][u]: the product of lane u of the row rides on the DPP source of the multiply-add; one multiply-add per
            // row of the DPP row, all but this lane's own with a zero factor (exact)
            static_for<0, RB - 1>([&](auto U) {
                constexpr int u = decltype(U)::value;
                const double c = (a * inv) * d;                  // final in lane u of the row
                if constexpr (NG == 2) fmac_bcast2<u, RB + u>(a, c, lm[0][u], lm[1][u]);
                else { fmac_bcast2<u, RB + u>(a, c, lm[0][u], lm[1][u]); fmac_bcast2<2 * RB + u, 3 * RB + u>(a, c, lm[2][u], lm[3][u]); }
            });
            const double lv = a * inv;                           // zero where the pivot does not couple
            if (on) PL[m24(t, MS) + sr] = lv;
            // ... and into row i of L in global memory: the RB lanes of a row write one contiguous piece (the rows of the diagonal block
            // follow from the bulk).  Nothing waits for these stores: the bulk commits its entering rows - the only wait for memory in
            // the loop - after its tiles.
            if (on && q - t <= w) Lr[m24(i, LW) + (w - (q - t))] = lv;
        }
        BPROF(1)
        lds_barrier();              // the block's multipliers are in place; the pivots' slots may be overwritten now
        BPROF(2)
        // ---- rank-RB update of the lower triangle of the trailing window: 16 x 16 tiles on v_mfma_f64_16x16x4 ------------------------
        const int sb = wrap(sk + nb_);
        const int li = tx & 15, lk = tx >> 4;
        auto tile = [&](int tl) {
            const int ij = __builtin_amdgcn_readfirstlane(tile_tab[tl]), I = ij & 255, J = ij >> 8;
            const int ra = 16 * I + li, cb_ = 16 * J + li;
            const int sra = wrap(sb + min(ra, mt - 1)), scb = wrap(sb + min(cb_, mt - 1));
            // no predicates: addresses are clamped into the window, what a lane computes for a row / column past the end (or above the
            // diagonal - nothing reads those cells) lands in the dummy row
            d4 acc;
            double* cell[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int rw = 16 * I + lk + 4 * r4;
                const double* src = W + m24(wrap(sb + min(rw, mt - 1)), MS) + scb;
                acc[r4] = *src;
                cell[r4] = (rw < mt && cb_ < mt) ? (double*)src : W + m24(M, MS) + li;
            }
#pragma unroll
            for (int kb = 0; kb < RB / 4; ++kb) {
                const int t = 4 * kb + lk;
                const double av = -(PL[m24(t, MS) + sra] * dvp[t]);
                const double bv = PL[m24(t, MS) + scb];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) *cell[r4] = acc[r4];
        };
        const int skn = wrap(sk + RB);
        if (wv == 0) {              // ---- the chain: the next diagonal block's tile, then the next diagonal block
            if (mt > 0) tile(0);
            BPROF(3)
            if (k + RB < NP && tid < RB) P1(k + RB, skn, pb ^ 1, true, ypvp);
            BPROF(4)
        } else {                    // ---- the bulk
            // right-hand side of the rows below the block - but for the next block's pivot rows: wavefront 0 keeps those to itself
            if (tb >= RB && tb < w && k + RB + tb < NR) {
                const int sr = wrap(sk + RB + tb);
                double yr = yw[sr];
#pragma unroll
                for (int t = 0; t < RB; ++t) yr = fma(-PL[m24(t, MS) + sr], ypvp[t], yr);      // (rows below exist: the block is complete, nb_ = RB)
                yw[sr] = yr;
            }
            if (tb < (RB - 1) * RB) {                            // the diagonal block's multipliers -> rows of L (P2 stored those of the rows below)
                const int q = 1 + tb / RB, t = tb - (tb / RB) * RB;
                if (k + q < NP && t < nb_ && t < q) Lr[m24(k + q, LW) + (w - (q - t))] = L11p[q * RB + t];
            }
            if (tb >= 64 && tb < 64 + nb_) { yg[k + tb - 64] = ypvp[tb - 64]; Lr[m24(k + tb - 64, LW) + w] = dinvp[tb - 64]; }
            BPROF(3)
            // tiles 1 .. : the wavefronts that share wavefront 0's SIMD (4, 8, 12: wave v runs on SIMD v % 4) come last in the round-robin -
            // with 27 tiles on 15 wavefronts they take one tile each, the others two
            const int rank = (wv & 3) != 0 ? (wv >> 2) * 3 + (wv & 3) - 1 : (nty - nty / 4) + (wv >> 2) - 1;
            for (int tl = 1 + rank; tl < ntiles; tl += nty - 1) tile(tl);
            BPROF(4)
            // entering rows: their loads were issued a block ago; by now the stores of P2 that stand behind them in the memory queue are done
#pragma unroll
            for (int n = 0; n < NE; ++n) {
                if (n > 0 && !need2) break;
                if (ent_t[n] >= 0 && ent_t[n] < nb_ && k + Mw + ent_t[n] < NR) {     // row k + Mw + t (Mw = M: the slot of pivot k + t)
                    const int si = wrap(sk + Mw + ent_t[n]);
                    const double v = entry_done(nxt[n]);
                    if (ent_e[n] <= w) W[m24(si, MS) + wrap(sk + RB + ent_t[n] + ent_e[n])] = v;       // column k + RB + t + e (>= 0)
                    else yw[si] = v;
                }
            }
            // rows entering after the NEXT block: the loads are issued here and land during the next block's P2
#pragma unroll
            for (int n = 0; n < NE; ++n) {
                if (n > 0 && !need2) break;
                if (ent_t[n] >= 0) {
                    nxt[n] = entry_fetch_d(k + RB + Mw + ent_t[n], ent_e[n], eti[n], eki[n], dsc[n]);
                    eki[n] += RB; while (eki[n] >= s) { eki[n] -= s; ++eti[n]; }
                    dsc[n] = dtab[eki[n] * EW + ent_e[n]];
                }
            }
        }
        BPROF(5)
        lds_barrier();
        BPROF(6)
        sk = skn; pb ^= 1;
    }
    BTWSTAMP(4)
    if constexpr (TW == 2) {
        // The trace.  The rows behind the last pivot entered the window with a zero middle block and right-hand side, so what stands
        // there now is exactly what the eliminated rows contribute; the first RB of them still owe the last block's right-hand-side
        // update (the chain keeps the next block's pivot rows to itself - there is no next block).  Written in MATRIX order: trace row
        // a = matrix row m2 + a = reversed row NP + w - 1 - a; the matrix's lower entry (a, a2 <= a) is the window's (ia2, ia), ia2 >= ia.
        const double* ypl = ypv + (pb ^ 1) * RB;
        if (tid < RB && NP + tid < NR) {
            const int sr = (NP + tid) & (M - 1);
            double yr = yw[sr];
#pragma unroll
            for (int t = 0; t < RB; ++t) yr = fma(-PL[m24(t, MS) + sr], ypl[t], yr);
            yw[sr] = yr;
        }
        lds_barrier();
        for (int e = tid; e < w * (w + 1); e += nt) {
            const int a = e / (w + 1), a2 = e - a * (w + 1), ia = NP + w - 1 - a;
            if (a2 == w) xtr[e] = yw[ia & (M - 1)];
            else if (a2 <= a) xtr[e] = W[m24(((NP + w - 1 - a2) & (M - 1)), MS) + (ia & (M - 1))];
        }
        __threadfence();
        __syncthreads();
        if (tid == 0) astore(xfl + 0, S.kkt_tw_epoch);
    }
    __syncthreads();              // full barrier: the rows of L and y in global memory are read back below
    // ---- back substitution  L^T x = D^-1 y.  acc_j = sum_{i > j} L[i][j] x_i is built row by row (i descending); round 4: the pending
    //      acc_j live in REGISTERS of wavefront 0 (column j belongs to lane (j + 192) % 64, register ((j + 192) / 64) % 3 - a band of
    //      w < 192 columns never holds two columns of one lane in one register) and x_i needs one v_readlane.  The other fifteen
    //      wavefronts stage the rows of L one chunk ahead into LDS ALREADY IN THAT REGISTER LAYOUT (192 values per row, zero where the
    //      lane's column is outside the band, then 1/d_i and y_i): a row costs wavefront 0 five LDS reads, the hand-over and four
    //      multiply-adds, nothing else.  x is collected lane-wise (row i in lane (i + 192) % 64) and written once per 64 rows.
    //      History: 3.6 k cycles per row (LDS read-modify-write chain on acc, global-load latency behind a 16-row register ring), 640 with
    //      the sums in registers but the band bookkeeping on wavefront 0 (banded_prof_mfma.log).
    double* D = K.delta + (size_t)b * S.N;
    {
        constexpr int RS = 194;                                  // staged row: [3][64] values, 1/d, y
        const int avail = MS * MS + MS + RB * MS + 2 * (3 * RB + RB * RB + (RB + 1) * RB) + 64;      // doubles of LDS the window held
        int CR = avail / (2 * RS); if (CR > 64) CR = 64;
        const int nch = (NR + CR - 1) / CR, CS = CR * RS;
        // matrix index of this chain's row i in the reference layout (bottom chain: reversed rows behind the dummies)
        auto out_index = [&](int i) {
            const int r = REV ? N - 1 - (i - pad) : i, ti = r / s, ki = r - ti * s;
            if (reduced) return ki < nq ? ti * nr + nu + ki : H * nr + ti * nd + (ki - nq);
            return ki < nr ? ti * nr + ki : H * nr + ti * nd + (ki - nr);
        };
        BTWSTAMP(5)
        if constexpr (TW == 2) { tw_ok = kkt_tw_wait(xfl + 1, S.kkt_tw_epoch, S.kkt_tw_spins); if (!tw_ok) kkt_tw_give_up(xfl, S.kkt_tw_epoch, tid == 0); }     // the middle rows' solution (the top chain's first w values) is in D
        BTWSTAMP(6)
        auto stage = [&](int ch, int t0, int nth) {              // chunk ch = rows NR-1 - ch CR downwards -> buffer ch & 1  (threads t0 .. of nth)
            const int i_hi = NR - 1 - ch * CR, i_lo = max(i_hi - CR + 1, 0), n = i_hi - i_lo + 1, tot = n * 192;
            double* bufp = sm + (ch & 1) * CS;
            for (int e0 = t0; e0 < tot; e0 += 8 * nth) {
                double tmp[8];
                int dst[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int e = e0 + u * nth;
                    tmp[u] = 0.0; dst[u] = -1;
                    if (e < tot) {
                        const int rw = e / 192, rem = e - rw * 192, sl = rem >> 6, ln = rem & 63, i = i_hi - rw;
                        const int bb = i - w + 192, blk = bb >> 6, bm = blk % 3;
                        int dB = sl - bm; if (dB < 0) dB += 3;
                        int jj = 64 * (blk + dB) + ln; if (jj < bb) jj += 192;
                        dst[u] = rw * RS + rem;
                        // (bottom chain, middle rows i >= NP: only their multipliers w.r.t. the pivots are part of the factor)
                        if (jj >= 192 && jj < i + 192 && (!REV || i < NP || jj - 192 < NP)) tmp[u] = Lr[i * LW + (jj - bb)];
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (dst[u] >= 0) bufp[dst[u]] = tmp[u];
            }
            for (int e = t0; e < n; e += nth) {
                const int i = i_hi - e;
                if (REV && i >= NP) {      // given: x of matrix row N-1-(i-pad), computed by the top chain (1/d = 1, y = x, nothing pending on its column)
                    bufp[e * RS + 192] = 1.0; bufp[e * RS + 193] = tw_ok ? D[out_index(i)] : __builtin_nan("");
                } else { bufp[e * RS + 192] = Lr[i * LW + w]; bufp[e * RS + 193] = yg[i]; }
            }
        };
        stage(0, tid, nt);
        __syncthreads();
        double acc[3] = {0.0, 0.0, 0.0};
        for (int ch = 0; ch < nch; ++ch) {
            if (tid >= 64) { if (ch + 1 < nch) stage(ch + 1, tid - 64, (int)nt - 64); }
            else {
                const int i_hi = NR - 1 - ch * CR, i_lo = max(i_hi - CR + 1, 0);
                const double* bufp = sm + (ch & 1) * CS;
                int i = i_hi;
                while (i >= i_lo) {                              // runs of rows whose own column sits in the same register
                    const int seg_lo = max(i_lo, ((i + 192) & ~63) - 192), so = ((i + 192) >> 6) % 3;
                    auto run = [&](auto SO) {
                        constexpr int so_ = decltype(SO)::value;
                        const double* rp = bufp + (i_hi - i) * RS;
                        double L0 = rp[tx], L1 = rp[64 + tx], L2 = rp[128 + tx], di = rp[192], yi = rp[193];      // di = 1 / d_i
                        double xs = 0.0;
                        for (int r = i; r >= seg_lo; --r) {
                            const double l0 = L0, l1 = L1, l2 = L2, d_ = di, y_ = yi;
                            if (r > seg_lo) { rp += RS; L0 = rp[tx]; L1 = rp[64 + tx]; L2 = rp[128 + tx]; di = rp[192]; yi = rp[193]; }   // next row in flight
                            const int ln = (r + 192) & 63;
                            const double xi = y_ * d_ - rl(acc[so_], ln);
                            const bool own = tx == ln;
                            xs = own ? xi : xs;
                            acc[so_] = own ? 0.0 : acc[so_];
                            acc[0] = fma(l0, xi, acc[0]); acc[1] = fma(l1, xi, acc[1]); acc[2] = fma(l2, xi, acc[2]);
                        }
                        const int lnA = (i + 192) & 63, il = i - (lnA - tx);      // the row whose x this lane collected
                        if (tx <= lnA && il >= seg_lo && (!REV || (il >= pad && il < NP))) D[out_index(il)] = xs;      // (not the dummy rows, not the given ones)
                    };
                    if (so == 0) run(std::integral_constant<int, 0>{});
                    else if (so == 1) run(std::integral_constant<int, 1>{});
                    else run(std::integral_constant<int, 2>{});
                    i = seg_lo - 1;
                }
            }
            __syncthreads();
            if constexpr (TW == 1) {      // the middle rows (this chain's last w) are solved: the bottom chain may start its back substitution
                const int i_lo_done = max(NR - 1 - ch * CR - CR + 1, 0);
                if (i_lo_done <= m2 && i_lo_done + CR > m2 && tid < 64) {
                    __threadfence();
                    if (tid == 0) astore(xfl + 1, S.kkt_tw_epoch);
                }
            }
        }
    }
    BPROF(8)
    BTWSTAMP(7)
    if constexpr (TW != 0) {
        // The chain that finishes last has every Dnu in front of it: it recovers the controls and starts the line search.
        int* const tw_prev = (int*)sm;      // (the staging buffers of the back substitution are free now)
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            *tw_prev = atomicAdd(xfl + 2, 1);      // (epoch-valued flags: nothing to reset)
        }
        __syncthreads();
        if (*tw_prev == 0) return;
        __syncthreads();
        __threadfence();
        if (tid == 0) astore(xfl + 2, 0);
        if (aload(xfl + 3) == S.kkt_tw_epoch) {      // a hand-over of this solve timed out (poisoned numbers): queue the KKT stage again
            if (tid == 0) kkt_tw_requeue(S, b, K.finish);
            return;
        }
    }
    if (reduced) {      // Du_t = R_t^-1 (r_u,t - du1_t^T Dnu_t)
        __threadfence_block();
        double* tu = sm;                                         // [H][nu]
        for (int e = tid; e < H * nu; e += nt) {
            const int t = e / nu, m = e - t * nu;
            const double* du = dzb + ((size_t)t * S.nths + 2 * nq + m) * nd;
            const double* dn = D + H * nr + t * nd;
            double a = rb[t * nr + m];
            for (int k = 0; k < nd; ++k) a = fma(-du[k], dn[k], a);
            tu[e] = a;
        }
        __syncthreads();
        for (int e = tid; e < H * nu; e += nt) {
            const int t = e / nu, m = e - t * nu;
            const double* Ri = S.Rinv + (size_t)t * nu * nu + m;      // row m of the column-major inverse (R need not be symmetric)
            double a = 0.0;
            for (int k = 0; k < nu; ++k) a = fma(Ri[(size_t)k * nu], tu[t * nu + k], a);
            D[t * nr + m] = a;
        }
        __syncthreads();
    }
    BPROF(9)
#ifdef CIMPC_BANDED_PROF
    if (b == 0 && (tid & 63) == 0) for (int j = 0; j < 10; ++j) ((long long*)S.stats)[8 + (tid >> 6) * 10 + j] = bp[j];      // (needs B >= 42)
#endif
    if (K.finish) {
        __threadfence_block();
        start_line_search<BlockSync>(S, b, 1, tid, nt);
    }
}

template <int RB, int SLOTS>
__global__ __launch_bounds__(CIMPC_BANDED_THREADS) void kkt_banded_kernel(NewtonDev S, KktArgs K, double* ws_all) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    kkt_banded_body<RB, SLOTS, 0>(S, K, ws_all, sm, (int)blockIdx.x + S.b0);
}

// Twisted launch (round 5): two workgroups per rollout - block 2k the bottom chain (it waits for nothing until its trace is out),
// block 2k + 1 the top chain.  The chains are real functions with their own register allocation, like the condensed solve's
// (newton_impl.h: kkt_tw_chain).
struct KktBandTwArgs { NewtonDev S; KktArgs K; double* ws; };
template <int RB, int SLOTS, int TW>
static __device__ __attribute__((noinline)) void kkt_banded_chain(unsigned long long ka, int b, lds_double_ptr sm3) {
    const NewtonDev S = kkt_kernarg<NewtonDev, offsetof(KktBandTwArgs, S)>(ka);
    const KktArgs K = kkt_kernarg<KktArgs, offsetof(KktBandTwArgs, K)>(ka);
    double* const ws = kkt_kernarg<double*, offsetof(KktBandTwArgs, ws)>(ka);
    kkt_banded_body<RB, SLOTS, TW>(S, K, ws, (double*)sm3, b);
}
template <int RB, int SLOTS>
__global__ __launch_bounds__(CIMPC_BANDED_THREADS) void kkt_banded_twisted_kernel(KktBandTwArgs A) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int b = ((int)blockIdx.x >> 1) + A.S.b0;
    const unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    if (((int)blockIdx.x & 1) == 0) kkt_banded_chain<RB, SLOTS, 2>(ka, b, (lds_double_ptr)sm);
    else kkt_banded_chain<RB, SLOTS, 1>(ka, b, (lds_double_ptr)sm);
}

static size_t banded_lds_bytes_slots(int slots, int rb) {      // window (slots + dummy)^2, right-hand side, rb multiplier rows, 2 x (pivots / reciprocals / rhs, diagonal block), tile table
    const size_t MS = (size_t)slots + 1;
    return (MS * MS + MS + (size_t)rb * MS + 2 * (3 * (size_t)rb + (size_t)rb * rb + ((size_t)rb + 1) * rb) + 64) * sizeof(double);
}
static size_t banded_lds_bytes(int w, int rb = 4, bool pow2 = false) { return banded_lds_bytes_slots(pow2 ? banded_pow2_slots(w + rb) : w + rb, rb); }
static int band_halfwidth(const NewtonDev& S) {      // (the kernel's own formula: reduced form when the controls are eliminated)
    if (S.band_reduce != 0 && S.dm.nu > 0) { const int s = S.dm.nq + S.nd; return std::min(3 * s - 1, S.dm.H * s - 1); }
    const int s = S.nr + S.nd;
    return std::min(3 * s - 1 - S.dm.nu, S.N - 1);
}
static int banded_rb(const NewtonDev& S) {      // pivots per window update: 8 where that window fits
    // (P2 has 16 / RB rows per 16 lanes of 1024 threads: 128 rows below a block of eight - the LDS bound (w <= 126) is the tighter one)
    const int w = band_halfwidth(S);
    return ((S.band_form & 1) == 0 && w <= 128 && banded_lds_bytes(w, 8) <= 156 * 1024) ? 8 : 4;
}
static bool banded_pow2(const NewtonDev& S) {      // power-of-two slot count where THAT window fits
    return (S.band_form & 2) == 0 && banded_lds_bytes(band_halfwidth(S), banded_rb(S), true) <= 156 * 1024;
}
static size_t banded_lds_bytes(const NewtonDev& S) {      // ... and, before the window is in use, T = du1 R^-1 of every step ([H][nd x nu])
    const size_t win = banded_lds_bytes(band_halfwidth(S), banded_rb(S), banded_pow2(S));
    const size_t pre = (S.band_reduce != 0 && S.dm.nu > 0) ? (size_t)S.dm.H * S.nd * S.dm.nu * sizeof(double) : 0;
    return std::max(win, pre);
}
bool kkt_banded_available(const NewtonDev& S) {      // window + two vectors in 160 KB of LDS, back substitution: w < 192
    const int w = band_halfwidth(S);
    return S.dm.mode == CIMPC_MODE_CONFIGURATION && w <= 190 && banded_lds_bytes(S) <= 160 * 1024;
}

// the twisted form: asked for (NewtonDev::kkt_tw_band, set by the host where the launch is latency-bound), reduced form, blocks of
// eight, compile-time power-of-two window, and a band long enough for two chains
bool kkt_banded_twisted_available(const NewtonDev& S) {
    if (S.kkt_tw_band == 0 || S.kkt_tw_flags == nullptr || S.band_reduce == 0 || S.dm.nu <= 0) return false;
    if (banded_rb(S) != 8 || !banded_pow2(S)) return false;
    const int w = band_halfwidth(S), slots = banded_pow2_slots(w + 8), N = S.dm.H * (S.dm.nq + S.nd);
    if (slots != 128 && slots != 64 && slots != 32) return false;
    if (N < 6 * (w + 8)) return false;
    const BandSplit sp = banded_twisted_split(N, w, 8);
    return sp.m2 >= 16 && sp.m2 % 8 == 0 && sp.Nb % 8 == 0 && sp.Nb - sp.pad >= 8;
}

size_t kkt_dense_workspace_doubles(const NewtonDev& S, bool banded) {
    if (banded) return (size_t)S.dm.B * banded_ws_per_rollout(S);
    return (size_t)S.dm.B * ((size_t)S.N * S.N + 2 * (size_t)S.N);
}

static int launch_kkt_dense(const NewtonDev& S, const KktArgs& K, double* ws, hipStream_t s, bool banded) {
    if (banded) {
        const size_t lds = banded_lds_bytes(S);
        static LdsOptIn optin[7];      // one per kernel
        auto go = [&](auto kern, int which) {
            if (lds_opt_in(optin[which], (const void*)kern, lds) != CIMPC_OK) return (int)CIMPC_ERR_HIP;
            hipLaunchKernelGGL(kern, dim3(S.nb_launch), dim3(CIMPC_BANDED_THREADS), lds, s, S, K, ws);
            return hipGetLastError() == hipSuccess ? (int)CIMPC_OK : (int)CIMPC_ERR_HIP;
        };
        const bool p2 = banded_pow2(S);
        const int slots = p2 ? banded_pow2_slots(band_halfwidth(S) + banded_rb(S)) : 0;
        if (kkt_banded_twisted_available(S)) {      // two chains per rollout from the two ends of the band (latency-bound launches)
            static LdsOptIn optin_tw[3];
            auto go_tw = [&](auto kern, int which) {
                if (lds_opt_in(optin_tw[which], (const void*)kern, lds) != CIMPC_OK) return (int)CIMPC_ERR_HIP;
                hipLaunchKernelGGL(kern, dim3(2 * S.nb_launch), dim3(CIMPC_BANDED_THREADS), lds, s, KktBandTwArgs{S, K, ws});
                return hipGetLastError() == hipSuccess ? (int)CIMPC_OK : (int)CIMPC_ERR_HIP;
            };
            if (slots == 128) return go_tw(kkt_banded_twisted_kernel<8, 128>, 0);
            if (slots == 64) return go_tw(kkt_banded_twisted_kernel<8, 64>, 1);
            if (slots == 32) return go_tw(kkt_banded_twisted_kernel<8, 32>, 2);
        }
        if (banded_rb(S) == 8) {
            if (slots == 128) return go(kkt_banded_kernel<8, 128>, 4);
            if (slots == 64) return go(kkt_banded_kernel<8, 64>, 5);
            if (slots == 32) return go(kkt_banded_kernel<8, 32>, 6);
            return p2 ? go(kkt_banded_kernel<8, -1>, 0) : go(kkt_banded_kernel<8, 0>, 1);
        }
        return p2 ? go(kkt_banded_kernel<4, -1>, 2) : go(kkt_banded_kernel<4, 0>, 3);
    }
    hipLaunchKernelGGL(kkt_dense_kernel, dim3(S.nb_launch), dim3(256), 0, s, S, K, ws);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}
int launch_kkt_dense_args(const NewtonDev& S, const KktArgs& K, double* ws, hipStream_t s, bool banded) {
    return launch_kkt_dense(S, K, ws, s, banded);
}
int launch_kkt_dense_newton(const NewtonDev& S, double* ws, hipStream_t s, bool banded) {
    KktArgs K{S.res, S.delta, S.beta, 0.0, S.stage, 1};
    return launch_kkt_dense(S, K, ws, s, banded);
}
int launch_kkt_dense_raw(const NewtonDev& S, const double* r_dev, double beta, double* delta_dev, double* ws, hipStream_t s, bool banded) {
    KktArgs K{r_dev, delta_dev, nullptr, beta, nullptr, 0};
    return launch_kkt_dense(S, K, ws, s, banded);
}

// ---------------------------------------------------------------------------------------------------------
// Stand-alone B1 seam: `linear_solve!(solver, x, A::SparseMatrixCSC, b)` with the matrix PASSED IN
// (/root/reference/src/controller/newton.jl:86,218; contract src/solver/lu.jl:4-12, src/solver/ldl.jl:144-149: the
// solver factorizes the CSC matrix it is handed and overwrites x).  The reference default `lu_solver` densifies
// (`Array(A)`) and runs LU with partial pivoting; so does this: scatter on the device, the LU of the dense backend
// above, one workgroup.  It exists so that the Julia host can swap `opts.solver` alone; it is not a fast path - the
// fast path is cimpc_kkt_solve on a handle whose sensitivities are device-resident (no matrix crosses the boundary).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void csc_dense_solve_kernel(int n, const long long* colptr, const long long* rowval, const double* nzval,
                                                              const double* bvec, double* A, double* x, int* piv, int* bad) {
    const int tid = threadIdx.x, nt = blockDim.x;
    __shared__ double s_val[256];
    __shared__ int s_idx[256];
    for (size_t e = tid; e < (size_t)n * n; e += nt) A[e] = 0.0;
    for (int i = tid; i < n; i += nt) x[i] = bvec[i];
    __syncthreads();
    for (int c = tid; c < n; c += nt)                       // 1-based CSC as Julia stores it (duplicate entries add up)
        for (long long p = colptr[c] - 1; p < colptr[c + 1] - 1; ++p) {
            const long long r = rowval[p] - 1;
            if (r < 0 || r >= n) { bad[0] = 1; continue; }
            A[(size_t)c * n + r] += nzval[p];
        }
    __syncthreads();
    dense_lu_solve(A, x, piv, n, tid, nt, s_val, s_idx, bad + 1);
}

}  // namespace cimpc

extern "C" int cimpc_linear_solve_csc(int device, int n, const long long* colptr, const long long* rowval, const double* nzval,
                                      const double* b, double* x) {
    using namespace cimpc;
    if (n <= 0 || !colptr || !rowval || !nzval || !b || !x) return CIMPC_ERR_INVALID;
    if (colptr[0] != 1) return CIMPC_ERR_INVALID;           // 1-based column pointers (SparseMatrixCSC)
    const long long nnz = colptr[n] - 1;
    if (nnz < 0 || (size_t)n * n > ((size_t)1 << 31)) return CIMPC_ERR_INVALID;
    for (int c = 0; c < n; ++c) if (colptr[c + 1] < colptr[c]) return CIMPC_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CIMPC_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return CIMPC_ERR_NO_DEVICE;
    long long *d_cp = nullptr, *d_rv = nullptr;
    double *d_nz = nullptr, *d_b = nullptr, *d_A = nullptr, *d_x = nullptr;
    int* d_i = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_cp); (void)hipFree(d_rv); (void)hipFree(d_nz); (void)hipFree(d_b); (void)hipFree(d_A); (void)hipFree(d_x); (void)hipFree(d_i); };
    const size_t nz = (size_t)(nnz > 0 ? nnz : 1);
    bool ok = hipMalloc(&d_cp, (n + 1) * sizeof(long long)) == hipSuccess && hipMalloc(&d_rv, nz * sizeof(long long)) == hipSuccess &&
              hipMalloc(&d_nz, nz * sizeof(double)) == hipSuccess && hipMalloc(&d_b, n * sizeof(double)) == hipSuccess &&
              hipMalloc(&d_A, (size_t)n * n * sizeof(double)) == hipSuccess && hipMalloc(&d_x, n * sizeof(double)) == hipSuccess &&
              hipMalloc(&d_i, (n + 2) * sizeof(int)) == hipSuccess;
    ok = ok && hipMemcpy(d_cp, colptr, (n + 1) * sizeof(long long), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_rv, rowval, (size_t)nnz * sizeof(long long), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_nz, nzval, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(d_b, b, n * sizeof(double), hipMemcpyHostToDevice) == hipSuccess && hipMemset(d_i + n, 0, 2 * sizeof(int)) == hipSuccess;
    int bad[2] = {0, 0};      // [0] row index out of range, [1] singular matrix
    if (ok) {
        hipLaunchKernelGGL(csc_dense_solve_kernel, dim3(1), dim3(256), 0, nullptr, n, d_cp, d_rv, d_nz, d_b, d_A, d_x, d_i, d_i + n);
        ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
             hipMemcpy(x, d_x, n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess &&
             hipMemcpy(bad, d_i + n, 2 * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    }
    cleanup();
    if (!ok) return CIMPC_ERR_HIP;
    if (bad[0]) return CIMPC_ERR_INVALID;
    return bad[1] ? CIMPC_ERR_SINGULAR : CIMPC_OK;      // x then holds Inf / NaN, as after a failed factorization
}

