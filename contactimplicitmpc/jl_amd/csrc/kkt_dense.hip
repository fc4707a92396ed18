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
// rows of L go to a global workspace for the back substitution (one wavefront, axpy form, LDS window).
// Work: N w^2 / 2 multiply-adds (centroidal H = 60: 25 M, against (2/3) N^3 = 16 G of the dense LU).
// ---------------------------------------------------------------------------------------------------------
namespace {
struct BandRows {
    const NewtonDev& S; const DenseLayout& L; const double* dzb; double rho; int s, nths;
    const double* G;      // reduced form (controls eliminated): [H][nd x nd] du1_t R_t^-1 du1_t^T, column-major; nullptr = full form
    // entry (i, j), j <= i, of the interleaved KKT matrix
    __device__ double operator()(int i, int j) const {
        const int nq = L.nq, nu = L.nu, nr = L.nr, nd = L.nd, H = L.H;
        const int ti = i / s, ki = i - ti * s, tj = j / s, kj = j - tj * s;
        const int dt = ti - tj;
        if (G != nullptr) {      // ordering [q_{t+2}, nu_t] per step, s = nq + nd
            if (ki < nq) {                                       // q row
                if (kj >= nq) return 0.0;
                const size_t e = (size_t)kj * nq + ki;
                if (dt == 0) {
                    double v = S.Q[(size_t)ti * nq * nq + e];
                    if (S.V != nullptr) {
                        v += S.V[(size_t)ti * nq * nq + e];
                        if (ti + 1 < H) v += S.V[(size_t)(ti + 1) * nq * nq + e];
                    }
                    return v;
                }
                if (dt == 1 && S.V != nullptr) return -S.V[(size_t)ti * nq * nq + e];
                return 0.0;
            }
            const int kd = ki - nq;                              // dual row nu_t
            const double* dz = dzb + (size_t)ti * nths * nd;
            if (dt == 0) {
                if (kj < nq) return (kj == kd) ? -1.0 : 0.0;     // -I at q_{t+2}
                const double g = G[(size_t)ti * nd * nd + (size_t)(kj - nq) * nd + kd];
                return (kj == ki) ? -rho - g : -g;               // -(rho I + du1 R^-1 du1^T)
            }
            if (kj >= nq) return 0.0;
            if (dt == 1) return dz[(size_t)(nq + kj) * nd + kd]; // dq1 at q_{t+1}
            if (dt == 2) return dz[(size_t)kj * nd + kd];        // dq0 at q_t
            return 0.0;
        }
        if (ki < nu) {                                           // u row: R_t (same block only)
            return (dt == 0 && kj < nu) ? S.R[(size_t)ti * nu * nu + (size_t)kj * nu + ki] : 0.0;
        }
        if (ki < nr) {                                           // q row
            const int kq = ki - nu;
            if (kj < nu || kj >= nr) return 0.0;                 // (lower part: no dual columns before a q row)
            const int cq = kj - nu;
            const size_t e = (size_t)cq * nq + kq;
            if (dt == 0) {
                double v = S.Q[(size_t)ti * nq * nq + e];
                if (S.V != nullptr) {
                    v += S.V[(size_t)ti * nq * nq + e];
                    if (ti + 1 < H) v += S.V[(size_t)(ti + 1) * nq * nq + e];
                }
                return v;
            }
            if (dt == 1 && S.V != nullptr) return -S.V[(size_t)ti * nq * nq + e];
            return 0.0;
        }
        const int kd = ki - nr;                                  // dual row nu_t
        const double* dz = dzb + (size_t)ti * nths * nd;
        if (dt == 0) {
            if (kj < nu) return dz[(size_t)(2 * nq + kj) * nd + kd];          // du1
            if (kj < nr) return (kj - nu == kd) ? -1.0 : 0.0;                 // -I at q_{t+2}
            return (kj == ki) ? -rho : 0.0;
        }
        if (kj < nu || kj >= nr) return 0.0;
        if (dt == 1) return dz[(size_t)(nq + kj - nu) * nd + kd];             // dq1 at q_{t+1}
        if (dt == 2) return dz[(size_t)(kj - nu) * nd + kd];                  // dq0 at q_t
        return 0.0;
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

// doubles of workspace per rollout: rows of L (N x (w + 1)), y (N), then G and g of the reduced form.  Sized for the FULL form
// (the larger one) so that the host's allocation does not depend on which form a launch takes.
__host__ __device__ inline size_t banded_ws_per_rollout(const NewtonDev& S) {
    const size_t sF = (size_t)S.nr + S.nd, NF = (size_t)S.dm.H * sF;
    size_t wF = 3 * sF - 1 - S.dm.nu; if (wF > NF - 1) wF = NF - 1;
    return NF * (wF + 1) + NF + (size_t)S.dm.H * S.nd * (S.nd + 1);
}

// -DCIMPC_BANDED_PROF (diagnostic builds, with -DCIMPC_KKT_PROF for the accessor): shader-clock accounting of the banded kernel, rollout 0,
// wavefront 0 -> NewtonDev::stats[8..]: [0] pre-pass + window fill  [1] P1 diagonal block  [2] fetch issue + barrier  [3] P2 rows  [4] barrier
// [5] stores + commit  [6] update  [7] barrier  [8] back substitution  [9] control recovery; wavefront 4 -> stats[20..], same slots
#ifdef CIMPC_BANDED_PROF
#define BPROF(j) { const long long tn_ = clock64(); bp[j] += tn_ - bt; bt = tn_; }
#else
#define BPROF(j)
#endif
#ifndef CIMPC_BANDED_THREADS
#define CIMPC_BANDED_THREADS 1024      // (build parameter: 256 / 512 / 1024 measured in round 4)
#endif
// RB: pivots per window update (4, or 8 where the window of w + 8 slots fits the LDS: the reduced form of every compiled model).  The
// per-entry cost of an update pass is index arithmetic and LDS traffic, not its RB multiply-adds (profiles/r04/banded_prof_before.log:
// update + its barrier 43 % of a solve, 7-8 k cycles per pass at RB = 4), so half as many passes are worth the deeper panel.  Every entry
// still receives its pivots' contributions one after the other in pivot order: the factors do not depend on RB.
template <int RB>
__global__ __launch_bounds__(CIMPC_BANDED_THREADS) void kkt_banded_kernel(NewtonDev S, KktArgs K, double* ws_all) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int b = blockIdx.x + S.b0, tid = threadIdx.x, nt = blockDim.x;
    if (K.stage != nullptr && K.stage[b] != STAGE_KKT) return;
    const DenseLayout L(S);
    // Round 4: the controls are eliminated first.  u_t appears in its own block only (R_t, and du1_t in the row of nu_t), so
    //   Du_t = R_t^-1 (r_u,t - du1_t^T Dnu_t)   and the nu_t row becomes   ... - (rho I + du1_t R_t^-1 du1_t^T) Dnu_t = r_d,t - du1_t R_t^-1 r_u,t
    // exactly: the factored matrix is the interleaved KKT matrix of [q_{t+2}, nu_t] - still symmetric quasi-definite - with
    // s = nq + nd rows per step instead of nr + nd and half-bandwidth 3 s - 1 (centroidal H = 60: N 2880 -> 2160, w 131 -> 107,
    // N w^2 49 M -> 25 M, a quarter fewer pivots on the sequential chain).  S.band_reduce = 0 (an R_t the host could not
    // invert) keeps the full form.
    const bool reduced = S.band_reduce != 0 && L.nu > 0;
    const int H = L.H, nr = L.nr, nd = L.nd, nq = L.nq, nu = L.nu;
    const int s = reduced ? nq + nd : nr + nd, N = H * s;
    const int w = min(reduced ? 3 * s - 1 : 3 * s - 1 - nu, N - 1), LW = w + 1, M = w + RB;   // window slots: pivot k+RB-1 reaches row k+RB-1+w
    double* wsb = ws_all + (size_t)b * banded_ws_per_rollout(S);
    double* Lr = wsb;                                            // row i: L[i][i-w .. i-1], slot w: 1 / d_i
    double* yg = Lr + (size_t)N * LW;
    double* Gm = yg + N;                                         // reduced form: [H][nd x nd] du1 R^-1 du1^T, then [H][nd] du1 R^-1 r_u
    const int MS = M + 1;                                        // row stride: slot M is a dummy row / column
    double* W = sm;                                              // [M+1][M+1] window, slot = index mod M; BOTH triangles kept
    double* yw = W + (size_t)MS * MS;                            // [M]   right-hand side of the rows in the window
    double* PL = yw + MS;                                        // [RB][M+1] multipliers of the block's pivots, by row SLOT
    double* dv = PL + (size_t)RB * MS;                           // [RB]  the block's pivots d
    double* dinv = dv + RB;                                      // [RB]  their reciprocals
    double* ypv = dinv + RB;                                     // [RB]  the pivot rows' right-hand sides (after the forward substitution among them)
    double* L11 = ypv + RB;                                      // [RB][RB] multipliers among the block's pivot rows (row-major, strictly lower)
    const double beta = K.beta ? K.beta[b] : K.beta_scalar;
    const double rho = (double)H * beta * S.kappa;               // newton_jacobian.jl:169-186 quirk
    const double* dzb = kkt_dz(S, K, b, H, S.nths, nd);
    const double* rb = K.r + (size_t)b * S.N;
    double* gv = Gm + (size_t)H * nd * nd;
#ifdef CIMPC_BANDED_PROF
    long long bp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, bt = clock64();
#endif
    if (reduced) {      // G_t = du1_t R_t^-1 du1_t^T, g_t = du1_t R_t^-1 r_u,t  (R_t^-1: NewtonDev::Rinv, symmetric); T = du1 R^-1 staged in LDS
        double* Tm = sm;                                         // [H][nd x nu] (the window is not in use yet)
        for (int e = tid; e < H * nd * nu; e += nt) {
            const int t = e / (nd * nu), k = e - t * nd * nu, r = k % nd, c = k / nd;
            const double* du = dzb + ((size_t)t * S.nths + 2 * nq) * nd;      // du1_t, nd x nu column-major
            const double* Ri = S.Rinv + (size_t)t * nu * nu + (size_t)c * nu;
            double a = 0.0;
            for (int m = 0; m < nu; ++m) a = fma(du[(size_t)m * nd + r], Ri[m], a);
            Tm[e] = a;
        }
        __syncthreads();
        for (int e = tid; e < H * nd * (nd + 1); e += nt) {
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
    BPROF(0)
    const BandRows row{S, L, dzb, rho, s, S.nths, reduced ? Gm : nullptr};
    // interleaved index -> index in the reference's layout (primal segment step-major, then the duals)
    auto orig = [&](int i) {
        const int t = i / s, k = i - t * s;
        if (reduced) return k < nq ? t * nr + nu + k : H * nr + t * nd + (k - nq);
        return k < nr ? t * nr + k : H * nr + t * nd + (k - nr);
    };
    // right-hand side entry of interleaved row i (reduced form: the dual rows carry r_d - du1 R^-1 r_u)
    auto rhs = [&](int i) {
        const double v = rb[orig(i)];
        if (!reduced) return v;
        const int t = i / s, k = i - t * s;
        return k < nq ? v : v - gv[t * nd + (k - nq)];
    };
    // A row entering the window has w + 2 entries: columns i-w .. i and the right-hand side.  Only the LOWER triangle of the window is
    // kept (row >= column: nothing ever reads the other one).  The entries of the RB rows that enter per block belong to the threads from
    // 128 on, at most NE each, and are FETCHED one block ahead (global loads of the sensitivities / weights stay off the critical path).
    constexpr int NE = 2;
    const int EW = w + 2;
    auto entry_value = [&](int i, int e) -> double {
        if (i >= N) return 0.0;
        if (e <= w) { const int j = i - w + e; return j >= 0 ? row(i, j) : 0.0; }
        return rhs(i);
    };
    auto entry_commit = [&](int i, int e, double v) {
        if (i >= N) return;
        const int si = i % M;
        if (e <= w) { const int j = i - w + e; if (j >= 0) W[(size_t)si * MS + j % M] = v; }
        else yw[si] = v;
    };
    {
        const int n0 = min(M, N) * EW;                           // the first M rows, all threads
        for (int idx = tid; idx < n0; idx += nt) { const int i = idx / EW, e = idx - i * EW; entry_commit(i, e, entry_value(i, e)); }
    }
    for (int e = tid; e < RB * MS; e += nt) PL[e] = 0.0;
    int ent_t[NE], ent_e[NE];                                    // (row of the block, entry) of this thread's entering entries; -1: none
#pragma unroll
    for (int n = 0; n < NE; ++n) {
        const int idx = tid - 128 + n * ((int)nt - 128);
        const bool on = tid >= 128 && idx < RB * EW;
        ent_t[n] = on ? idx / EW : -1;
        ent_e[n] = on ? idx - (idx / EW) * EW : 0;
    }
    double nxt[NE];
#pragma unroll
    for (int n = 0; n < NE; ++n) nxt[n] = ent_t[n] >= 0 ? entry_value(M + ent_t[n], ent_e[n]) : 0.0;      // rows entering after the first block
    __syncthreads();
    const int tx = tid & 63, ty = tid >> 6, nty = nt >> 6;
    auto rl = [](double v, int lane) {                           // v of another lane of the wavefront (lane: uniform)
        const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
        return __hiloint2double(hi, lo);
    };
    auto wrap = [&](int x) { if (x >= M) x -= M; if (x >= M) x -= M; return x; };      // slot of an index < 3 M past a slot
    // Round 4 (second pass): a block of RB pivots is
    //   P1  lanes 0 .. RB-1 of wavefront 0: the RB x RB diagonal block - pivots, reciprocals, the multipliers among the pivot rows, the
    //       forward substitution among them (lane r owns pivot row r, right-looking, hand-overs by v_readlane, no branches, no global
    //       stores) - the only sequential part;
    //   P2  RB lanes per row below the block (lane t owns the row's entry in pivot column t): the triangular solve against the diagonal
    //       block runs ACROSS the lanes - per pivot u two multiplies, a DPP row broadcast of l_u d_u and one multiply-add;
    //   P3  all threads: the multipliers go to global memory (RB lanes write one row's contiguous piece), the right-hand side of the rows
    //       below is updated, the entering rows are committed, and the lower triangle of the trailing window gets its rank-RB update
    //       as 16 x 16 tiles on v_mfma_f64_16x16x4 (A = -(l d) of the tile's rows, B = l of its columns, K = RB).
    // P1 / P2 apply the same operations to every entry in the same order as the one-pivot-at-a-time form.  Three LDS-only barriers per block.
    // History (cycles per block of 8 pivots, w = 107): round-4 first form 42 k (banded_prof_rb8_regpanel.log) - the panel of all 115 rows on
    // one wavefront 12.5 k, its scattered 8-byte stores draining at the next barrier 7.8 k, an LDS-bound update of both triangles 15 k;
    // split into P1 / P2 / P3 with a register-tiled VALU update 27 k (banded_prof_split.log: P1 10.5 k - integer divisions for the slots,
    // a branch per predicate -, P2 3.5 k, update 7.6 k).
    int sk = 0;                                                  // k % M, kept by counting
    for (int k = 0; k < N; k += RB) {
        const int nb_ = min(RB, N - k);                          // pivots of this block
        if (tid < RB) {                                          // ---- P1
            const int r = tid;
            const bool vr = r < nb_;
            const int sr = wrap(sk + r);
            double a[RB];
            double yr = vr ? yw[sr] : 0.0;
#pragma unroll
            for (int t = 0; t < RB; ++t) a[t] = (vr && t <= r && t < nb_) ? W[(size_t)sr * MS + wrap(sk + t)] : 0.0;
#pragma unroll
            for (int t = 0; t < RB; ++t) {
                if (t < nb_) {
                    const double d = rl(a[t], t), inv = 1.0 / d, yp = rl(yr, t);
                    const double lv = r > t ? a[t] * inv : 0.0;  // (rows past the end carry zeros)
                    yr = fma(-lv, yp, yr);                       // forward substitution rides along
                    const double c = lv * d;
#pragma unroll
                    for (int t2 = t + 1; t2 < RB; ++t2) a[t2] = fma(-c, rl(lv, t2), a[t2]);      // right-looking inside the block
                    L11[r * RB + t] = lv;
                    dv[t] = d; dinv[t] = inv; ypv[t] = yp;       // (the same value from every lane)
                }
            }
        }
        BPROF(1)
        double nxt2[NE];                                         // rows entering after the NEXT block: loads in flight across P1 / P2
#pragma unroll
        for (int n = 0; n < NE; ++n) nxt2[n] = ent_t[n] >= 0 ? entry_value(k + RB + M + ent_t[n], ent_e[n]) : 0.0;
        lds_barrier();
        BPROF(2)
        {                                                        // ---- P2: row k + q, q = RB .. RB + w - 1, entry t
            constexpr int NG = 16 / RB;                          // rows per DPP row of 16 lanes
            const int l16 = tx & 15, g = l16 / RB, t = l16 - g * RB;
            const int q = RB + (tid >> 4) * NG + g, i = k + q;
            const bool on = q < RB + w && i < N && t < nb_;
            const int sr = wrap(sk + q);
            double a = (on && q - t <= w) ? W[(size_t)sr * MS + wrap(sk + t)] : 0.0;
            const double inv = dinv[t], d = dv[t];
            double lm[RB];                                       // the pivot row t's multipliers for the earlier pivots
#pragma unroll
            for (int u = 0; u < RB; ++u) lm[u] = u < t ? L11[t * RB + u] : 0.0;
            static_for<0, RB - 1>([&](auto U) {
                constexpr int u = decltype(U)::value;
                const double c = (a * inv) * d;                  // final in lane u of the row
                double cb = 0.0;
                static_for<0, NG>([&](auto Gq) {
                    constexpr int g2 = decltype(Gq)::value;
                    const double v = __builtin_amdgcn_update_dpp(0.0, c, 0x150 + g2 * RB + u, 0xF, 0xF, true);      // row_newbcast
                    cb = g == g2 ? v : cb;
                });
                a = fma(-cb, lm[u], a);
            });
            if (on) PL[(size_t)t * MS + sr] = a * inv;           // zero where the pivot does not couple
        }
        BPROF(3)
        lds_barrier();              // the block's multipliers / pivots are in place; the pivots' slots may be overwritten now
        BPROF(4)
        // ---- P3 -------------------------------------------------------------------------------------------------------------------
        for (int idx = tid; idx < (RB - 1 + w) * RB; idx += nt) {       // multipliers -> rows of L in global memory, t fastest
            const int q = 1 + idx / RB, t = idx - (idx / RB) * RB, i = k + q;
            if (i < N && t < nb_ && t < q && q - t <= w)
                Lr[(size_t)i * LW + (w - (q - t))] = q < RB ? L11[q * RB + t] : PL[(size_t)t * MS + wrap(sk + q)];
        }
        if (tid < nb_) { yg[k + tid] = ypv[tid]; Lr[(size_t)(k + tid) * LW + w] = dinv[tid]; }
        if (tid >= 64 && tid < 64 + w && k + RB + tid - 64 < N) {       // right-hand side of the rows below the block
            const int sr = wrap(sk + RB + tid - 64);
            double yr = yw[sr];
#pragma unroll
            for (int t = 0; t < RB; ++t) if (t < nb_) yr = fma(-PL[(size_t)t * MS + sr], ypv[t], yr);
            yw[sr] = yr;
        }
#pragma unroll
        for (int n = 0; n < NE; ++n) {
            if (ent_t[n] >= 0 && ent_t[n] < nb_ && k + M + ent_t[n] < N) {       // row k + M + t: the slot of pivot k + t
                const int si = wrap(sk + ent_t[n]);
                if (ent_e[n] <= w) W[(size_t)si * MS + wrap(sk + RB + ent_t[n] + ent_e[n])] = nxt[n];   // column k + RB + t + e (>= 0)
                else yw[si] = nxt[n];
            }
            nxt[n] = nxt2[n];
        }
        BPROF(5)
        {
            const int base = k + nb_, mt = min(w, N - base);     // mt rows / columns of the trailing window present
            if (mt > 0) {
                const int sb = wrap(sk + nb_);
                const int nT = (mt + 15) >> 4, ntiles = nT * (nT + 1) / 2;
                const int li = tx & 15, lk = tx >> 4;
                for (int tl = ty; tl < ntiles; tl += nty) {
                    int I = 0;
                    while ((I + 1) * (I + 2) / 2 <= tl) ++I;     // tile (I, J), J <= I
                    const int J = tl - I * (I + 1) / 2;
                    const int ra = 16 * I + li, cb_ = 16 * J + li;
                    const int sra = wrap(sb + min(ra, mt - 1)), scb = wrap(sb + min(cb_, mt - 1));
                    d4 acc;
                    double* cell[4];
                    bool ok[4];
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const int rw = 16 * I + lk + 4 * r4;
                        ok[r4] = rw < mt && cb_ < mt && (I != J || rw >= cb_);      // lower triangle only
                        cell[r4] = W + (size_t)wrap(sb + min(rw, mt - 1)) * MS + scb;
                        acc[r4] = ok[r4] ? *cell[r4] : 0.0;
                    }
#pragma unroll
                    for (int kb = 0; kb < RB / 4; ++kb) {
                        const int t = 4 * kb + lk;
                        const double av = ra < mt ? -(PL[(size_t)t * MS + sra] * dv[t]) : 0.0;
                        const double bv = cb_ < mt ? PL[(size_t)t * MS + scb] : 0.0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                    }
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) if (ok[r4]) *cell[r4] = acc[r4];
                }
            }
        }
        BPROF(6)
        lds_barrier();
        BPROF(7)
        sk += RB; if (sk >= M) sk -= M;
    }
    __syncthreads();              // full barrier: the rows of L and y in global memory are read back below
    // ---- back substitution  L^T x = D^-1 y.  acc_j = sum_{i > j} L[i][j] x_i is built row by row (i descending); round 4: the pending
    //      acc_j live in REGISTERS of wavefront 0 (column j belongs to lane (j + 192) % 64, register ((j + 192) / 64) % 3 - a band of
    //      w < 192 columns never holds two columns of one lane in one register), x_i needs one v_readlane, and the rows of L come
    //      through LDS in chunks of CR rows that the other fifteen wavefronts copy one chunk ahead (a contiguous piece of global memory).
    //      A lane follows its column of a register by counting: offset c = j - (i - w) grows by one per row; at c = w the column is the
    //      row's own (x_j is read off, the sum restarts) and the lane moves on to column j - 192.
    //      Before: 3.6 k cycles per row - an LDS read-modify-write chain on acc and global-load latency behind a 16-row register ring.
    double* D = K.delta + (size_t)b * S.N;
    {
        const int avail = MS * MS + MS + RB * MS + 3 * RB + RB * RB;      // doubles of LDS the window held
        int CR = avail / (2 * (LW + 1)); if (CR > 64) CR = 64;
        const int nch = (N + CR - 1) / CR, CS = CR * (LW + 1);
        auto stage = [&](int ch, int t0, int nth) {              // chunk ch = rows N-1 - ch CR downwards -> buffer ch & 1  (threads t0 .. of nth)
            const int i_hi = N - 1 - ch * CR, i_lo = max(i_hi - CR + 1, 0), n = i_hi - i_lo + 1, tot = n * LW;
            double* bufp = sm + (size_t)(ch & 1) * CS;
            const double* src = Lr + (size_t)i_lo * LW;
            for (int e0 = t0; e0 < tot; e0 += 8 * nth) {
                double tmp[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int e = e0 + u * nth; tmp[u] = e < tot ? src[e] : 0.0; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int e = e0 + u * nth; if (e < tot) bufp[e] = tmp[u]; }
            }
            for (int e = t0; e < n; e += nth) bufp[CR * LW + e] = yg[i_lo + e];
        };
        stage(0, tid, nt);
        __syncthreads();
        double acc[3] = {0.0, 0.0, 0.0};
        int cc[3];                                               // offset c of this lane's column of register sl at the current row
        {
            const int bb = N - 1 - w + 192, blk = bb >> 6, bm = blk % 3;
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) {
                int dB = sl - bm; if (dB < 0) dB += 3;
                int jj = 64 * (blk + dB) + tx; if (jj < bb) jj += 192;
                cc[sl] = jj - bb; if (cc[sl] > w) cc[sl] -= 192;
            }
        }
        for (int ch = 0; ch < nch; ++ch) {
            if (tid >= 64) { if (ch + 1 < nch) stage(ch + 1, tid - 64, (int)nt - 64); }
            else {
                const int i_hi = N - 1 - ch * CR, i_lo = max(i_hi - CR + 1, 0);
                const double* bufp = sm + (size_t)(ch & 1) * CS;
                int ti = i_hi / s, ki = i_hi - ti * s;           // (step, index in the step) of row i, kept by counting
                auto load = [&](int i, double (&Lv)[3], double& di, double& yi) {
                    const double* Lrow = bufp + (size_t)(i - i_lo) * LW;
                    di = Lrow[w]; yi = bufp[CR * LW + (i - i_lo)];       // di = 1 / d_i
                    const int lo = max(w - i, 0);                // columns j >= 0 only
#pragma unroll
                    for (int sl = 0; sl < 3; ++sl) Lv[sl] = (cc[sl] >= lo && cc[sl] < w) ? Lrow[cc[sl]] : 0.0;
                };
                double Lv[3], di, yi;
                load(i_hi, Lv, di, yi);
                for (int i = i_hi; i >= i_lo; --i) {
                    // this row's own column: the lane / register whose offset is w
                    const double mine = cc[0] == w ? acc[0] : (cc[1] == w ? acc[1] : acc[2]);
                    const double xi = yi * di - rl(mine, (i + 192) & 63);
#pragma unroll
                    for (int sl = 0; sl < 3; ++sl) {
                        acc[sl] = cc[sl] == w ? 0.0 : fma(Lv[sl], xi, acc[sl]);
                        cc[sl] = cc[sl] == w ? w - 191 : cc[sl] + 1;
                    }
                    if (tid == 0) {
                        int o;
                        if (reduced) o = ki < nq ? ti * nr + nu + ki : H * nr + ti * nd + (ki - nq);
                        else o = ki < nr ? ti * nr + ki : H * nr + ti * nd + (ki - nr);
                        D[o] = xi;
                    }
                    if (--ki < 0) { ki = s - 1; --ti; }
                    if (i > i_lo) load(i - 1, Lv, di, yi);
                }
            }
            __syncthreads();
        }
    }
    BPROF(8)
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
            const double* Ri = S.Rinv + (size_t)t * nu * nu + (size_t)m * nu;      // column m = row m (symmetric)
            double a = 0.0;
            for (int k = 0; k < nu; ++k) a = fma(Ri[k], tu[t * nu + k], a);
            D[t * nr + m] = a;
        }
        __syncthreads();
    }
    BPROF(9)
#ifdef CIMPC_BANDED_PROF
    if (b == 0 && (tid == 0 || tid == 256)) for (int j = 0; j < 12; ++j) ((long long*)S.stats)[8 + (tid == 0 ? 0 : 12) + j] = bp[j];
#endif
    if (K.finish) {
        __threadfence_block();
        start_line_search<BlockSync>(S, b, 1, tid, nt);
    }
}

static size_t banded_lds_bytes(int w, int rb = 4) {      // window (w + rb slots + dummy)^2, right-hand side, rb multiplier rows, pivots / reciprocals / rhs, diagonal block
    const size_t MS = (size_t)w + rb + 1;
    return (MS * MS + MS + (size_t)rb * MS + 3 * (size_t)rb + (size_t)rb * rb) * sizeof(double);
}
static int band_halfwidth(const NewtonDev& S) {      // (the kernel's own formula: reduced form when the controls are eliminated)
    if (S.band_reduce != 0 && S.dm.nu > 0) { const int s = S.dm.nq + S.nd; return std::min(3 * s - 1, S.dm.H * s - 1); }
    const int s = S.nr + S.nd;
    return std::min(3 * s - 1 - S.dm.nu, S.N - 1);
}
static int banded_rb(const NewtonDev& S) {      // pivots per window update: 8 where that window fits
    return banded_lds_bytes(band_halfwidth(S), 8) <= 156 * 1024 ? 8 : 4;
}
static size_t banded_lds_bytes(const NewtonDev& S) {      // ... and, before the window is in use, T = du1 R^-1 of every step ([H][nd x nu])
    const size_t win = banded_lds_bytes(band_halfwidth(S), banded_rb(S));
    const size_t pre = (S.band_reduce != 0 && S.dm.nu > 0) ? (size_t)S.dm.H * S.nd * S.dm.nu * sizeof(double) : 0;
    return std::max(win, pre);
}
bool kkt_banded_available(const NewtonDev& S) {      // window + two vectors in 160 KB of LDS, back substitution: w < 192
    const int w = band_halfwidth(S);
    return S.dm.mode == CIMPC_MODE_CONFIGURATION && w <= 190 && banded_lds_bytes(S) <= 160 * 1024;
}

size_t kkt_dense_workspace_doubles(const NewtonDev& S, bool banded) {
    if (banded) return (size_t)S.dm.B * banded_ws_per_rollout(S);
    return (size_t)S.dm.B * ((size_t)S.N * S.N + 2 * (size_t)S.N);
}

static int launch_kkt_dense(const NewtonDev& S, const KktArgs& K, double* ws, hipStream_t s, bool banded) {
    if (banded) {
        const size_t lds = banded_lds_bytes(S);
        static LdsOptIn optin;
        if (banded_rb(S) == 8) {
            static LdsOptIn optin8;
            if (lds_opt_in(optin8, (const void*)kkt_banded_kernel<8>, lds) != CIMPC_OK) return CIMPC_ERR_HIP;
            hipLaunchKernelGGL(kkt_banded_kernel<8>, dim3(S.nb_launch), dim3(CIMPC_BANDED_THREADS), lds, s, S, K, ws);
            return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
        }
        if (lds_opt_in(optin, (const void*)kkt_banded_kernel<4>, lds) != CIMPC_OK) return CIMPC_ERR_HIP;
        hipLaunchKernelGGL(kkt_banded_kernel<4>, dim3(S.nb_launch), dim3(CIMPC_BANDED_THREADS), lds, s, S, K, ws);
        return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
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

