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
// wavefront 0 -> NewtonDev::stats[8..]: [0] pre-pass  [1] panel  [2] fetch of the entering rows  [3] barrier A  [4] commit  [5] update
// [6] barrier B  [7] back substitution  [8] control recovery; wavefront 4 -> stats[20..]: [0] time before barrier A  [1] update  [2] barrier B
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
    // row i enters the window (columns i-w .. i): a row has at most w + 1 <= 192 entries - one per thread; the
    // values are FETCHED one block ahead (global loads of the sensitivities / weights stay off the critical path)
    // (round 4: the entering rows belong to the threads FROM 128 ON - wavefront 0 runs the panel of a block on its own meanwhile)
    const int rt = (int)nt >= 128 + w + 2 ? tid - 128 : tid;      // row-entry index of this thread (negative: none)
    const bool fast_panel = rt != tid && M <= 128;                 // one wavefront holds the window's rows two per lane
    auto row_value = [&](int i) -> double {
        if (i >= N || rt < 0) return 0.0;
        if (rt <= w) { const int j = i - w + rt; return j >= 0 ? row(i, j) : 0.0; }
        return rt == w + 1 ? rhs(i) : 0.0;                       // entry w + 1 carries the right-hand side
    };
    auto row_commit = [&](int i, double v) {
        if (i >= N || rt < 0) return;
        const int si = i % M;
        if (rt <= w) {
            const int j = i - w + rt;
            if (j >= 0) { const int sj = j % M; W[(size_t)si * MS + sj] = v; W[(size_t)sj * MS + si] = v; }   // and its mirror image
        } else if (rt == w + 1) yw[si] = v;
    };
    for (int i = 0; i < M && i < N; ++i) row_commit(i, row_value(i));
    for (int e = tid; e < RB * MS; e += nt) PL[e] = 0.0;
    __syncthreads();
    const int tx = tid & 63, ty = tid >> 6, nty = nt >> 6;       // update: column tx + 64 q of row ty + nty p (relative to k + RB)
    double nxt[RB];
#pragma unroll
    for (int t = 0; t < RB; ++t) nxt[t] = row_value(M + t);      // rows entering after the first block
    for (int k = 0; k < N; k += RB) {
        const int nb_ = min(RB, N - k);                          // pivots of this block
        // ---- panel: the block's pivots one after the other, each column first brought up to date with the
        //      earlier pivots of the block (left-looking inside the panel) ------------------------------------
        // Round 4: ONE wavefront runs the whole panel (two rows of the window per lane, hand-overs between the steps inside the
        // wavefront - LDS operations of a wavefront execute in order) while the others fetch the entering rows; the workgroup meets
        // once per block before the update instead of once per pivot (the per-pivot cost was 2.2 us, most of it four 16-wave
        // barriers per block with two of the sixteen waves working).  Same arithmetic per entry.
        if (fast_panel) {
            if (tid < 64) {
                // rows of the window by their offset q from the block's first pivot: lane tid owns q = tid and tid + 64 for the whole
                // block, so its entries in the block's pivot columns, its multipliers and its right-hand-side entries stay in
                // REGISTERS through the RB steps; what a step needs of another row (the pivot row's multipliers and right-hand side,
                // the pivot itself) comes by v_readlane from the lane that owns it - no LDS round trip inside the panel.
                auto rl = [](double v, int lane) {
                    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
                    return __hiloint2double(hi, lo);
                };
                int sq[2]; bool vq[2];
                double a[2][RB], l[2][RB], yr[2], dvr[RB];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int q = tid + 64 * hh;
                    vq[hh] = q < M;
                    sq[hh] = vq[hh] ? (k + q) % M : 0;
                    yr[hh] = vq[hh] ? yw[sq[hh]] : 0.0;
#pragma unroll
                    for (int t = 0; t < RB; ++t) { a[hh][t] = (vq[hh] && t < nb_) ? W[(size_t)sq[hh] * MS + (k + t) % M] : 0.0; l[hh][t] = 0.0; }
                }
#pragma unroll
                for (int t = 0; t < RB; ++t) {
                    dvr[t] = 0.0;
                    if (t < nb_) {
                        const int p = k + t, m = min(w, N - 1 - p);
#pragma unroll
                        for (int u = 0; u < t; ++u) {       // left-looking: bring column t up to date with the block's earlier pivots
                            const double lp = rl(l[0][u], t), c = dvr[u];
#pragma unroll
                            for (int hh = 0; hh < 2; ++hh) a[hh][t] = fma(-l[hh][u] * c, lp, a[hh][t]);
                        }
                        const double d = rl(a[0][t], t), inv = 1.0 / d, yp = rl(yr[0], t);
                        dvr[t] = d;
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const int rr = tid + 64 * hh - t;
                            const bool coupled = vq[hh] && rr >= 1 && rr <= m;
                            const double lv = coupled ? a[hh][t] * inv : 0.0;
                            l[hh][t] = lv;
                            if (coupled) {
                                Lr[(size_t)(p + rr) * LW + (w - rr)] = lv;
                                yr[hh] = fma(-lv, yp, yr[hh]);            // forward substitution rides along
                            }
                            if (vq[hh]) PL[(size_t)t * MS + sq[hh]] = lv;   // all M slots: zero where the pivot does not couple
                        }
                        if (tid == t) { yg[p] = yp; Lr[(size_t)p * LW + w] = inv; }
                    }
                }
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) if (vq[hh]) yw[sq[hh]] = yr[hh];
                if (tid == 0) {
#pragma unroll
                    for (int t = 0; t < RB; ++t) dv[t] = dvr[t];
                }
            }
        } else
        for (int t = 0; t < nb_; ++t) {
            const int p = k + t, sp = p % M, m = min(w, N - 1 - p);
            if (tid < M) {
                int sr = sp + tid; if (sr >= M) sr -= M;         // slot of row p + tid
                double v = 0.0;
                if (tid <= m) {
                    v = W[(size_t)sr * MS + sp];
                    for (int u = 0; u < t; ++u) v = fma(-PL[(size_t)u * MS + sr] * dv[u], PL[(size_t)u * MS + sp], v);
                }
                // every thread needs the updated pivot: recompute it (row p itself: sr = sp)
                double d = W[(size_t)sp * MS + sp];
                for (int u = 0; u < t; ++u) d = fma(-PL[(size_t)u * MS + sp] * dv[u], PL[(size_t)u * MS + sp], d);
                const double inv = 1.0 / d, yp = yw[sp];
                const double l = (tid >= 1 && tid <= m) ? v * inv : 0.0;
                // (this phase only writes PL[t], dv[t], yw of the coupled rows: nothing another thread reads here)
                if (tid >= 1 && tid <= m) {
                    Lr[(size_t)(p + tid) * LW + (w - tid)] = l;
                    yw[sr] = fma(-l, yp, yw[sr]);                // forward substitution rides along
                }
                if (tid == 0) { dv[t] = d; yg[p] = yp; Lr[(size_t)p * LW + w] = inv; }
                PL[(size_t)t * MS + sr] = l;                     // all M slots: zero where the pivot does not couple
            }
            lds_barrier();
        }
        // ---- rows entering the window: fetch for the NEXT block, commit this block's (their slots - the
        //      pivots' - are not touched by the update below) -------------------------------------------------
        BPROF(1)
        double nxt2[RB];
#pragma unroll
        for (int t = 0; t < RB; ++t) nxt2[t] = row_value(k + M + RB + t);
        BPROF(2)
        if (fast_panel) lds_barrier();        // the panel's multipliers / pivots are in place; the pivots' slots may be overwritten now
        BPROF(3)
#pragma unroll
        for (int t = 0; t < RB; ++t) { if (t < nb_) row_commit(k + M + t, nxt[t]); nxt[t] = nxt2[t]; }
        BPROF(4)
        // ---- rank-nb_ update of the trailing window, rows / columns k+RB .. k+RB-1+w --------------------------
        {
            const int base = k + nb_, mt = min(w, N - base);     // mt rows / columns present
            if (mt > 0) {
                int sb = base % M;
                double dd[RB];
#pragma unroll
                for (int t = 0; t < RB; ++t) dd[t] = t < nb_ ? dv[t] : 0.0;
                const int nqf = mt >> 6, tail = mt - (nqf << 6);
                double lj[2][RB];
                int sjv[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    int sj = sb + tx + 64 * q; if (sj >= M) sj -= M; if (sj >= M) sj -= M;
                    sjv[q] = q < nqf ? sj : M;
#pragma unroll
                    for (int t = 0; t < RB; ++t) lj[q][t] = (q < nqf && t < nb_) ? PL[(size_t)t * MS + sj] : 0.0;
                }
                if (nqf > 0) {
                    for (int r0 = ty; r0 < mt; r0 += 3 * nty) {
                        double t_[3][2], li[3][RB];
                        double* Wi[3];
#pragma unroll
                        for (int u = 0; u < 3; ++u) {
                            const int r = r0 + nty * u;
                            const bool on = r < mt;
                            int si = sb + (on ? r : 0); if (si >= M) si -= M; if (si >= M) si -= M;
#pragma unroll
                            for (int t = 0; t < RB; ++t) li[u][t] = on ? PL[(size_t)t * MS + si] * dd[t] : 0.0;
                            Wi[u] = W + (size_t)(on ? si : M) * MS;
#pragma unroll
                            for (int q = 0; q < 2; ++q) if (q < nqf) t_[u][q] = Wi[u][sjv[q]];
                        }
#pragma unroll
                        for (int u = 0; u < 3; ++u)
#pragma unroll
                            for (int q = 0; q < 2; ++q) if (q < nqf) {
                                double acc = t_[u][q];
#pragma unroll
                                for (int t = 0; t < RB; ++t) acc = fma(-li[u][t], lj[q][t], acc);
                                Wi[u][sjv[q]] = acc;
                            }
                    }
                }
                for (int e = tid; e < mt * tail; e += nt) {      // tail columns: one entry per thread over all rows
                    const int rr = e / tail, c = (nqf << 6) + (e - rr * tail);
                    int si = sb + rr; if (si >= M) si -= M; if (si >= M) si -= M;
                    int sj = sb + c; if (sj >= M) sj -= M; if (sj >= M) sj -= M;
                    double* cell = W + (size_t)si * MS + sj;
                    double acc = *cell;
#pragma unroll
                    for (int t = 0; t < RB; ++t) acc = fma(-PL[(size_t)t * MS + si] * dd[t], PL[(size_t)t * MS + sj], acc);
                    *cell = acc;
                }
            }
        }
        BPROF(5)
        lds_barrier();
        BPROF(6)
    }
    __syncthreads();              // full barrier: the rows of L and y in global memory are read back below
    // ---- back substitution  L^T x = D^-1 y  (wavefront 0; acc[j] collects sum_{i > j} L[i][j] x_i) ---------
    double* D = K.delta + (size_t)b * S.N;
    if (tid < 64) {
        double* acc = yw;                                        // reuse: [M]
        for (int c = tid; c < M; c += 64) acc[c] = 0.0;
        __builtin_amdgcn_wave_barrier();
        // the rows of L come from global memory: a ring of PD rows in flight (one wavefront, nothing else to hide the latency)
        constexpr int PD = 16;
        double pre[PD][3], pre_y[PD];
        auto fetch = [&](int i, int slot) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { const int c = tid + 64 * q; pre[slot][q] = (i >= 0 && c <= w) ? Lr[(size_t)i * LW + c] : 0.0; }
            pre_y[slot] = i >= 0 ? yg[i] : 0.0;
        };
#pragma unroll
        for (int u = 0; u < PD; ++u) fetch(N - 1 - u, u);
        for (int i0 = N - 1; i0 >= 0; i0 -= PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int i = i0 - u;
                const double cur[3] = {pre[u][0], pre[u][1], pre[u][2]}, yi = pre_y[u];
                fetch(i - PD, u);
                if (i < 0) continue;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const int si = i % M;
                // 1 / d_i sits at c = w: lane (w % 64), register (w / 64)
                double di = 0.0;
#pragma unroll
                for (int q = 0; q < 3; ++q) if (w / 64 == q) di = __shfl(cur[q], w % 64, 64);
                const double xi = yi * di - acc[si];               // di = 1 / d_i
                __builtin_amdgcn_wave_barrier();
                if (tid == 0) { acc[si] = 0.0; D[orig(i)] = xi; }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int c = tid + 64 * q, j = i - w + c;
                    if (c < w && j >= 0) acc[j % M] = fma(cur[q], xi, acc[j % M]);
                }
            }
        }
    }
    __syncthreads();
    BPROF(7)
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
    BPROF(8)
#ifdef CIMPC_BANDED_PROF
    if (b == 0 && (tid == 0 || tid == 256)) for (int j = 0; j < 12; ++j) ((long long*)S.stats)[8 + (tid == 0 ? 0 : 12) + j] = bp[j];
#endif
    if (K.finish) {
        __threadfence_block();
        start_line_search<BlockSync>(S, b, 1, tid, nt);
    }
}

static size_t banded_lds_bytes(int w, int rb = 4) {      // window (w + rb slots + dummy)^2, right-hand side, rb multiplier rows, pivots
    const size_t MS = (size_t)w + rb + 1;
    return (MS * MS + MS + (size_t)rb * MS + 8) * sizeof(double);
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

