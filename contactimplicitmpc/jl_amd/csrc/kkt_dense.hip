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

    // ---- LU with partial pivoting (getrf-style, right-looking), rhs permuted on the fly -----------
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
        __syncthreads();
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
    // entry (i, j), j <= i, of the interleaved KKT matrix
    __device__ double operator()(int i, int j) const {
        const int nq = L.nq, nu = L.nu, nr = L.nr, nd = L.nd, H = L.H;
        const int ti = i / s, ki = i - ti * s, tj = j / s, kj = j - tj * s;
        const int dt = ti - tj;
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

#ifndef CIMPC_BANDED_THREADS
#define CIMPC_BANDED_THREADS 1024
#endif
__global__ __launch_bounds__(CIMPC_BANDED_THREADS) void kkt_banded_kernel(NewtonDev S, KktArgs K, double* ws_all) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int b = blockIdx.x + S.b0, tid = threadIdx.x, nt = blockDim.x;
    if (K.stage != nullptr && K.stage[b] != STAGE_KKT) return;
    const DenseLayout L(S);
    const int N = L.N, H = L.H, nr = L.nr, nd = L.nd, s = nr + nd;
    const int w = min(3 * s - 1 - L.nu, N - 1), M = w + 1;
    double* Lr = ws_all + (size_t)b * ((size_t)N * M + N);       // row i: L[i][i-w .. i-1], slot w: d_i
    double* yg = Lr + (size_t)N * M;
    const int MS = M + 1;                                        // row stride: slot M is a dummy row / column
    double* W = sm;                                              // [M+1][M+1] window, slot = index mod M; BOTH triangles kept
    double* yw = W + (size_t)MS * MS;                            // [M]
    double* lv = yw + M;                                         // [M]: lv[0] = d_k, lv[r] = L[k+r][k]
    const double beta = K.beta ? K.beta[b] : K.beta_scalar;
    const double rho = (double)H * beta * S.kappa;               // newton_jacobian.jl:169-186 quirk
    const double* dzb = kkt_dz(S, K, b, H, S.nths, nd);
    const BandRows row{S, L, dzb, rho, s, S.nths};
    const double* rb = K.r + (size_t)b * S.N;
    // interleaved index -> index in the reference's layout (primal segment step-major, then the duals)
    auto orig = [&](int i) { const int t = i / s, k = i - t * s; return k < nr ? t * nr + k : H * nr + t * nd + (k - nr); };
    // row i enters the window (columns i-w .. i): a row has at most w + 1 <= 192 entries - one per thread; the
    // value is FETCHED one pivot ahead (global loads of the sensitivities / weights stay off the critical path)
    auto row_value = [&](int i) -> double {
        if (i >= N) return 0.0;
        if (tid <= w) { const int j = i - w + tid; return j >= 0 ? row(i, j) : 0.0; }
        return tid == w + 1 ? rb[orig(i)] : 0.0;                 // thread w + 1 carries the right-hand side entry
    };
    auto row_commit = [&](int i, double v) {
        if (i >= N) return;
        const int si = i % M;
        if (tid <= w) {
            const int j = i - w + tid;
            if (j >= 0) { const int sj = j % M; W[(size_t)si * MS + sj] = v; W[(size_t)sj * MS + si] = v; }   // and its mirror image
        } else if (tid == w + 1) yw[si] = v;
    };
    for (int i = 0; i <= w && i < N; ++i) row_commit(i, row_value(i));
    __syncthreads();
    const int tx = tid & 63, ty = tid >> 6, nty = nt >> 6;       // update: column 1 + tx + 64 q of row 1 + ty + nty p
    double nxt = row_value(M);                                   // row entering after pivot 0
    for (int k = 0; k < N; ++k) {
        const int sk = k % M, m = min(w, N - 1 - k);             // rows k+1 .. k+m are coupled to the pivot
        {
            const double d = W[(size_t)sk * MS + sk], inv = 1.0 / d, yk = yw[sk];
            for (int r = tid; r <= m; r += nt) {
                int sr = sk + r; if (sr >= M) sr -= M;
                const double v = r == 0 ? inv : W[(size_t)sr * MS + sk] * inv;    // (slot w of a row of L keeps 1 / d)
                lv[r] = r == 0 ? d : v;
                Lr[(size_t)(k + r) * M + (w - r)] = v;           // r = 0: the diagonal slot
                if (r == 0) yg[k] = yk;
                else yw[sr] = fma(-v, yk, yw[sr]);               // forward substitution rides along
            }
        }
        lds_barrier();
        const double nxt2 = row_value(k + M + 1);                // fetch for the NEXT pivot while this one updates
        // the entering row takes the slots the pivot row / column just freed: nothing below touches slot sk (the
        // multipliers were copied to lv), so it is committed next to the update and one barrier per pivot is saved
        row_commit(k + M, nxt);
        nxt = nxt2;
        {
            // rank-1 update of the whole (symmetric) m x m window, branch-free: a wavefront takes 64 consecutive
            // columns of a row, lanes beyond the coupled rows / columns work on the dummy slot with a zero multiplier.
            // Measured with clock64 per phase (centroidal H = 50: N = 2400, w = 131, 1024 threads, 7.9 k cycles per pivot):
            // this update 5.3 k, pivot column 0.85 k, row generation 0.8 k, commit 0.4 k, barriers 0.6 k.  The update
            // is LDS-bandwidth bound: 16 wavefronts x 36 load + store pairs of 512 bytes at 128 B / clk = 4.6 k cycles,
            // of which 46 % move useful entries (131 of 192 column lanes, 8.2 of 12 row slots).  Blocking several
            // pivots per window update (LDS traffic / block size, MFMA-shaped) is the next step.
            const double d = lv[0];
            const int nqf = m >> 6, tail = m - (nqf << 6);       // uniform: full column chunks of 64 + a tail of < 64 columns
            double lj[2];
            int sjv[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r = 1 + tx + 64 * q;
                lj[q] = q < nqf ? lv[r] : 0.0;
                int sj = sk + r; if (sj >= M) sj -= M;
                sjv[q] = q < nqf ? sj : M;
            }
            if (nqf > 0) {
                for (int r0 = 1 + ty; r0 <= m; r0 += 3 * nty) {  // three rows per trip: their loads are issued together
                    double li[3], t[3][2];
                    double* Wi[3];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const int r = r0 + nty * u;
                        const bool on = r <= m;
                        li[u] = on ? lv[on ? r : 0] * d : 0.0;
                        int si = sk + r; if (si >= M) si -= M;
                        Wi[u] = W + (size_t)(on ? si : M) * MS;
#pragma unroll
                        for (int q = 0; q < 2; ++q) if (q < nqf) t[u][q] = Wi[u][sjv[q]];
                    }
#pragma unroll
                    for (int u = 0; u < 3; ++u)
#pragma unroll
                        for (int q = 0; q < 2; ++q) if (q < nqf) Wi[u][sjv[q]] = fma(-li[u], lj[q], t[u][q]);
                }
            }
            // the tail columns (131 = 2 x 64 + 3 for the centroidal sizes) as one entry per thread over all rows:
            // a third 64-lane chunk would spend a third of the LDS bandwidth on three useful lanes
            for (int e = tid; e < m * tail; e += nt) {
                const int rr = e / tail, r = 1 + rr, c = 1 + (nqf << 6) + (e - rr * tail);
                int si = sk + r; if (si >= M) si -= M;
                int sj = sk + c; if (sj >= M) sj -= M;
                double* cell = W + (size_t)si * MS + sj;
                *cell = fma(-lv[r] * d, lv[c], *cell);
            }
        }
        lds_barrier();
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
            for (int q = 0; q < 3; ++q) { const int c = tid + 64 * q; pre[slot][q] = (i >= 0 && c <= w) ? Lr[(size_t)i * M + c] : 0.0; }
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
    if (K.finish) {
        __threadfence_block();
        start_line_search<BlockSync>(S, b, 1, tid, nt);
    }
}

static int band_halfwidth(const NewtonDev& S) {
    const int s = S.nr + S.nd;
    return std::min(3 * s - 1 - S.dm.nu, S.N - 1);
}
bool kkt_banded_available(const NewtonDev& S) {      // window + two vectors in 160 KB of LDS, back substitution: w < 192
    const int M = band_halfwidth(S) + 1;
    return S.dm.mode == CIMPC_MODE_CONFIGURATION && M <= 192 && ((size_t)(M + 1) * (M + 1) + 2 * (size_t)M) * sizeof(double) <= 160 * 1024;
}

size_t kkt_dense_workspace_doubles(const NewtonDev& S, bool banded) {
    if (banded) return (size_t)S.dm.B * ((size_t)S.N * (band_halfwidth(S) + 1) + (size_t)S.N);
    return (size_t)S.dm.B * ((size_t)S.N * S.N + 2 * (size_t)S.N);
}

static int launch_kkt_dense(const NewtonDev& S, const KktArgs& K, double* ws, hipStream_t s, bool banded) {
    if (banded) {
        const int M = band_halfwidth(S) + 1;
        const size_t lds = ((size_t)(M + 1) * (M + 1) + 2 * (size_t)M) * sizeof(double);
        static LdsOptIn optin;
        if (lds_opt_in(optin, (const void*)kkt_banded_kernel, lds) != CIMPC_OK) return CIMPC_ERR_HIP;
        hipLaunchKernelGGL(kkt_banded_kernel, dim3(S.nb_launch), dim3(CIMPC_BANDED_THREADS), lds, s, S, K, ws);
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

}  // namespace cimpc
