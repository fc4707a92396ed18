// Horizon-level Newton kernels, one workgroup per rollout:
//   reset_kernel         reset!                    /root/reference/src/controller/newton.jl:130-167
//   resid_decide_kernel  residual! + gradient!     newton_residual.jl:113-138, 178-219, 256-281
//                        line search / accept / beta schedule   newton.jl:223-280
//                        update_traj! + update_theta!            newton_residual.jl:140-176
//   kkt_kernel           jacobian! + linear_solve!(solver, D, R, r)  newton_jacobian.jl:148-198,
//                        newton.jl:210-218, solved through the condensing of
//                        newton_structure_solver/methods.jl:386-557 (dual Schur complement
//                        Y = C P^-1 C^T + rho I, block-pentadiagonal, block Cholesky).
// The Newton iteration of every rollout advances in lock-step rounds driven by the host
// (cimpc_host.cpp); each rollout carries its own stage / alpha / beta, so rollouts that
// backtrack and rollouts that start their next Newton iteration share the same launches.
#include "newton_state.h"

namespace cimpc {

__device__ __forceinline__ void theta_update(const NewtonDev& S, const TrajDev& T, int b, int i,
                                             int tid, int nthreads) {
    // update_theta!(traj, i) (trajectory.jl:67-82): th_i = [q_i; q_{i+1}; u_i; w_i; mu; h]
    const cimpc_dims& m = S.dm;
    double* th = T.th + ((size_t)b * m.H + i) * S.nth;
    const double* q = T.q + ((size_t)b * (m.H + 2) + i) * m.nq;
    const double* u = T.u + ((size_t)b * m.H + i) * m.nu;
    const double* w = T.w + ((size_t)b * m.H + i) * m.nw;
    for (int k = tid; k < 2 * m.nq; k += nthreads) th[k] = q[k];   // q_i, q_{i+1} contiguous
    for (int k = tid; k < m.nu; k += nthreads) th[2 * m.nq + k] = u[k];
    for (int k = tid; k < m.nw; k += nthreads) th[2 * m.nq + m.nu + k] = w[k];
}

__device__ __forceinline__ void copy_n(double* dst, const double* src, int n, int tid, int nt) {
    for (int k = tid; k < n; k += nt) dst[k] = src[k];
}

__global__ __launch_bounds__(256) void reset_kernel(NewtonDev S, const double* q0,
                                                    const double* q1, int warm) {
    const cimpc_dims& m = S.dm;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int H = m.H;
    const size_t oq = (size_t)b * (H + 2) * m.nq;
    if (!warm) {
        for (int k = tid; k < H * S.nd; k += nt) {
            S.nu[(size_t)b * H * S.nd + k] = 0.0;
            S.nu_cand[(size_t)b * H * S.nd + k] = 0.0;
        }
        copy_n(S.traj.q + oq, S.ref.q + oq, (H + 2) * m.nq, tid, nt);
        copy_n(S.traj.u + (size_t)b * H * m.nu, S.ref.u + (size_t)b * H * m.nu, H * m.nu, tid, nt);
        copy_n(S.traj.w + (size_t)b * H * m.nw, S.ref.w + (size_t)b * H * m.nw, H * m.nw, tid, nt);
        copy_n(S.traj.g + (size_t)b * H * m.nc, S.ref.g + (size_t)b * H * m.nc, H * m.nc, tid, nt);
        copy_n(S.traj.b + (size_t)b * H * m.nb, S.ref.b + (size_t)b * H * m.nb, H * m.nb, tid, nt);
        copy_n(S.traj.th + (size_t)b * H * S.nth, S.ref.th + (size_t)b * H * S.nth, H * S.nth, tid, nt);
    }
    __syncthreads();
    for (int k = tid; k < m.nq; k += nt) {
        S.traj.q[oq + k] = q0[(size_t)b * m.nq + k];
        S.traj.q[oq + m.nq + k] = q1[(size_t)b * m.nq + k];
    }
    __syncthreads();
    theta_update(S, S.traj, b, 0, tid, nt);
    if (H > 1) theta_update(S, S.traj, b, 1, tid, nt);
    __syncthreads();
    // copy_traj!(traj_cand, traj)
    copy_n(S.cand.q + oq, S.traj.q + oq, (H + 2) * m.nq, tid, nt);
    copy_n(S.cand.u + (size_t)b * H * m.nu, S.traj.u + (size_t)b * H * m.nu, H * m.nu, tid, nt);
    copy_n(S.cand.w + (size_t)b * H * m.nw, S.traj.w + (size_t)b * H * m.nw, H * m.nw, tid, nt);
    copy_n(S.cand.g + (size_t)b * H * m.nc, S.traj.g + (size_t)b * H * m.nc, H * m.nc, tid, nt);
    copy_n(S.cand.b + (size_t)b * H * m.nb, S.traj.b + (size_t)b * H * m.nb, H * m.nb, tid, nt);
    copy_n(S.cand.th + (size_t)b * H * S.nth, S.traj.th + (size_t)b * H * S.nth, H * S.nth, tid, nt);
    if (warm) copy_n(S.nu_cand + (size_t)b * H * S.nd, S.nu + (size_t)b * H * S.nd, H * S.nd, tid, nt);
    if (tid == 0) {
        S.beta[b] = S.beta_init;
        S.stage[b] = STAGE_INIT;
        S.need_sweep[b] = 1;
        S.newton_l[b] = 0;
        S.alpha[b] = 1.0;
        S.ls_iter[b] = 0;
        S.r_norm[b] = 0.0;
        S.ro_sweeps[b] = 0;
        S.ro_ip_iters[b] = 0;
        S.ro_ip_fail[b] = 0;
    }
}

// x_cand = x - alpha*Delta for q_{t+2}, u_t, nu_t  (+ gamma, b in cf mode), then update_theta!
__device__ void apply_step(const NewtonDev& S, const TrajDev& dst, double* nu_dst, int b,
                           double alpha, int tid, int nt) {
    const cimpc_dims& m = S.dm;
    const int H = m.H, nq = m.nq, nu = m.nu, nr = S.nr, nd = S.nd;
    const double* D = S.delta + (size_t)b * S.N;
    const bool cf = m.mode == CIMPC_MODE_CONFIGURATIONFORCE;
    const int oq_in_block = cf ? nu + m.nc + m.nb : nu;
    for (int k = tid; k < H * nq; k += nt) {
        const int t = k / nq, c = k - t * nq;
        const size_t iq = ((size_t)b * (H + 2) + t + 2) * nq + c;
        dst.q[iq] = S.traj.q[iq] - alpha * D[t * nr + oq_in_block + c];
    }
    for (int k = tid; k < H * nu; k += nt) {
        const int t = k / nu, c = k - t * nu;
        const size_t iu = ((size_t)b * H + t) * nu + c;
        dst.u[iu] = S.traj.u[iu] - alpha * D[t * nr + c];
    }
    if (cf) {
        for (int k = tid; k < H * m.nc; k += nt) {
            const int t = k / m.nc, c = k - t * m.nc;
            const size_t ig = ((size_t)b * H + t) * m.nc + c;
            dst.g[ig] = S.traj.g[ig] - alpha * D[t * nr + nu + c];
        }
        for (int k = tid; k < H * m.nb; k += nt) {
            const int t = k / m.nb, c = k - t * m.nb;
            const size_t ib = ((size_t)b * H + t) * m.nb + c;
            dst.b[ib] = S.traj.b[ib] - alpha * D[t * nr + nu + m.nc + c];
        }
    }
    for (int k = tid; k < H * nd; k += nt) {
        const size_t in = (size_t)b * H * nd + k;
        nu_dst[in] = S.nu[in] - alpha * D[H * nr + k];
    }
    __syncthreads();
    for (int t = 0; t < H; ++t) theta_update(S, dst, b, t, tid, nt);
    __syncthreads();
}

__global__ __launch_bounds__(256) void resid_decide_kernel(NewtonDev S) {
    const cimpc_dims& m = S.dm;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    if (S.need_sweep[b] == 0) return;     // nothing was evaluated for this rollout
    const int H = m.H, nq = m.nq, nu = m.nu, nc = m.nc, nb = m.nb, nr = S.nr, nd = S.nd;
    const int nths = S.nths;
    const bool cf = m.mode == CIMPC_MODE_CONFIGURATIONFORCE;
    const int oq_in_block = cf ? nu + nc + nb : nu;
    __shared__ double red[256];
    __shared__ int action;
    double* r = S.res_cand + (size_t)b * S.N;
    const double* nuc = S.nu_cand + (size_t)b * H * nd;
    const double* dzb = S.dz + (size_t)b * H * nths * nd;
    double part = 0.0;
    // ---- residual! on (cand, nu_cand) -----------------------------------------------------
    for (int e = tid; e < S.N; e += nt) {
        double v = 0.0;
        if (e >= H * nr) {                       // rd[i] = d_i
            v = S.d[(size_t)b * H * nd + (e - H * nr)];
        } else {
            const int i = e / nr, c = e - i * nr;
            if (c < nu) {                         // u1[i]: R_i (u - u_ref) + du1_i^T nu_i
                const double* Rm = S.R + (size_t)i * nu * nu;
                const double* uu = S.cand.u + ((size_t)b * H + i) * nu;
                const double* ur = S.ref.u + ((size_t)b * H + i) * nu;
                for (int k = 0; k < nu; ++k) v = fma(Rm[c + k * nu], uu[k] - ur[k], v);
                const double* A0 = dzb + ((size_t)i * nths + 2 * nq + c) * nd;   // column c of du1_i
                double s = 0.0;
                for (int k = 0; k < nd; ++k) s = fma(A0[k], nuc[i * nd + k], s);
                v += s;
            } else if (c >= oq_in_block) {        // q2[i]
                const int cq = c - oq_in_block;
                const double* Qm = S.Q + (size_t)i * nq * nq;
                const double* qq = S.cand.q + ((size_t)b * (H + 2) + i + 2) * nq;
                const double* qr = S.ref.q + ((size_t)b * (H + 2) + i + 2) * nq;
                for (int k = 0; k < nq; ++k) v = fma(Qm[cq + k * nq], qq[k] - qr[k], v);
                v -= nuc[i * nd + cq];                                      // rI[i] -= nu_i
                if (i + 1 < H) {                                            // dq1_{i+1}^T nu_{i+1}
                    const double* A1 = dzb + ((size_t)(i + 1) * nths + nq + cq) * nd;
                    double s = 0.0;
                    for (int k = 0; k < nd; ++k) s = fma(A1[k], nuc[(i + 1) * nd + k], s);
                    v += s;
                }
                if (i + 2 < H) {                                            // dq0_{i+2}^T nu_{i+2}
                    const double* A2 = dzb + ((size_t)(i + 2) * nths + cq) * nd;
                    double s = 0.0;
                    for (int k = 0; k < nd; ++k) s = fma(A2[k], nuc[(i + 2) * nd + k], s);
                    v += s;
                }
            } else if (c < nu + nc) {             // gamma1[i] (cf)
                const int cg = c - nu;
                const double* Cm = S.Cg + (size_t)i * nc * nc;
                const double* gg = S.cand.g + ((size_t)b * H + i) * nc;
                const double* gr = S.ref.g + ((size_t)b * H + i) * nc;
                for (int k = 0; k < nc; ++k) v = fma(Cm[cg + k * nc], gg[k] - gr[k], v);
                v -= nuc[i * nd + nq + cg];
            } else {                              // b1[i] (cf)
                const int cb = c - nu - nc;
                const double* Cm = S.Cb + (size_t)i * nb * nb;
                const double* bb = S.cand.b + ((size_t)b * H + i) * nb;
                const double* br = S.ref.b + ((size_t)b * H + i) * nb;
                for (int k = 0; k < nb; ++k) v = fma(Cm[cb + k * nb], bb[k] - br[k], v);
                v -= nuc[i * nd + nq + nc + cb];
            }
        }
        r[e] = v;
        part += fabs(v);
    }
    red[tid] = part;
    __syncthreads();
    for (int s = nt / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    // ---- decision (newton.jl:198-280) ------------------------------------------------------
    if (tid == 0) {
        const double r_cand = red[0];
        S.r_cand[b] = r_cand;
        int act;   // 0 = accept initial evaluation, 1 = accept step, 2 = backtrack
        if (S.stage[b] == STAGE_INIT) {
            act = 0;
        } else {
            const double rn = S.r_norm[b];
            double a = S.alpha[b];
            if (r_cand * r_cand >= (1.0 - 0.001 * a) * rn * rn) {
                a *= 0.5;
                const int it = S.ls_iter[b] + 1;
                S.alpha[b] = a;
                S.ls_iter[b] = it;
                act = (it > 6) ? 1 : 2;
            } else {
                act = 1;
            }
        }
        action = act;
        // statistics of the sweep that was just consumed
        long long its = 0, fails = 0;
        for (int i = 0; i < H; ++i) {
            its += S.ip_iters[(size_t)b * H + i];
            fails += (S.ip_status[(size_t)b * H + i] == 0);
        }
        S.ro_sweeps[b] += 1;
        S.ro_ip_iters[b] += (int)its;
        S.ro_ip_fail[b] += (int)fails;
        atomicAdd((unsigned long long*)&S.stats[0], 1ull);
        atomicAdd((unsigned long long*)&S.stats[1], (unsigned long long)H);
        atomicAdd((unsigned long long*)&S.stats[2], (unsigned long long)its);
        atomicAdd((unsigned long long*)&S.stats[3], (unsigned long long)fails);
    }
    __syncthreads();
    const int act = action;
    if (act == 2) {                       // backtrack: new candidate with the halved alpha
        apply_step(S, S.cand, S.nu_cand, b, S.alpha[b], tid, nt);
        if (tid == 0) atomicAdd(&S.counters[0], 1);
        return;                           // stage stays LINESEARCH, need_sweep stays 1
    }
    if (act == 1) {                       // accept: traj <- traj - alpha*Delta (newton.jl:273)
        apply_step(S, S.traj, S.nu, b, S.alpha[b], tid, nt);
    }
    {   // res <- res_cand ; r_norm <- r_cand
        double* rr = S.res + (size_t)b * S.N;
        for (int e = tid; e < S.N; e += nt) rr[e] = r[e];
    }
    if (tid == 0) {
        const double rn = S.r_cand[b];
        S.r_norm[b] = rn;
        int l = S.newton_l[b];
        if (act == 1) {
            l += 1;
            S.newton_l[b] = l;
            const double be = S.beta[b];
            S.beta[b] = (S.ls_iter[b] > 6) ? fmin(be * 1.3, 1.0e2) : fmax(1.0e1, be / 1.3);
        }
        const bool done = (l >= S.max_iter) || (rn / (double)S.N < S.r_tol);
        S.need_sweep[b] = 0;
        if (done) {
            S.stage[b] = STAGE_DONE;
        } else {
            S.stage[b] = STAGE_KKT;
            atomicAdd(&S.counters[1], 1);
        }
    }
}

// -------------------------------------------------------------------------------------------
// KKT: condensed solve, one wavefront per rollout, blocks staged in LDS (v1: scalar loops).
// -------------------------------------------------------------------------------------------
// C[m x n] = alpha * A[m x k] * op(B) + betaC * C ;  op(B) = B^T (B is n x k) or B (k x n)
__device__ __forceinline__ void mm(double* C, int ldc, const double* A, int lda, const double* B,
                                   int ldb, int mrows, int ncols, int kk, bool transB,
                                   double alpha, double betaC, int lane) {
    for (int idx = lane; idx < mrows * ncols; idx += 64) {
        const int r = idx % mrows, c = idx / mrows;
        double s = 0.0;
        if (transB) for (int k = 0; k < kk; ++k) s = fma(A[r + k * lda], B[c + k * ldb], s);
        else for (int k = 0; k < kk; ++k) s = fma(A[r + k * lda], B[k + c * ldb], s);
        const double old = (betaC != 0.0) ? betaC * C[r + c * ldc] : 0.0;
        C[r + c * ldc] = alpha * s + old;
    }
}
// y[m] = alpha * op(A) x + betaY*y ; A is m x k (or k x m when trans)
__device__ __forceinline__ void mv(double* y, const double* A, int lda, const double* x, int mrows,
                                   int kk, bool transA, double alpha, double betaY, int lane) {
    for (int r = lane; r < mrows; r += 64) {
        double s = 0.0;
        if (transA) for (int k = 0; k < kk; ++k) s = fma(A[k + r * lda], x[k], s);
        else for (int k = 0; k < kk; ++k) s = fma(A[r + k * lda], x[k], s);
        const double old = (betaY != 0.0) ? betaY * y[r] : 0.0;
        y[r] = alpha * s + old;
    }
}

struct KktArgs {
    const double* r;     // [B][N] right-hand side
    double* delta;       // [B][N]
    const double* beta;  // [B] or null (then beta_scalar)
    double beta_scalar;
    const int* stage;    // only rollouts with stage == STAGE_KKT (null = all)
    int finish;          // 1: set alpha/ls_iter/cand/stage after the solve (newton loop)
};

__global__ __launch_bounds__(64) void kkt_kernel(NewtonDev S, KktArgs K) {
    const cimpc_dims& m = S.dm;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (K.stage != nullptr && K.stage[b] != STAGE_KKT) return;
    const int H = m.H, nq = m.nq, nu = m.nu, nr = S.nr, nd = S.nd, nths = S.nths;
    const int n2 = nd * nd;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    // LDS carve (doubles); nd == nq in :configuration mode
    double* A0 = sm;                 // nd x nu   du1_i
    double* A1 = A0 + nd * nu;       // nd x nq   dq1_i
    double* A2 = A1 + nd * nq;       // nd x nq   dq0_i
    double* A1p = A2 + nd * nq;      // dq1_{i-1}
    double* T0 = A1p + nd * nq;      // A0 * Rinv_i
    double* T1 = T0 + nd * nu;       // A1 * Qinv_{i-1}
    double* T2 = T1 + nd * nq;       // A2 * Qinv_{i-2}
    double* Y0 = T2 + nd * nq;       // nd x nd
    double* Y1 = Y0 + n2;
    double* Y2 = Y1 + n2;
    double* Lc = Y2 + n2;            // L0_i then scratch
    double* Li = Lc + n2;            // L0inv_i
    double* Li1 = Li + n2;           // L0inv_{i-1}
    double* Li2 = Li1 + n2;          // L0inv_{i-2}
    double* L1c = Li2 + n2;          // L1_i
    double* L1p = L1c + n2;          // L1_{i-1}
    double* L2c = L1p + n2;          // L2_i
    double* Mt = L2c + n2;           // temp
    double* bet = Mt + n2;           // nd
    double* yc = bet + nd;           // y_i
    double* y1 = yc + nd;            // y_{i-1}
    double* y2 = y1 + nd;            // y_{i-2}
    double* tv = y2 + nd;            // temp vec (max(nq,nu,nd))
    const double* rb = K.r + (size_t)b * S.N;
    const double beta = K.beta ? K.beta[b] : K.beta_scalar;
    const double rho = (double)H * beta * S.kappa;     // newton_jacobian.jl:169-186 quirk
    const double* dzb = S.dz + (size_t)b * H * nths * nd;
    double* ws = S.kkt_ws + (size_t)b * H * (3 * n2 + nd);
    const int oq = nu;   // offset of q2 inside a primal block (:configuration)

    for (int i = 0; i < H; ++i) {
        // ---- load sensitivities of step i --------------------------------------------------
        const double* dzi = dzb + (size_t)i * nths * nd;
        for (int k = lane; k < nd * nq; k += 64) { A2[k] = dzi[k]; A1[k] = dzi[nd * nq + k]; }
        for (int k = lane; k < nd * nu; k += 64) A0[k] = dzi[2 * nd * nq + k];
        __syncthreads();
        // ---- Y blocks and beta_i ------------------------------------------------------------
        mm(T0, nd, A0, nd, S.Rinv + (size_t)i * nu * nu, nu, nd, nu, nu, false, 1.0, 0.0, lane);
        if (i >= 1) mm(T1, nd, A1, nd, S.Qinv + (size_t)(i - 1) * nq * nq, nq, nd, nq, nq, false, 1.0, 0.0, lane);
        if (i >= 2) mm(T2, nd, A2, nd, S.Qinv + (size_t)(i - 2) * nq * nq, nq, nd, nq, nq, false, 1.0, 0.0, lane);
        __syncthreads();
        for (int k = lane; k < n2; k += 64) {
            const int r = k % nd, c = k / nd;
            Y0[k] = S.Qinv[(size_t)i * nq * nq + k] + ((r == c) ? rho : 0.0);
        }
        __syncthreads();
        mm(Y0, nd, T0, nd, A0, nd, nd, nd, nu, true, 1.0, 1.0, lane);
        __syncthreads();
        if (i >= 1) { mm(Y0, nd, T1, nd, A1, nd, nd, nd, nq, true, 1.0, 1.0, lane); __syncthreads(); }
        if (i >= 2) { mm(Y0, nd, T2, nd, A2, nd, nd, nd, nq, true, 1.0, 1.0, lane); __syncthreads(); }
        // beta_i = T0 rp_u[i] - Qinv_i rp_q[i] + T1 rp_q[i-1] + T2 rp_q[i-2] - rd[i]
        mv(bet, T0, nd, rb + i * nr, nd, nu, false, 1.0, 0.0, lane);
        __syncthreads();
        mv(bet, S.Qinv + (size_t)i * nq * nq, nq, rb + i * nr + oq, nd, nq, false, -1.0, 1.0, lane);
        __syncthreads();
        if (i >= 1) { mv(bet, T1, nd, rb + (i - 1) * nr + oq, nd, nq, false, 1.0, 1.0, lane); __syncthreads(); }
        if (i >= 2) { mv(bet, T2, nd, rb + (i - 2) * nr + oq, nd, nq, false, 1.0, 1.0, lane); __syncthreads(); }
        for (int k = lane; k < nd; k += 64) bet[k] -= rb[H * nr + i * nd + k];
        // Y1 = -T1 + T2 * dq1_{i-1}^T ;  Y2 = -T2
        if (i >= 1) {
            for (int k = lane; k < n2; k += 64) Y1[k] = -T1[k];
            __syncthreads();
            if (i >= 2) { mm(Y1, nd, T2, nd, A1p, nd, nd, nd, nq, true, 1.0, 1.0, lane); }
        }
        if (i >= 2) for (int k = lane; k < n2; k += 64) Y2[k] = -T2[k];
        __syncthreads();
        // ---- block Cholesky step ------------------------------------------------------------
        // L2_i = Y2 * L0inv_{i-2}^T ;  L1_i = (Y1 - L2_i L1_{i-1}^T) * L0inv_{i-1}^T
        if (i >= 2) { mm(L2c, nd, Y2, nd, Li2, nd, nd, nd, nd, true, 1.0, 0.0, lane); __syncthreads(); }
        if (i >= 1) {
            if (i >= 2) { mm(Y1, nd, L2c, nd, L1p, nd, nd, nd, nd, true, -1.0, 1.0, lane); __syncthreads(); }
            mm(L1c, nd, Y1, nd, Li1, nd, nd, nd, nd, true, 1.0, 0.0, lane);
            __syncthreads();
            mm(Y0, nd, L1c, nd, L1c, nd, nd, nd, nd, true, -1.0, 1.0, lane);
            __syncthreads();
        }
        if (i >= 2) { mm(Y0, nd, L2c, nd, L2c, nd, nd, nd, nd, true, -1.0, 1.0, lane); __syncthreads(); }
        // chol(Y0) -> Lc (lower), right-looking
        for (int k = lane; k < n2; k += 64) Lc[k] = Y0[k];
        __syncthreads();
        for (int k = 0; k < nd; ++k) {
            const double dkk = sqrt(Lc[k + k * nd]);
            __syncthreads();
            for (int r = k + lane; r < nd; r += 64) Lc[r + k * nd] = (r == k) ? dkk : Lc[r + k * nd] / dkk;
            __syncthreads();
            const int rem = nd - k - 1;
            for (int idx = lane; idx < rem * rem; idx += 64) {
                const int r = k + 1 + idx % rem, c = k + 1 + idx / rem;
                if (r >= c) Lc[r + c * nd] -= Lc[r + k * nd] * Lc[c + k * nd];
            }
            __syncthreads();
        }
        // Li = inv(Lc) (lower triangular), one column per lane
        for (int c = lane; c < nd; c += 64) {
            for (int r = 0; r < nd; ++r) {
                if (r < c) { Li[r + c * nd] = 0.0; continue; }
                double s = (r == c) ? 1.0 : 0.0;
                for (int k = c; k < r; ++k) s -= Lc[r + k * nd] * Li[k + c * nd];
                Li[r + c * nd] = s / Lc[r + r * nd];
            }
        }
        __syncthreads();
        // y_i = Li * (beta_i - L1_i y_{i-1} - L2_i y_{i-2})
        if (i >= 1) { mv(bet, L1c, nd, y1, nd, nd, false, -1.0, 1.0, lane); __syncthreads(); }
        if (i >= 2) { mv(bet, L2c, nd, y2, nd, nd, false, -1.0, 1.0, lane); __syncthreads(); }
        mv(yc, Li, nd, bet, nd, nd, false, 1.0, 0.0, lane);
        __syncthreads();
        // ---- spill factors for the backward pass, rotate the window ------------------------
        double* wsi = ws + (size_t)i * (3 * n2 + nd);
        for (int k = lane; k < n2; k += 64) {
            wsi[k] = L1c[k];
            wsi[n2 + k] = L2c[k];
            wsi[2 * n2 + k] = Li[k];
            Li2[k] = Li1[k];
            Li1[k] = Li[k];
            L1p[k] = L1c[k];
        }
        for (int k = lane; k < nd * nq; k += 64) A1p[k] = A1[k];
        for (int k = lane; k < nd; k += 64) { wsi[3 * n2 + k] = yc[k]; y2[k] = y1[k]; y1[k] = yc[k]; }
        __syncthreads();
    }
    // ---- backward substitution: dnu_i = Li_i^T (y_i - L1_{i+1}^T dnu_{i+1} - L2_{i+2}^T dnu_{i+2})
    double* D = K.delta + (size_t)b * S.N;
    double* dn1 = y1;   // dnu_{i+1}
    double* dn2 = y2;   // dnu_{i+2}
    for (int i = H - 1; i >= 0; --i) {
        const double* wsi = ws + (size_t)i * (3 * n2 + nd);
        for (int k = lane; k < nd; k += 64) bet[k] = wsi[3 * n2 + k];
        __syncthreads();
        if (i + 1 < H) { mv(bet, ws + (size_t)(i + 1) * (3 * n2 + nd), nd, dn1, nd, nd, true, -1.0, 1.0, lane); __syncthreads(); }
        if (i + 2 < H) { mv(bet, ws + (size_t)(i + 2) * (3 * n2 + nd) + n2, nd, dn2, nd, nd, true, -1.0, 1.0, lane); __syncthreads(); }
        mv(yc, wsi + 2 * n2, nd, bet, nd, nd, true, 1.0, 0.0, lane);
        __syncthreads();
        for (int k = lane; k < nd; k += 64) { D[H * nr + i * nd + k] = yc[k]; dn2[k] = dn1[k]; dn1[k] = yc[k]; }
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    // ---- primal recovery: Delta_x = P^-1 (r_p - C^T dnu) ----------------------------------
    const double* dn = D + H * nr;
    for (int i = 0; i < H; ++i) {
        const double* dzi = dzb + (size_t)i * nths * nd;
        // u block
        for (int c = lane; c < nu; c += 64) {
            double s = 0.0;
            const double* a0 = dzi + (size_t)(2 * nq + c) * nd;
            for (int k = 0; k < nd; ++k) s = fma(a0[k], dn[i * nd + k], s);
            tv[c] = rb[i * nr + c] - s;
        }
        __syncthreads();
        mv(D + i * nr, S.Rinv + (size_t)i * nu * nu, nu, tv, nu, nu, false, 1.0, 0.0, lane);
        __syncthreads();
        // q block: r_q - (-dnu_i + dq1_{i+1}^T dnu_{i+1} + dq0_{i+2}^T dnu_{i+2})
        for (int c = lane; c < nq; c += 64) {
            double s = -dn[i * nd + c];
            if (i + 1 < H) {
                const double* a1 = dzb + ((size_t)(i + 1) * nths + nq + c) * nd;
                double t = 0.0;
                for (int k = 0; k < nd; ++k) t = fma(a1[k], dn[(i + 1) * nd + k], t);
                s += t;
            }
            if (i + 2 < H) {
                const double* a2 = dzb + ((size_t)(i + 2) * nths + c) * nd;
                double t = 0.0;
                for (int k = 0; k < nd; ++k) t = fma(a2[k], dn[(i + 2) * nd + k], t);
                s += t;
            }
            tv[c] = rb[i * nr + oq + c] - s;
        }
        __syncthreads();
        mv(D + i * nr + oq, S.Qinv + (size_t)i * nq * nq, nq, tv, nq, nq, false, 1.0, 0.0, lane);
        __syncthreads();
    }
    if (K.finish) {
        __threadfence_block();
        __syncthreads();
        // line search start (newton.jl:223-228): alpha = 1, candidate = traj - Delta
        apply_step(S, S.cand, S.nu_cand, b, 1.0, lane, 64);
        if (lane == 0) {
            S.alpha[b] = 1.0;
            S.ls_iter[b] = 0;
            S.stage[b] = STAGE_LINESEARCH;
            S.need_sweep[b] = 1;
            atomicAdd(&S.counters[0], 1);
        }
    }
}

static size_t kkt_lds_bytes(const NewtonDev& S) {
    const int nd = S.nd, nq = S.dm.nq, nu = S.dm.nu;
    const int mx = nd > nq ? (nd > nu ? nd : nu) : (nq > nu ? nq : nu);
    size_t dbl = 2 * (size_t)nd * nu + 5 * (size_t)nd * nq + 11 * (size_t)nd * nd + 4 * nd + mx;
    return dbl * sizeof(double);
}

int launch_reset(const NewtonDev& S, const double* q0, const double* q1, int warm, hipStream_t s) {
    hipLaunchKernelGGL(reset_kernel, dim3(S.dm.B), dim3(256), 0, s, S, q0, q1, warm);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}
int launch_resid_decide(const NewtonDev& S, hipStream_t s) {
    hipLaunchKernelGGL(resid_decide_kernel, dim3(S.dm.B), dim3(256), 0, s, S);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}
int launch_kkt(const NewtonDev& S, hipStream_t s) {
    KktArgs K{S.res, S.delta, S.beta, 0.0, S.stage, 1};
    hipLaunchKernelGGL(kkt_kernel, dim3(S.dm.B), dim3(64), kkt_lds_bytes(S), s, S, K);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}
int launch_kkt_raw(const NewtonDev& S, const double* r_dev, double beta, double* delta_dev,
                   hipStream_t s) {
    KktArgs K{r_dev, delta_dev, nullptr, beta, nullptr, 0};
    hipLaunchKernelGGL(kkt_kernel, dim3(S.dm.B), dim3(64), kkt_lds_bytes(S), s, S, K);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}

}  // namespace cimpc
