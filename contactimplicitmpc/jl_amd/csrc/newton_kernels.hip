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
#include <cstdlib>
#ifdef CIMPC_KKT_PROF
#define CIMPC_KKT_WPROF      // per-wave clocks of the pipelined KKT kernel: this translation unit only (the asynchronous kernel's does not survive them)
#endif
#include "newton_impl.h"

namespace cimpc {

__global__ __launch_bounds__(256) void reset_kernel(NewtonDev S, const double* q0,
                                                    const double* q1, int warm) {
    const cimpc_dims& m = S.dm;
    const int b = blockIdx.x + S.b0, tid = threadIdx.x, nt = blockDim.x;
    const int H = m.H;
    const size_t oq = (size_t)b * (H + 2) * m.nq;
    const size_t sb0 = (size_t)b * CS;
    if (!warm) {
        for (int k = tid; k < H * S.nd; k += nt) {
            S.nu[(size_t)b * H * S.nd + k] = 0.0;
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
    // copy_traj!(traj_cand, traj): evaluation slot 0 <- traj, nu_cand <- nu
    copy_n(S.cand.q + sb0 * (H + 2) * m.nq, S.traj.q + oq, (H + 2) * m.nq, tid, nt);
    copy_n(S.cand.u + sb0 * H * m.nu, S.traj.u + (size_t)b * H * m.nu, H * m.nu, tid, nt);
    copy_n(S.cand.w + sb0 * H * m.nw, S.traj.w + (size_t)b * H * m.nw, H * m.nw, tid, nt);
    copy_n(S.cand.g + sb0 * H * m.nc, S.traj.g + (size_t)b * H * m.nc, H * m.nc, tid, nt);
    copy_n(S.cand.b + sb0 * H * m.nb, S.traj.b + (size_t)b * H * m.nb, H * m.nb, tid, nt);
    copy_n(S.cand.th + sb0 * H * S.nth, S.traj.th + (size_t)b * H * S.nth, H * S.nth, tid, nt);
    copy_n(S.nu_cand + sb0 * H * S.nd, S.nu + (size_t)b * H * S.nd, H * S.nd, tid, nt);
    for (int c = 1; c < CS; ++c)       // w (disturbance) never changes during a solve
        copy_n(S.cand.w + (sb0 + c) * H * m.nw, S.traj.w + (size_t)b * H * m.nw, H * m.nw, tid, nt);
    if (tid == 0) {
        S.beta[b] = S.beta_init;
        S.stage[b] = STAGE_INIT;
        for (int c = 1; c < CS; ++c) S.need_sweep[sb0 + c] = 0;
        S.cur_slot[b] = 0;
        S.newton_l[b] = 0;
        S.alpha[b] = 1.0;
        S.ls_iter[b] = 0;
        S.r_norm[b] = 0.0;
        S.ro_sweeps[b] = 0;
        S.ro_ip_iters[b] = 0;
        S.ro_ip_fail[b] = 0;
    }
    for (int k = tid; k < CS * H; k += nt) S.pflag[sb0 * H + k] = 0;
    __syncthreads();
    if (S.A.on) enqueue_eval_async<BlockSync>(S, sb0, 0, 1, b, tid, nt);
    else enqueue_eval(S, sb0, b, S.WQ.par, tid, nt, b - S.b0);      // round 0 evaluates slot 0 of every rollout: list entry b
}

template <int NQ, int NU>
static int launch_kkt_t(const NewtonDev& S, const KktArgs& K, hipStream_t s, bool f32 = false) {
    const bool force_scalar = S.kkt_scalar != 0;      // (CIMPC_KKT_SCALAR, read at cimpc_create)
    if (NQ <= 24 && NU <= 24 && S.dm.H <= kkt_max_h<NQ, NU>() && !force_scalar) {   // dnu / recovery staging bounds H
        const size_t lds = (size_t)kkt_lds_doubles<NQ, NU, 1>() * sizeof(double);
        if (f32) {      // block products on the fp32 MFMA (launch_kkt_mixed refines the result in fp64)
            static LdsOptIn optin32;
            if (lds_opt_in(optin32, (const void*)kkt_kernel<NQ, NU, true>, lds) != CIMPC_OK) return CIMPC_ERR_HIP;
            hipLaunchKernelGGL((kkt_kernel<NQ, NU, true>), dim3(S.nb_launch), dim3(64), lds, s, S, K);
            return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
        }
        static LdsOptIn optin;
        if (lds_opt_in(optin, (const void*)kkt_kernel<NQ, NU>, lds) != CIMPC_OK) return CIMPC_ERR_HIP;
        hipLaunchKernelGGL((kkt_kernel<NQ, NU>), dim3(S.nb_launch), dim3(64), lds, s, S, K);
    } else {
        if (f32) return CIMPC_ERR_INVALID;
        constexpr int LD = ((NQ > NU ? NQ : NU) + 3) & ~3;
        const size_t lds = (size_t)(KKT_TILES * LD * LD + 10 * LD) * sizeof(double);
        static LdsOptIn optin;
        if (lds_opt_in(optin, (const void*)kkt_kernel_scalar<NQ, NU>, lds) != CIMPC_OK) return CIMPC_ERR_HIP;
        hipLaunchKernelGGL((kkt_kernel_scalar<NQ, NU>), dim3(S.nb_launch), dim3(64), lds, s, S, K);
    }
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}

// (nq, nu) pairs of the built models (the horizon-level kernels depend on these two sizes only): pushbot / particle_2D,
// hopper_2D, quadruped, flamingo, centroidal_quadruped (= point_foot_quadruped, centroidal_quadruped_box), hopper_3D,
// walledcartpole, particle
#define CIMPC_NQNU(X) X(2, 2) X(4, 2) X(11, 8) X(9, 6) X(18, 12) X(7, 3) X(4, 1) X(3, 3)

static int launch_kkt_any(const NewtonDev& S, const KktArgs& K, hipStream_t s, bool f32 = false) {
    if (S.dm.mode != CIMPC_MODE_CONFIGURATION) return CIMPC_ERR_INVALID;
    const int nq = S.dm.nq, nu = S.dm.nu;
#define X(q, u) if (nq == q && nu == u) return launch_kkt_t<q, u>(S, K, s, f32);
    CIMPC_NQNU(X)
#undef X
    return CIMPC_ERR_INVALID;
}

// ---------------------------------------------------------------------------------------------------------------------
// Mixed-precision condensed KKT solve (BASELINE configs[4]: "fp32 mixed-precision Schur GEMM on MFMA" + fp64 refinement).
//   P0:  D   = solve32(r)                       block products of the recursion on v_mfma_f32_16x16x4_f32 (kkt_body<.., F32>)
//   Ck:  res = r - R D  in fp64, matrix-free    R = [P C^T; C -rho I] applied from the objective blocks and the accepted
//                                               sensitivities (the same data the solve condenses - newton_jacobian.jl:148-198)
//        |res|_inf <= 1e-9 max(1, |r|_inf) ?  done (Newton loop: the line search starts) : stays flagged
//   Pk:  D  += solve32(res)                     at most four corrections (contraction ~ cond(Y) 2^-24 per pass: dual Schur
//                                               complements up to cond ~ 1e5 reach 1e-9; the cold start's beta = 1e-5 - rho = 1e-7,
//                                               cond > 1e8 - does not and takes the fp64 solve)
//   F:   D   = solve64(r)                       fp64 fallback for whatever is still flagged (ill-conditioned Schur blocks,
//                                               a non-positive fp32 pivot -> NaN -> fails the check: test/solver/schur.jl:19-62)
// Every kernel covers all rollouts and leaves at once where it has nothing to do (flag array), so the host enqueues the
// whole sequence without reading anything back.
// ---------------------------------------------------------------------------------------------------------------------
// (R x)[e] for rollout data at dzb / x; :configuration mode, TrackingObjective (dense Q_i, R_i allowed)
__device__ __forceinline__ double kkt_apply_row(const NewtonDev& S, const double* dzb, const double* x, double rho, int e) {
    const int H = S.dm.H, nq = S.dm.nq, nu = S.dm.nu, nr = nq + nu, nd = nq, blk = (2 * nq + nu) * nd;
    const double* nuv = x + (size_t)H * nr;
    if (e >= H * nr) {                                   // dual row (i, k):  C x - rho nu
        const int i = (e - H * nr) / nd, k = (e - H * nr) - i * nd;
        const double* dz = dzb + (size_t)i * blk;
        double s = -x[(size_t)i * nr + nu + k] - rho * nuv[(size_t)i * nd + k];
        for (int c = 0; c < nu; ++c) s = fma(dz[(size_t)(2 * nq + c) * nd + k], x[(size_t)i * nr + c], s);
        if (i >= 1) for (int c = 0; c < nq; ++c) s = fma(dz[(size_t)(nq + c) * nd + k], x[(size_t)(i - 1) * nr + nu + c], s);
        if (i >= 2) for (int c = 0; c < nq; ++c) s = fma(dz[(size_t)c * nd + k], x[(size_t)(i - 2) * nr + nu + c], s);
        return s;
    }
    const int i = e / nr, c = e - i * nr;
    double s = 0.0;
    if (c < nu) {                                        // u row: R_i u + du1_i^T nu_i
        const double* Rm = S.R + (size_t)i * nu * nu;
        for (int k = 0; k < nu; ++k) s = fma(Rm[c + k * nu], x[(size_t)i * nr + k], s);
        const double* A0 = dzb + (size_t)i * blk + (size_t)(2 * nq + c) * nd;
        for (int k = 0; k < nd; ++k) s = fma(A0[k], nuv[(size_t)i * nd + k], s);
        return s;
    }
    const int cq = c - nu;                               // q row: Q_i q - nu_i + dq1_{i+1}^T nu_{i+1} + dq0_{i+2}^T nu_{i+2}
    const double* Qm = S.Q + (size_t)i * nq * nq;
    for (int k = 0; k < nq; ++k) s = fma(Qm[cq + k * nq], x[(size_t)i * nr + nu + k], s);
    s -= nuv[(size_t)i * nd + cq];
    if (i + 1 < H) {
        const double* A1 = dzb + (size_t)(i + 1) * blk + (size_t)(nq + cq) * nd;
        for (int k = 0; k < nd; ++k) s = fma(A1[k], nuv[(size_t)(i + 1) * nd + k], s);
    }
    if (i + 2 < H) {
        const double* A2 = dzb + (size_t)(i + 2) * blk + (size_t)cq * nd;
        for (int k = 0; k < nd; ++k) s = fma(A2[k], nuv[(size_t)(i + 2) * nd + k], s);
    }
    return s;
}
struct MixArgs {
    int* flag;          // [B] 1 = still to be solved
    double* corr;       // [B][N] correction of the running pass (null in the first check)
    double* res;        // [B][N] residual out
    double tol;
    int last;           // 1: a rollout that fails this check keeps its flag for the fp64 fallback (else for another correction)
    int* n_fallback;    // device counter: rollouts handed to the fp64 solve (statistics)
};
__global__ __launch_bounds__(256) void kkt_mixed_init_kernel(NewtonDev S, KktArgs K, int* flag) {
    const int b = blockIdx.x * 256 + threadIdx.x + S.b0;
    if (b < S.b0 + S.nb_launch) flag[b] = (K.stage == nullptr || K.stage[b] == STAGE_KKT) ? 1 : 0;
}
// D += corr ; res = r - R D ; verdict.  One workgroup per rollout.
__global__ __launch_bounds__(256) void kkt_mixed_check_kernel(NewtonDev S, KktArgs K, MixArgs M) {
    const int b = blockIdx.x + S.b0, tid = threadIdx.x, nt = blockDim.x;
    if (M.flag[b] == 0) return;
    __shared__ double red[2 * 256];
    const int N = S.N, H = S.dm.H;
    double* D = K.delta + (size_t)b * N;
    const double* r = K.r + (size_t)b * N;
    if (M.corr != nullptr) {
        const double* c = M.corr + (size_t)b * N;
        for (int e = tid; e < N; e += nt) D[e] += c[e];
        __syncthreads();
    }
    const double beta = K.beta ? K.beta[b] : K.beta_scalar;
    const double rho = (double)H * beta * S.kappa;
    const double* dzb = kkt_dz(S, K, b, H, S.nths, S.nd);
    double* res = M.res + (size_t)b * N;
    double mr = 0.0, mres = 0.0;
    for (int e = tid; e < N; e += nt) {
        const double v = r[e] - kkt_apply_row(S, dzb, D, rho, e);
        res[e] = v;
        mr = fmax(mr, fabs(r[e]));
        mres = (v == v) ? fmax(mres, fabs(v)) : HUGE_VAL;       // NaN (failed fp32 pivot) never passes
    }
    red[tid] = mr; red[256 + tid] = mres;
    __syncthreads();
    for (int st = nt / 2; st > 0; st >>= 1) {
        if (tid < st) { red[tid] = fmax(red[tid], red[tid + st]); red[256 + tid] = fmax(red[256 + tid], red[256 + tid + st]); }
        __syncthreads();
    }
    const bool ok = red[256] <= M.tol * fmax(1.0, red[0]);
    __syncthreads();
    if (ok) {
        if (tid == 0) M.flag[b] = 0;
        if (K.finish) {
            __threadfence_block();
            start_line_search<BlockSync>(S, b, 1, tid, nt);
        }
    } else if (M.last && tid == 0 && M.n_fallback != nullptr) {
        atomicAdd(M.n_fallback, 1);
    }
}
size_t kkt_mixed_workspace_doubles(const NewtonDev& S) { return 2 * (size_t)S.dm.B * S.N + (size_t)S.dm.B; }
bool kkt_mixed_available(const NewtonDev& S) {
    return S.dm.mode == CIMPC_MODE_CONFIGURATION && S.V == nullptr && S.dm.nq <= 24 && S.dm.nu <= 24 && S.dm.H <= 96;
}
int launch_kkt_mixed(const NewtonDev& S, const KktArgs& K0, double* ws, int* n_fallback, hipStream_t s) {
    if (!kkt_mixed_available(S)) return CIMPC_ERR_INVALID;
    const size_t BN = (size_t)S.dm.B * S.N;
    double* corr = ws; double* res = ws + BN;
    int* flag = reinterpret_cast<int*>(ws + 2 * BN);
    hipLaunchKernelGGL(kkt_mixed_init_kernel, dim3((S.nb_launch + 255) / 256), dim3(256), 0, s, S, K0, flag);
    KktArgs P = K0; P.finish = 0; P.only_flag = flag;
    int rc = launch_kkt_any(S, P, s, true);                                   // P0
    if (rc != CIMPC_OK) return rc;
    constexpr int CORRECTIONS = 4;
    for (int k = 0; k <= CORRECTIONS; ++k) {
        MixArgs M{flag, k == 0 ? nullptr : corr, res, 1.0e-9, k == CORRECTIONS ? 1 : 0, n_fallback};
        hipLaunchKernelGGL(kkt_mixed_check_kernel, dim3(S.nb_launch), dim3(256), 0, s, S, K0, M);       // Ck
        if (k == CORRECTIONS) break;
        KktArgs Pk = P; Pk.r = res; Pk.delta = corr;
        rc = launch_kkt_any(S, Pk, s, true);                                  // P(k+1)
        if (rc != CIMPC_OK) return rc;
    }
    KktArgs F = K0; F.only_flag = flag;                                        // fp64 fallback (keeps K0.finish)
    rc = launch_kkt_any(S, F, s, false);
    if (rc != CIMPC_OK) return rc;
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}

#ifdef CIMPC_RESID_PROF
extern "C" int cimpc_debug_resid_prof(unsigned long long* out32) {      // diagnostic builds: read and clear the decision-stage clocks (64 words)
    unsigned long long z[64] = {0};
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(cimpc::g_resid_prof), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(cimpc::g_resid_prof), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif
__global__ __launch_bounds__(64) void enqueue_all_kernel(NewtonDev S) {
    const int b = blockIdx.x + S.b0;
    enqueue_eval(S, (size_t)b * CS, b, S.WQ.par, threadIdx.x, 64, b - S.b0);
    if (threadIdx.x == 0) S.cur_slot[b] = 0;
    for (int k = threadIdx.x; k < S.dm.H; k += 64) S.pflag[(size_t)b * CS * S.dm.H + k] = 0;
}
int launch_enqueue_all(const NewtonDev& S, hipStream_t s) {
    hipLaunchKernelGGL(enqueue_all_kernel, dim3(S.nb_launch), dim3(64), 0, s, S);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}
// Hybrid solve: the lock-step rounds hand the still-active rollouts over to the asynchronous kernel
// (newton_async_impl.h) once few are left.  Called between two rounds.  The lock-step queue of the
// NEXT round (LQ, parity LQ.par) already holds everything that is still to be solved - evaluations
// requested by the residual / KKT kernels and solves parked by the last sweep - so it is copied
// entry by entry into the live queues; per rollout only the bookkeeping of the asynchronous stages
// is set up: finished -> counted, waiting for KKT -> KKT job, in a line search -> number of
// evaluation slots that are not complete yet.
__global__ __launch_bounds__(64) void async_handoff_kernel(NewtonDev S, IpQueues LQ) {
    const int i = blockIdx.x, tid = threadIdx.x;
    if (i < S.WQ.K) {      // knot i: copy the pending entries
        const int n = *qcount(LQ, LQ.par, i);
        const int* src = LQ.items + ((size_t)LQ.par * LQ.K + i) * LQ.cap;
        int* dst = S.WQ.items + (size_t)i * S.WQ.cap;
        for (int k = tid; k < n; k += 64) dst[k] = src[k];
        if (tid == 0) *qcount(S.WQ, 0, i) = n;
    }
    if (i < S.dm.B && tid == 0) {
        const int b = i, stage = S.stage[b];
        if (stage == STAGE_DONE) {
            atomicAdd(S.A.n_done, 1);
        } else if (stage == STAGE_KKT) {
            push_kkt_job(S.A, b);      // (the wake-up words it bumps are read by nobody yet: the persistent kernel starts behind this one)
        } else {
            const size_t sb0 = (size_t)b * CS;
            int n = 0;
            for (int c = 0; c < CS; ++c) n += (S.need_sweep[sb0 + c] != 0 && S.WQ.done_count[sb0 + c] < S.dm.H);
            S.A.evals_left[b] = n;
        }
    }
}
int launch_async_handoff(const NewtonDev& S, const IpQueues& LQ, hipStream_t s) {
    const int grid = S.dm.B > S.WQ.K ? S.dm.B : S.WQ.K;
    hipLaunchKernelGGL(async_handoff_kernel, dim3(grid), dim3(64), 0, s, S, LQ);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}
// MPC-loop glue between two solves, per rollout: rot_n_stride! (mpc_utils.jl:1-101) on the controller's
// copy of the reference trajectory - every array moves one step to the left, the first entry goes last
// (rotate!), the last two configurations repeat the first two shifted by `stride` and the theta slices
// that read them are refreshed (mpc_stride!) - and update_window! (policy.jl:162-171).
__global__ __launch_bounds__(64) void mpc_advance_kernel(NewtonDev S, int* window, const double* stride, int H_ref) {
    const cimpc_dims& m = S.dm;
    const int b = blockIdx.x, c = threadIdx.x, H = m.H, nq = m.nq, nth = S.nth;
    auto rot = [&](double* a, int rows, int n) {          // column c of a [rows][n] array, in place
        if (c >= n) return;
        const double first = a[c];
        for (int t = 0; t + 1 < rows; ++t) a[(size_t)t * n + c] = a[(size_t)(t + 1) * n + c];
        a[(size_t)(rows - 1) * n + c] = first;
    };
    double* q = S.ref.q + (size_t)b * (H + 2) * nq;
    double* th = S.ref.th + (size_t)b * H * nth;
    rot(q, H + 2, nq);
    rot(S.ref.u + (size_t)b * H * m.nu, H, m.nu);
    rot(S.ref.w + (size_t)b * H * m.nw, H, m.nw);
    rot(S.ref.g + (size_t)b * H * m.nc, H, m.nc);
    rot(S.ref.b + (size_t)b * H * m.nb, H, m.nb);
    rot(th, H, nth);
    __syncthreads();
    if (c < nq) {                                          // mpc_stride!: q_{H+1}, q_{H+2} (1-based)
        q[(size_t)H * nq + c] = q[c] + stride[c];
        q[(size_t)(H + 1) * nq + c] = q[nq + c] + stride[c];
    }
    __syncthreads();
    if (c < nq) {                                          // update_theta! of steps H-2, H-1, H (1-based): q0, q1 slices
        for (int tau = H - 2; tau <= H; ++tau) {
            if (tau < 1) continue;
            th[(size_t)(tau - 1) * nth + c] = q[(size_t)(tau - 1) * nq + c];
            th[(size_t)(tau - 1) * nth + nq + c] = q[(size_t)tau * nq + c];
        }
    }
    for (int i = c; i < H + 2; i += 64) {                  // update_window! (0-based knots)
        int* w = window + (size_t)b * (H + 2) + i;
        *w = (*w + 1) % H_ref;
    }
}
// The same glue when the controller's FULL reference trajectory is resident (cimpc_set_gait): the state after k
// applications of rot_n_stride! has a closed form.  Entry i holds "absolute" knot a = k + i.  u, w, gamma, b, theta
// only rotate: row a mod H_ref.  Configurations: mpc_stride! overwrites q_{H+1}, q_{H+2} with q_1 + stride, q_2 + stride
// after every rotation (1-based), so q(a) = q(a - H_ref) + stride for a > H_ref while q_{H+1} of the file itself
// survives its lap through the horizon: with a - 1 = lap * H_ref + r,  q(a) = gait_q[r + 1] + lap * stride  (a >= 1),
// q(0) = gait_q[0]; the only other file value is gait_q[H_ref + 1], visible at k = 0.  theta's q0 / q1 slices equal
// the configurations of their step (update_theta! after every overwrite).  Rebuilds a rollout's window-length
// reference and window after adding `advance` to its step counter.
__global__ __launch_bounds__(128) void gait_window_kernel(NewtonDev S, GaitDev G, int* window, int advance) {
    const cimpc_dims& m = S.dm;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x, H = m.H, nq = m.nq, nth = S.nth, Hr = G.H_ref;
    const int k = G.phase[b] + advance;
    __syncthreads();
    if (tid == 0) G.phase[b] = k;
    double* q = S.ref.q + (size_t)b * (H + 2) * nq;
    double* th = S.ref.th + (size_t)b * H * nth;
    for (int e = tid; e < (H + 2) * nq; e += nt) {
        const int i = e / nq, c = e - i * nq, a = k + i;
        const int lap = a >= 1 ? (a - 1) / Hr : 0, j = a >= 1 ? (a - 1) - lap * Hr + 1 : 0;
        q[e] = (k == 0 && a == Hr + 1) ? G.q[(size_t)a * nq + c]
                                       : (lap ? G.q[(size_t)j * nq + c] + (double)lap * G.stride[c] : G.q[(size_t)j * nq + c]);
    }
    auto rows = [&](double* dst, const double* src, int n) {
        for (int e = tid; e < H * n; e += nt) {
            const int i = e / n, c = e - i * n;
            dst[e] = src[(size_t)((k + i) % Hr) * n + c];
        }
    };
    rows(S.ref.u + (size_t)b * H * m.nu, G.u, m.nu);
    rows(S.ref.w + (size_t)b * H * m.nw, G.w, m.nw);
    rows(S.ref.g + (size_t)b * H * m.nc, G.g, m.nc);
    rows(S.ref.b + (size_t)b * H * m.nb, G.b, m.nb);
    rows(th, G.th, nth);
    __syncthreads();
    for (int e = tid; e < H * 2 * nq; e += nt) {           // update_theta!: q0, q1 slices of every step
        const int i = e / (2 * nq), c = e - i * 2 * nq;
        th[(size_t)i * nth + c] = q[(size_t)i * nq + c];   // c < nq: q_i, else q_{i+1} (contiguous in q)
    }
    for (int i = tid; i < H + 2; i += nt) window[(size_t)b * (H + 2) + i] = (k + i) % Hr;
}
int launch_gait_window(const NewtonDev& S, const GaitDev& G, int* window, int advance, hipStream_t s) {
    hipLaunchKernelGGL(gait_window_kernel, dim3(S.dm.B), dim3(128), 0, s, S, G, window, advance);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}
int launch_mpc_advance(const NewtonDev& S, int* window, const double* stride, int H_ref, hipStream_t s) {
    hipLaunchKernelGGL(mpc_advance_kernel, dim3(S.dm.B), dim3(64), 0, s, S, window, stride, H_ref);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}
// ---------------------------------------------------------------------------------------------------------
// Sensitivity memory follows the KNOTS.  The reference owns one interior-point solver per reference knot (im_traj.ip[t],
// implicit_dynamics.jl:71-86, 169-176): ip[t].dz is overwritten by every successful solve of knot t and left alone by failed
// ones - across candidate evaluations, Newton solves and the window shifts of the MPC loop.  On the device that memory is
// kept per horizon STEP (dz_good: the accepted evaluation of the Newton loop; slot 0 of S.dz: the B3 seam), which is the
// same thing as long as the window stands still.  When the window changes the blocks are re-keyed through a per-knot
// archive: dir 0  knot[b][window[b][i]] <- step(b, i)   (with the window in force),
//          dir 1  step(b, i) <- knot[b][window[b][i]]   (with the new window).
// One workgroup per (rollout, step).  A knot that occurs twice among the H steps (H > H_ref) keeps its LAST step's block,
// the order in which the reference's loop over the steps would have written it.
__global__ __launch_bounds__(128) void dz_rekey_kernel(double* step_base, size_t stride_b, size_t stride_i, double* knot,
                                                       const int* window, int H, int K, int blk, int dir) {
    const int b = (int)blockIdx.x / H, i = (int)blockIdx.x - b * H;
    const int* w = window + (size_t)b * (H + 2);
    const int t = w[i];
    if (dir == 0)
        for (int j = i + 1; j < H; ++j) if (w[j] == t) return;
    double* st = step_base + (size_t)b * stride_b + (size_t)i * stride_i;
    double* kn = knot + ((size_t)b * K + t) * (size_t)blk;
    const double* src = dir == 0 ? st : kn;
    double* dst = dir == 0 ? kn : st;
    for (int e = (int)threadIdx.x; e < blk; e += (int)blockDim.x) dst[e] = src[e];
}
// which: 0 = dz_good (Newton loop), 1 = evaluation slot 0 of S.dz (implicit_dynamics seam)
int launch_dz_rekey(const NewtonDev& S, double* knot, const int* window, int which, int dir, hipStream_t s) {
    const size_t blk = (size_t)S.nths * S.nd, H = S.dm.H;
    double* base = which == 0 ? S.dz_good : S.dz;
    const size_t stride_b = which == 0 ? H * blk : (size_t)CS * H * blk;
    hipLaunchKernelGGL(dz_rekey_kernel, dim3(S.dm.B * S.dm.H), dim3(128), 0, s, base, stride_b, blk, knot, window, S.dm.H, S.dm.H_ref, (int)blk, dir);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}
int launch_reset(const NewtonDev& S, const double* q0, const double* q1, int warm, hipStream_t s) {
    hipLaunchKernelGGL(reset_kernel, dim3(S.nb_launch), dim3(256), 0, s, S, q0, q1, warm);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}
// phase: 0 = both kernels, 1 = the per-slot residual kernel only, 2 = the decision kernel only (the host puts the join with the
// overlapped KKT kernel between the two: the slot kernel reads nothing that kernel writes)
template <int NQ, int NU, bool CF>
static int launch_resid_m(const NewtonDev& S, hipStream_t s, int n_slots, int phase) {
    const size_t lds2 = (size_t)2 * CIMPC_RESID_THREADS * sizeof(double);
    if (n_slots == -1) {    // small batches: one launch, the rollout's slots one after the other in its workgroup
                            // (-2: single rollouts - the two launches over ALL (rollout, slot) pairs, a handful of blocks: a deep
                            //  line search has seven slots, which the one workgroup would take one after the other)
        const size_t lds1 = (size_t)(CS * 256 + S.N) * sizeof(double);
        if (lds1 <= 64 * 1024) {
            if (phase == 1) return CIMPC_OK;
            hipLaunchKernelGGL((resid_decide_kernel<NQ, NU, CF, true>), dim3(S.nb_launch), dim3(CIMPC_RESID_THREADS), lds1, s, S);
            return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
        }
    }
    const int* list = (n_slots >= 0 && S.slot_list != nullptr) ? S.slot_list + (size_t)S.WQ.par * S.dm.B * CS : nullptr;
    const int grid = list != nullptr ? n_slots : S.nb_launch * CS;
    const size_t lds = (size_t)(256 + (S.N <= SLOT_ABS_MAX ? S.N : 0)) * sizeof(double);
    if (grid > 0 && phase != 2) hipLaunchKernelGGL((resid_slot_kernel<NQ, NU, CF>), dim3(grid), dim3(CIMPC_SLOT_THREADS), lds, s, S, list);
    if (phase != 1) hipLaunchKernelGGL((resid_decide_kernel<NQ, NU, CF, false>), dim3(S.nb_launch), dim3(CIMPC_RESID_THREADS), lds2, s, S);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}
template <int NQ, int NU>
static int launch_resid_t(const NewtonDev& S, hipStream_t s, int n_slots, int phase) {
    return S.dm.mode == CIMPC_MODE_CONFIGURATIONFORCE ? launch_resid_m<NQ, NU, true>(S, s, n_slots, phase) : launch_resid_m<NQ, NU, false>(S, s, n_slots, phase);
}
// End of a solve: what the host wants back, packed into ONE block (one device-to-host copy instead of five):
//   out[0..3] = sums over the rollouts of NewtonDev::stats, out[4] = sum of the Newton iteration counts,
//   out[8 + b (nu + 2) + ..] = [u_1 (nu) | newton iterations | r_norm] of rollout b   (core.traj.u[1], policy.jl:142).
// Counts travel as doubles (exact below 2^53).
__global__ __launch_bounds__(256) void solve_finish_kernel(NewtonDev S, double* out) {
    __shared__ double red[5][256];
    const int tid = threadIdx.x, B = S.dm.B, H = S.dm.H, nu = S.dm.nu, w = nu + 2;
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int b = tid; b < B; b += 256) {
        for (int k = 0; k < 4; ++k) acc[k] += (double)S.stats[(size_t)b * 4 + k];
        acc[4] += (double)S.newton_l[b];
        double* o = out + 8 + (size_t)b * w;
        for (int k = 0; k < nu; ++k) o[k] = S.traj.u[(size_t)b * H * nu + k];
        o[nu] = (double)S.newton_l[b];
        o[nu + 1] = S.r_norm[b];
    }
    for (int k = 0; k < 5; ++k) red[k][tid] = acc[k];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) for (int k = 0; k < 5; ++k) red[k][tid] += red[k][tid + st];
        __syncthreads();
    }
    if (tid < 5) out[tid] = red[tid][0];
}
// Flush of the lazily committed sensitivities (NewtonDev::good_src) at the end of a solve: the blocks an accept left in the
// evaluation slots of a rollout that had no KKT stage afterwards (it converged, ran out of iterations, or the solve was cut
// short) move to dz_good; afterwards every index is -1 and dz_good is what rounds 3-5 kept at all times.
// (one workgroup per (rollout, step): with one per rollout a lone rollout's 105 KB went through 256 threads in 52 dependent trips -
//  22 us at the end of every B = 1 solve)
__global__ __launch_bounds__(256) void dz_commit_kernel(NewtonDev S) {
    const int b = blockIdx.x, i = blockIdx.y, tid = threadIdx.x, H = S.dm.H;
    const int blk = S.nths * S.nd;
    const int s_ = S.good_src[(size_t)b * H + i];      // (uniform: every thread reads the same word)
    if (s_ < 0) return;
    double* good = S.dz_good + ((size_t)b * H + i) * blk;
    const double* slot = S.dz + (((size_t)b * CS + s_) * H + i) * blk;
    for (int e = tid; e < blk; e += 256) good[e] = slot[e];
    __syncthreads();                                     // every thread has read the index before it is reset
    if (tid == 0) S.good_src[(size_t)b * H + i] = -1;
}
int launch_dz_commit(const NewtonDev& S, hipStream_t s) {
    if (S.good_src == nullptr) return CIMPC_OK;
    hipLaunchKernelGGL(dz_commit_kernel, dim3(S.dm.B, S.dm.H), dim3(256), 0, s, S);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}
bool kkt_lazy_commit_available(const NewtonDev& S) {      // every KKT stage of the Newton loop runs kkt_body on fp64 tiles (launch_kkt_packed / launch_kkt_t)
    const int nq = S.dm.nq, nu = S.dm.nu;
    if (S.dm.mode != CIMPC_MODE_CONFIGURATION || S.kkt_list == nullptr || nq > 24 || nu > 24 || S.dm.H > 96 || S.dm.H > 128 || S.kkt_scalar != 0) return false;
#define X(q, u) if (nq == q && nu == u) return S.dm.H <= kkt_max_h<q, u>();
    CIMPC_NQNU(X)
#undef X
    return false;
}
int launch_solve_finish(const NewtonDev& S, double* out, hipStream_t s) {
    hipLaunchKernelGGL(solve_finish_kernel, dim3(1), dim3(256), 0, s, S, out);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}
__global__ void queue_recycle_kernel(IpQueues Q, int par) {
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k < Q.K) { *qcount(Q, par, k) = 0; *qhead(Q, k) = 0; }
}
int launch_queue_recycle(const IpQueues& Q, int par, hipStream_t s) {
    hipLaunchKernelGGL(queue_recycle_kernel, dim3((Q.K + 255) / 256), dim3(256), 0, s, Q, par);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}
int launch_resid_decide(const NewtonDev& S, hipStream_t s, int n_slots, int phase) {
    const int nq = S.dm.nq, nu = S.dm.nu;
#define X(q, u) if (nq == q && nu == u) return launch_resid_t<q, u>(S, s, n_slots, phase);
    CIMPC_NQNU(X)
#undef X
    return launch_resid_t<0, 0>(S, s, n_slots, phase);        // runtime dimensions (models without a compiled set)
}
template <int NQ, int NU>
static int launch_kkt_packed_t(const NewtonDev& S, const KktArgs& K, const int* list, int n, const int* n_dev, hipStream_t s, int pipe) {
    if constexpr (NQ <= 24 && NU <= 24) {
        constexpr int PACK = kkt_pack<NQ, NU>();
        const size_t lds = (size_t)PACK * kkt_lds_doubles<NQ, NU, 1>() * sizeof(double);
        static LdsOptIn optin;
        if (lds_opt_in(optin, (const void*)kkt_kernel_packed<NQ, NU>, lds) != CIMPC_OK) return CIMPC_ERR_HIP;
        // `pipe` (host schedule, CIMPC_KKT_PIPE): where the KKT solve is on the critical path (small batches, chained rounds).
        // Next to a busy sweep the pipelined kernel takes twice the CUs for half the time - measured neutral - so the
        // packed one-wave kernel stays there.
        if (pipe == 3) {      // duo: one workgroup of two one-wave chains per rollout (16-wide tiles)
            if constexpr (kkt_tld<NQ, NU>() == 16) {
                const size_t ldsd = (size_t)(kkt_duo_slice_doubles(1) + kkt_duo_slice_doubles(2)) * sizeof(double);
                static LdsOptIn optind;
                if (lds_opt_in(optind, (const void*)kkt_kernel_duo<NQ, NU>, ldsd) != CIMPC_OK) return CIMPC_ERR_HIP;
                hipLaunchKernelGGL((kkt_kernel_duo<NQ, NU>), dim3(n), dim3(128), ldsd, s, KktTwArgs{S, K, list, n, n_dev});
                return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
            } else return CIMPC_ERR_INVALID;
        }
        if (pipe == 2) {      // twisted: two workgroups of three wavefronts per rollout, one chain from either end
            const size_t lds3 = (size_t)kkt_tw_lds_doubles<NQ, NU>() * sizeof(double);
            static LdsOptIn optin3;
            if (lds_opt_in(optin3, (const void*)kkt_kernel_twisted<NQ, NU>, lds3) != CIMPC_OK) return CIMPC_ERR_HIP;
            hipLaunchKernelGGL((kkt_kernel_twisted<NQ, NU>), dim3(2 * n), dim3(192), lds3, s, KktTwArgs{S, K, list, n, n_dev});
            return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
        }
        if (pipe) {      // three wavefronts per rollout, software-pipelined forward recursion
            const size_t lds2 = (size_t)kkt_lds_doubles<NQ, NU, 3>() * sizeof(double);
            static LdsOptIn optin2;
            if (lds_opt_in(optin2, (const void*)kkt_kernel_pipe<NQ, NU>, lds2) != CIMPC_OK) return CIMPC_ERR_HIP;
            hipLaunchKernelGGL((kkt_kernel_pipe<NQ, NU>), dim3(n), dim3(192), lds2, s, S, K, list, n, n_dev);
            return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
        }
        hipLaunchKernelGGL((kkt_kernel_packed<NQ, NU>), dim3((n + PACK - 1) / PACK), dim3(64 * PACK), lds, s, S, K, list, n, n_dev);
        return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
    }
    return CIMPC_ERR_INVALID;
}
int launch_kkt_packed(const NewtonDev& S, int n_kkt, int list_par, hipStream_t s, const int* n_dev, int pipe) {
    const int latency = pipe >= 0 ? pipe : (S.kkt_same_round == 1 ? 1 : 0);
    const int nq = S.dm.nq, nu = S.dm.nu;
    if (n_dev != nullptr) n_kkt = S.dm.B;      // upper bound of the grid; surplus workgroups leave at once
    if (n_kkt <= 0) return CIMPC_OK;
    const bool force_scalar = S.kkt_scalar != 0;
    if (S.kkt_list == nullptr || S.dm.mode != CIMPC_MODE_CONFIGURATION || nq > 24 || nu > 24 || S.dm.H > 96 || force_scalar) return launch_kkt(S, s);
    const KktArgs K{S.res, S.delta, S.beta, 0.0, S.stage, 1};
    const int* list = S.kkt_list + (size_t)list_par * S.dm.B;
#define X(q, u) if (nq == q && nu == u) return launch_kkt_packed_t<q, u>(S, K, list, n_kkt, n_dev, s, latency);
    CIMPC_NQNU(X)
#undef X
    return launch_kkt(S, s);
}
// ---------------------------------------------------------------------------------------------------------
// :configurationforce mode through the :configuration solvers.  In cf mode the contact impulses (gamma_i, b_i) are extra
// primal variables (newton_residual.jl:18-54) that appear in exactly two places: their own rows of the dynamics
// constraint, d_i^y = y*(theta_i) - y_i with C = [S_i  -I]  (S_i = dy/d(q0, q1, u1): the y rows of the sensitivities),
// and the objective with the weights G = blkdiag(Cg, Cb) - 1e-100 in every example of the reference.  Eliminating them
// is exact:      G Dy - Dnu_y = r_y ,   S_i Dtheta_i - Dy - rho Dnu_y = r_nuy
//   =>  Dnu_y = G Dy - r_y ,  (I + rho G) Dy = S_i Dtheta_i + rho r_y - r_nuy ,
// and the rows of (u_i, q_i, q_{i+1}) see  S_i^T Dnu_y = -S_i^T r_y + O(G).  For G below fp64 resolution (the host
// takes this path only for max |G| <= 1e-30) the system for (Du, Dq, Dnu_q) is the :configuration-mode KKT system with
// the right-hand side  r_x + S^T r_y  - solved by the condensed MFMA kernel or, with a velocity objective, the banded
// LDL^T - followed by the two formulas above.  (General weights keep the dense LU of the reference.)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cf_reduce_pre_kernel(NewtonDev S, KktArgs K, CfReduce W) {
    const int b = blockIdx.x + S.b0, tid = threadIdx.x, nt = blockDim.x;
    if (K.stage != nullptr && K.stage[b] != STAGE_KKT) return;
    const int nq = S.dm.nq, nu = S.dm.nu, ny = S.dm.nc + S.dm.nb, H = S.dm.H, nr = S.nr, nd = S.nd, nths = S.nths;
    const int nr2 = nu + nq, N2 = H * (nr2 + nq);
    const double* dz = kkt_dz(S, K, b, H, nths, nd);
    const double* r = K.r + (size_t)b * S.N;
    double* dzq = W.dzq + (size_t)b * H * nths * nq;
    double* r2 = W.r2 + (size_t)b * N2;
    for (int e = tid; e < H * nths * nq; e += nt) {            // q rows of the sensitivities, packed
        const int ic = e / nq, rr = e - ic * nq;
        dzq[e] = dz[(size_t)ic * nd + rr];
    }
    auto dot_y = [&](int i, int c) {                            // (column c of S_i) . r_y,i
        const double* col = dz + ((size_t)i * nths + c) * nd + nq;
        const double* ry = r + (size_t)i * nr + nu;
        double acc = 0.0;
        for (int k = 0; k < ny; ++k) acc = fma(col[k], ry[k], acc);
        return acc;
    };
    for (int e = tid; e < H * nr2; e += nt) {                  // primal rows: r_x + S^T r_y
        const int j = e / nr2, k = e - j * nr2;
        double v;
        if (k < nu) v = r[(size_t)j * nr + k] + dot_y(j, 2 * nq + k);
        else {
            const int cq = k - nu;
            v = r[(size_t)j * nr + nu + ny + cq];
            if (j + 1 < H) v += dot_y(j + 1, nq + cq);          // q_{j+2} is q1 of step j+1 ...
            if (j + 2 < H) v += dot_y(j + 2, cq);               // ... and q0 of step j+2
        }
        r2[e] = v;
    }
    for (int e = tid; e < H * nq; e += nt) {                   // dual rows of q
        const int i = e / nq, k = e - i * nq;
        r2[(size_t)H * nr2 + e] = r[(size_t)H * nr + (size_t)i * nd + k];
    }
}

__global__ __launch_bounds__(256) void cf_reduce_post_kernel(NewtonDev S, KktArgs K, CfReduce W) {
    const int b = blockIdx.x + S.b0, tid = threadIdx.x, nt = blockDim.x;
    if (K.stage != nullptr && K.stage[b] != STAGE_KKT) return;
    const int nq = S.dm.nq, nu = S.dm.nu, nc = S.dm.nc, nbb = S.dm.nb, ny = nc + nbb, H = S.dm.H, nr = S.nr, nd = S.nd, nths = S.nths;
    const int nr2 = nu + nq, N2 = H * (nr2 + nq);
    const double* dz = kkt_dz(S, K, b, H, nths, nd);
    const double* r = K.r + (size_t)b * S.N;
    const double* d2 = W.d2 + (size_t)b * N2;
    double* D = K.delta + (size_t)b * S.N;
    const double beta = K.beta ? K.beta[b] : K.beta_scalar;
    const double rho = (double)H * beta * S.kappa;
    for (int e = tid; e < H * nr2; e += nt) {
        const int j = e / nr2, k = e - j * nr2;
        D[(size_t)j * nr + (k < nu ? k : ny + k)] = d2[e];
    }
    for (int e = tid; e < H * nq; e += nt) {
        const int i = e / nq, k = e - i * nq;
        D[(size_t)H * nr + (size_t)i * nd + k] = d2[(size_t)H * nr2 + e];
    }
    for (int e = tid; e < H * ny; e += nt) {                   // Dy_i = S_i Dtheta_i + rho r_y - r_nuy   ((I + rho G)^-1 = I here)
        const int i = e / ny, k = e - i * ny;
        double s = 0.0;
        for (int c = 0; c < nths; ++c) {
            double x;
            if (c < nq) x = i >= 2 ? d2[(size_t)(i - 2) * nr2 + nu + c] : 0.0;
            else if (c < 2 * nq) x = i >= 1 ? d2[(size_t)(i - 1) * nr2 + nu + (c - nq)] : 0.0;
            else x = d2[(size_t)i * nr2 + (c - 2 * nq)];
            s = fma(dz[((size_t)i * nths + c) * nd + nq + k], x, s);
        }
        D[(size_t)i * nr + nu + k] = s + rho * r[(size_t)i * nr + nu + k] - r[(size_t)H * nr + (size_t)i * nd + nq + k];
    }
    __syncthreads();
    for (int e = tid; e < H * ny; e += nt) {                   // Dnu_y = G Dy - r_y
        const int i = e / ny, k = e - i * ny;
        double g = 0.0;
        if (k < nc) { for (int c = 0; c < nc; ++c) g = fma(S.Cg[(size_t)i * nc * nc + (size_t)c * nc + k], D[(size_t)i * nr + nu + c], g); }
        else { const int kb = k - nc; for (int c = 0; c < nbb; ++c) g = fma(S.Cb[(size_t)i * nbb * nbb + (size_t)c * nbb + kb], D[(size_t)i * nr + nu + nc + c], g); }
        D[(size_t)H * nr + (size_t)i * nd + nq + k] = g - r[(size_t)i * nr + nu + k];
    }
    __syncthreads();
    if (K.finish) {
        // (the reduced solve may have been the twisted banded kernel: a hand-over that timed out left its mark - words 3 / 7 of the
        //  rollout's flag line carry the launch's stamp - and poisoned numbers; the KKT stage of the rollout is queued again instead)
        if (S.kkt_tw_flags != nullptr) {
            const int* fl = S.kkt_tw_flags + (size_t)b * KKT_TW_FLAGS;
            if (aload(fl + 3) == S.kkt_tw_epoch || aload(fl + 7) == S.kkt_tw_epoch) {
                if (tid == 0) {
                    const int pos = atomicAdd(&S.counters[1 * CPAD], 1);
                    if (S.kkt_list != nullptr && pos < S.dm.B) S.kkt_list[(size_t)S.WQ.par * S.dm.B + pos] = b;
                }
                return;
            }
        }
        __threadfence_block();
        start_line_search<BlockSync>(S, b, 1, tid, nt);
    }
}

NewtonDev cf_shadow(const NewtonDev& S) {          // the :configuration-mode problem the reduction solves
    NewtonDev S2 = S;
    S2.dm.mode = CIMPC_MODE_CONFIGURATION;
    S2.nd = S.dm.nq; S2.nr = S.dm.nu + S.dm.nq; S2.N = S.dm.H * (S.dm.nu + 2 * S.dm.nq);
    S2.Cg = nullptr; S2.Cb = nullptr;
    return S2;
}
size_t cf_reduce_doubles(const NewtonDev& S) {
    return (size_t)S.dm.B * ((size_t)S.dm.H * S.nths * S.dm.nq + 2 * (size_t)S.dm.H * (S.dm.nu + 2 * S.dm.nq));
}
int launch_kkt_cf_reduced(const NewtonDev& S, const KktArgs& K, double* ws, double* dense_ws, hipStream_t s) {
    const size_t B = S.dm.B, n_dz = (size_t)S.dm.H * S.nths * S.dm.nq, N2 = (size_t)S.dm.H * (S.dm.nu + 2 * S.dm.nq);
    CfReduce W{ws, ws + B * n_dz, ws + B * n_dz + B * N2};
    hipLaunchKernelGGL(cf_reduce_pre_kernel, dim3(S.nb_launch), dim3(256), 0, s, S, K, W);
    const NewtonDev S2 = cf_shadow(S);
    const KktArgs K2{W.r2, W.d2, K.beta, K.beta_scalar, K.stage, 0, W.dzq};
    const int rc = S.V != nullptr ? launch_kkt_dense_args(S2, K2, dense_ws, s, true) : launch_kkt_any(S2, K2, s);
    if (rc != CIMPC_OK) return rc;
    hipLaunchKernelGGL(cf_reduce_post_kernel, dim3(S.nb_launch), dim3(256), 0, s, S, K, W);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}
int launch_kkt_cf_reduced_newton(const NewtonDev& S, double* ws, double* dense_ws, hipStream_t s) {
    const KktArgs K{S.res, S.delta, S.beta, 0.0, S.stage, 1};
    return launch_kkt_cf_reduced(S, K, ws, dense_ws, s);
}
int launch_kkt_cf_reduced_raw(const NewtonDev& S, const double* r_dev, double beta, double* delta_dev, double* ws, double* dense_ws, hipStream_t s) {
    const KktArgs K{r_dev, delta_dev, nullptr, beta, nullptr, 0};
    return launch_kkt_cf_reduced(S, K, ws, dense_ws, s);
}
bool kkt_cf_reduce_available(const NewtonDev& S) {   // the reduced problem must have a :configuration-mode solver
    if (S.dm.mode != CIMPC_MODE_CONFIGURATIONFORCE) return false;
    const NewtonDev S2 = cf_shadow(S);
    if (S.V != nullptr) return kkt_banded_available(S2);
    const int nq = S.dm.nq, nu = S.dm.nu;
#define X(q, u) if (nq == q && nu == u) return true;
    CIMPC_NQNU(X)
#undef X
    return false;
}

bool kkt_twisted_available(const NewtonDev& S) {      // MFMA tiles, a horizon long enough for two chains, the backward staging
    if (S.dm.mode != CIMPC_MODE_CONFIGURATION || S.kkt_list == nullptr || S.kkt_scalar != 0 || S.kkt_tw_xch == nullptr) return false;
    if (S.dm.nq > 24 || S.dm.nu > 24 || S.dm.H < KKT_TW_MIN_H || S.dm.H > 96) return false;
    return kkt_condensed_available(S);
}

bool kkt_duo_available(const NewtonDev& S) {      // the twisted solve's conditions on 16-wide tiles
    return S.dm.nq <= 16 && S.dm.nu <= 16 && kkt_twisted_available(S);
}

bool kkt_condensed_available(const NewtonDev& S) {      // a compiled condensed solve exists for these (nq, nu)
    const int nq = S.dm.nq, nu = S.dm.nu;
#define X(q, u) if (nq == q && nu == u) return true;
    CIMPC_NQNU(X)
#undef X
    return false;
}
int launch_kkt(const NewtonDev& S, hipStream_t s) {
    KktArgs K{S.res, S.delta, S.beta, 0.0, S.stage, 1};
    return launch_kkt_any(S, K, s);
}
int launch_kkt_mixed_newton(const NewtonDev& S, double* ws, int* n_fallback, hipStream_t s) {
    const KktArgs K{S.res, S.delta, S.beta, 0.0, S.stage, 1};
    return launch_kkt_mixed(S, K, ws, n_fallback, s);
}
int launch_kkt_mixed_raw(const NewtonDev& S, const double* r_dev, double beta, double* delta_dev, double* ws, int* n_fallback, hipStream_t s) {
    const KktArgs K{r_dev, delta_dev, nullptr, beta, nullptr, 0};
    return launch_kkt_mixed(S, K, ws, n_fallback, s);
}
template <int NQ, int NU>
static int launch_kkt_twisted_t(const NewtonDev& S, const KktArgs& K, hipStream_t s) {
    if constexpr (NQ <= 24 && NU <= 24) {
        const size_t lds = (size_t)kkt_tw_lds_doubles<NQ, NU>() * sizeof(double);
        static LdsOptIn optin;
        if (lds_opt_in(optin, (const void*)kkt_kernel_twisted<NQ, NU>, lds) != CIMPC_OK) return CIMPC_ERR_HIP;
        hipLaunchKernelGGL((kkt_kernel_twisted<NQ, NU>), dim3(2 * S.nb_launch), dim3(192), lds, s, KktTwArgs{S, K, nullptr, S.nb_launch, nullptr});
        return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
    }
    return CIMPC_ERR_INVALID;
}
int launch_kkt_raw(const NewtonDev& S, const double* r_dev, double beta, double* delta_dev,
                   hipStream_t s) {
    KktArgs K{r_dev, delta_dev, nullptr, beta, nullptr, 0};
    if (S.kkt_tw_raw == 2 && kkt_duo_available(S)) {      // (CIMPC_KKT_DUO=2: the duo kernel at the B1 seam too - the parity tests reach it in isolation here)
        const int nq = S.dm.nq, nu = S.dm.nu;
        const int* list = nullptr;
#define X(q, u) if (nq == q && nu == u) return launch_kkt_packed_t<q, u>(S, K, list, S.nb_launch, nullptr, s, 3);
        CIMPC_NQNU(X)
#undef X
    }
    if (S.kkt_tw_raw != 0 && kkt_twisted_available(S)) {      // a lone solve is latency-bound: two chains from either end
        const int nq = S.dm.nq, nu = S.dm.nu;
#define X(q, u) if (nq == q && nu == u) return launch_kkt_twisted_t<q, u>(S, K, s);
        CIMPC_NQNU(X)
#undef X
    }
    return launch_kkt_any(S, K, s);
}

}  // namespace cimpc
