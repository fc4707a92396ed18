// Runtime-dimension interior-point sweep: the fallback for model dimensions without a compiled lane-group kernel
// (ip_dispatch.hip: CIMPC_MODELS).  The reference's operators are generic in (nx, ny, nθ)
// (/root/reference/src/controller/linearized_solver.jl:15-65); with this kernel any model with nx, ny <= 64 loads -
// e.g. centroidal_quadruped_wall (nq 18, nc 8, nb 32: ny = 48), which no 32-lane group holds.
//
// Correct before fast: ONE problem per wavefront (lane = row / column index), matrices in LDS, the knot's table read
// from global memory (L2) where it is needed, no parking, sensitivities right after convergence.  Same callbacks and the
// same iteration spec as the lane-group kernel (ip_kernel_impl.h):
//   rlin! linearized_solver.jl:364-373 | rzlin! + schur_factorize! :378-399, schur.jl:80-88 | MGS factorize! / qr_solve!
//   qr.jl:113-158 | linear_solve! (vector / matrix rhs) :424-479 | z_initialize! simulation.jl:59-63
// It consumes the lock-step queues exactly like ip_queue_kernel (items bucketed by reference knot, done_count per slot), so
// the Newton rounds above it do not know which sweep kernel ran.
#include <algorithm>
#include <cstdio>

#include "cimpc_internal.h"
#include "lin_table.h"
#include "newton_state.h"

namespace cimpc {

namespace {

constexpr int GW = 64;      // lanes of the (single) problem of a wavefront = lane stride of the packed table

__device__ __forceinline__ void wfence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// Wave-uniform values are handed to the compiler AS uniform (v_readfirstlane -> SGPR): every loop / branch of this kernel
// that depends on a reduction result is then a scalar branch.  (The whole wavefront is always active here, so lane 0 is the
// first active lane.)  The queue is partitioned statically for the same reason - see the knot loop below.
__device__ __forceinline__ double uniform(double v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane(__double2loint(v));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
    return uniform(v);
}
__device__ __forceinline__ double wmin(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmin(v, __shfl_xor(v, m, 64));
    return uniform(v);
}
__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return uniform(v);            // every lane holds lane 0's bits
}

struct Gen {
    int nx, ny, nth, nths, nd, nc, nb, nq, mode, ld;
    LinLayout L;
    const double* tab;    // this knot's table (global memory)
    double* Qm;           // LDS [ny][ld]  column j of Q at Qm + j*ld (orthonormal after factorize)
    double* Rm;           // LDS [ny][ld]  row k of R at Rm + k*ld (upper triangle incl. diagonal)
    double* buf;          // LDS [64] broadcast buffer
    int l;
    bool vx, vy;
    // per-lane constants of the knot
    double ry2, ry1d, caibd, rdyn0, rrst0, x0, y10, y20;
    double tthdyn, tthrst, altl;
    double x, y1, y2, rdyn, rrst, rbil, Dx_, Dy1_, Dy2_;
    double y1r, y2r;

    __device__ Gen(int nx_, int ny_, int nth_) : L(nx_, ny_, nth_, GW) {}

    __device__ __forceinline__ double T(int off, int k) const { return tab[off + k * GW + l]; }

    // y_l = sum_k M[l,k] v_k with v lane-indexed (n entries), M lane-strided in the table at `off`
    __device__ __forceinline__ double matvec(int off, int n, double v) {
        wfence();
        buf[l] = v;
        wfence();
        double s = 0.0;
        for (int k = 0; k < n; ++k) s = fma(T(off, k), buf[k], s);
        return s;
    }

    __device__ void residual(double kappa) {
        const double dx = x - x0, dy1 = y1 - y10, dy2 = y2 - y20;
        wfence();
        buf[l] = dx;
        wfence();
        double a = 0.0, c = 0.0;
        for (int k = 0; k < nx; ++k) { const double v = buf[k]; a = fma(T(L.oDx, k), v, a); c = fma(T(L.oRx, k), v, c); }
        wfence();
        buf[l] = dy1;
        wfence();
        double bb = 0.0, e = 0.0;
        for (int k = 0; k < ny; ++k) { const double v = buf[k]; bb = fma(T(L.oDy1, k), v, bb); e = fma(T(L.oRy1, k), v, e); }
        rdyn = vx ? ((rdyn0 + a) + bb) + tthdyn : 0.0;
        rrst = vy ? ((((rrst0 + c) + e) + ry2 * dy2) + tthrst) + altl : 0.0;
        rbil = vy ? (y1 * y2 - kappa) : 0.0;
    }
    __device__ double r_violation() const { return wmax(fmax(fabs(rdyn), fabs(rrst))); }
    __device__ double k_violation() const { return wmax(fabs(rbil)); }

    // rzlin! + schur_factorize! + MGS (right-looking; per column the reference's left-looking arithmetic, qr.jl:113-137)
    __device__ void factorize(double reg) {
        y1r = fmax(y1, reg);
        y2r = fmax(y2, reg);
        const double dd = vy ? ry2 * y2r / y1r : 0.0;
        wfence();
        if (vy) {   // column l of D - C A^-1 B: W is stored row-strided (row i at i*GW + j), diagonal kept apart
            for (int r = 0; r < ny; ++r) Qm[r + l * ld] = (r == l) ? ((ry1d - dd) - caibd) : tab[L.oW + r * GW + l];
            for (int k = 0; k < ny; ++k) Rm[l * ld + k] = 0.0;
        }
        wfence();
        for (int k = 0; k < ny; ++k) {
            double n2 = 0.0;
            for (int r = 0; r < ny; ++r) { const double v = Qm[r + k * ld]; n2 = fma(v, v, n2); }
            const double rkk = sqrt(n2), inv = 1.0 / rkk;
            wfence();
            if (vy) Qm[l + k * ld] *= inv;              // lane = row: q_k = a_k / |a_k|
            if (l == k) Rm[k * ld + k] = rkk;
            wfence();
            if (vy && l > k) {                          // lane = column j > k
                double dot = 0.0;
                for (int r = 0; r < ny; ++r) dot = fma(Qm[r + k * ld], Qm[r + l * ld], dot);
                Rm[k * ld + l] = dot;
                for (int r = 0; r < ny; ++r) Qm[r + l * ld] = fma(-dot, Qm[r + k * ld], Qm[r + l * ld]);
            }
            wfence();
        }
    }

    // x = R^-1 Q^T rhs (qr_solve!, qr.jl:142-158); rhs lane-indexed over ny
    __device__ double qr_solve(double rhs) {
        wfence();
        buf[l] = vy ? rhs : 0.0;
        wfence();
        double c = 0.0;
        if (vy) for (int r = 0; r < ny; ++r) c = fma(Qm[r + l * ld], buf[r], c);
        double t = 0.0;
        for (int k = ny - 1; k >= 0; --k) {
            wfence();
            if (l == k) { t = c / Rm[k * ld + k]; buf[0] = t; }
            wfence();
            const double xk = buf[0];
            if (vy && l < k) c = fma(-Rm[l * ld + k], xk, c);
        }
        return t;
    }

    // schur_solve! (schur.jl:93-110): returns temp (lane-indexed over ny); xs = Ai (u + B temp) (over nx)
    __device__ double schur_solve(double u, double v, double& xs) {
        const double bq = matvec(L.oCAi, nx, vx ? u : 0.0);
        const double t = qr_solve(vy ? bq - v : 0.0);
        const double w = matvec(L.oDy1, ny, vy ? t : 0.0);
        const double ww = vx ? u + w : 0.0;
        xs = matvec(L.oAi, nx, ww);
        return t;
    }
    __device__ void linear_solve() {
        const double u = rdyn;
        const double v = vy ? (rrst - ry2 * rbil / y1r) : 0.0;
        const double t = schur_solve(u, v, Dx_);
        Dy1_ = vy ? -t : 0.0;
        Dy2_ = vy ? ((rbil - y2r * Dy1_) / y1r) : 0.0;
    }
    __device__ double step_length(double tau) const {
        double a = 1.0;
        if (vy && Dy1_ > 0.0) a = fmin(a, tau * y1 / Dy1_);
        if (vy && Dy2_ > 0.0) a = fmin(a, tau * y2 / Dy2_);
        return wmin(a);
    }
    // ONE interior-point iteration (DESIGN.md "IP iteration spec"); true = stalled
    __device__ bool iterate(const cimpc_ip_opts& o, double& reg, double& r_vio, double& k_vio) {
        reg = (k_vio < o.kappa_reg) ? k_vio * o.gamma_reg : 0.0;
        factorize(reg);
        linear_solve();
        const double a_aff = step_length(1.0);
        const double mu = wsum(vy ? y1 * y2 : 0.0) / (double)ny;
        const double mu_aff = wsum(vy ? (y1 - a_aff * Dy1_) * (y2 - a_aff * Dy2_) : 0.0) / (double)ny;
        double sg = fmin(fmax(mu_aff / mu, 0.0), 1.0);
        sg = sg * sg * sg;
        const double kc = fmax(sg * mu, o.kappa_tol / o.undercut);
        rbil = vy ? ((y1 * y2 - kc) + Dy1_ * Dy2_) : 0.0;
        linear_solve();
        const double vm = fmax(r_vio, k_vio);
        const double tau = fmax(1.0 - o.eps_min, 1.0 - vm * vm);
        const double alpha = step_length(tau);
        if (alpha < o.stall_alpha) return true;
        x -= alpha * Dx_;
        y1 = vy ? (y1 - alpha * Dy1_) : 1.0;
        y2 = vy ? (y2 - alpha * Dy2_) : 1.0;
        double k_c = 0.0, r_c = 0.0, back = alpha;
        for (int s = 1; s <= o.max_ls; ++s) {
            residual(0.0);
            k_c = k_violation();
            r_c = r_violation();
            if (r_c <= r_vio || k_c <= k_vio) break;
            back *= o.ls_scale;
            x += back * Dx_;
            y1 = vy ? (y1 + back * Dy1_) : 1.0;
            y2 = vy ? (y2 + back * Dy2_) : 1.0;
        }
        k_vio = k_c;
        r_vio = r_c;
        return false;
    }
};

struct GenDims { int nq, nu, nw, nc, nb, mode; };

}  // namespace

__global__ __launch_bounds__(64) void ip_generic_kernel(IpParams p, GenDims gd) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int l = (int)threadIdx.x;
    const int nx = gd.nq, ny = 2 * gd.nc + gd.nb, nth = 2 * gd.nq + gd.nu + gd.nw + 2, nths = 2 * gd.nq + gd.nu;
    const int nz = gd.nq + 4 * gd.nc + 2 * gd.nb, nd = gd.mode ? gd.nq + gd.nc + gd.nb : gd.nq;
    Gen S(nx, ny, nth);
    S.nx = nx; S.ny = ny; S.nth = nth; S.nths = nths; S.nd = nd; S.nc = gd.nc; S.nb = gd.nb; S.nq = gd.nq; S.mode = gd.mode;
    S.ld = ny | 1;
    S.Qm = sm; S.Rm = sm + ny * S.ld; S.buf = S.Rm + ny * S.ld;
    double* dth = S.buf + GW;
    S.l = l; S.vx = l < nx; S.vy = l < ny;
    const LinLayout& L = S.L;
    const cimpc_ip_opts o = p.o;
    const int K = p.Q.K, par = p.Q.par, cap = p.Q.cap;
    const int PS = 2 * nx + 4 * ny + 4;
    for (int kk = 0; kk < K; ++kk) {
        const int knot = ((int)blockIdx.x + kk) % K;
        const int n = *qcount(p.Q, par, knot);
        if (n == 0) continue;
        const int* items = p.Q.items + ((size_t)par * K + knot) * cap;
        S.tab = p.tab + (size_t)knot * L.size;
        const double* tVec = S.tab + L.oVec;
        S.ry2 = tVec[LinLayout::V_RY2 * GW + l]; S.ry1d = tVec[LinLayout::V_RY1D * GW + l]; S.caibd = tVec[LinLayout::V_CAIBD * GW + l];
        S.rdyn0 = tVec[LinLayout::V_RDYN0 * GW + l]; S.rrst0 = tVec[LinLayout::V_RRST0 * GW + l];
        S.x0 = tVec[LinLayout::V_X0 * GW + l]; S.y10 = tVec[LinLayout::V_Y10 * GW + l]; S.y20 = tVec[LinLayout::V_Y20 * GW + l];
        // Problems of the knot: dynamic pull (default since round 4) or static partition over the workgroups (p.generic_static).
        // The pull's first two forms hung on the GPU in this single-wave workgroup: `if (l == 0) idx = atomicAdd(head, 1);
        // idx = __shfl(idx, 0)` was structurised by hipcc as a divergent loop whose latch never repeats the atomic (lanes 1..63
        // re-read a stale index for ever, seen in the ISA), and the v_readfirstlane variant of it still sat behind the same
        // divergent `if`.  This form has NO divergent control flow at all: every lane issues the atomic - lane 0 adds 1, the
        // others add 0 (their return values are discarded) - and the claim of lane 0, the first active lane, reaches the wave
        // through v_readfirstlane, so the loop condition is a scalar compare.
        int* head = qhead(p.Q, knot);
        const int G0 = (int)gridDim.x;
        int idx = p.generic_static ? (int)blockIdx.x : __builtin_amdgcn_readfirstlane(atomicAdd(head, l == 0 ? 1 : 0));
        for (; idx < n; idx = p.generic_static ? idx + G0 : __builtin_amdgcn_readfirstlane(atomicAdd(head, l == 0 ? 1 : 0))) {
            const int prob = items[idx];
            const int sb = prob / p.H, i = prob - sb * p.H;
            const size_t pi = (size_t)prob;
            // theta - theta0, then rthdyn (th - th0), rthrst (th - th0)
            wfence();
            for (int k = l; k < nth; k += GW) dth[k] = p.theta[pi * nth + k] - S.tab[L.oTh0 + k];
            wfence();
            {
                double a = 0.0, c = 0.0;
                for (int k = 0; k < nth; ++k) { const double dv = dth[k]; a = fma(S.T(L.oRthDyn, k), dv, a); c = fma(S.T(L.oRthRst, k), dv, c); }
                S.tthdyn = a; S.tthrst = c;
            }
            S.altl = (p.alt != nullptr && l < gd.nc) ? p.alt[(size_t)(sb / p.slots) * gd.nc + l] : 0.0;
            const double qinit = S.vx ? p.q[((size_t)sb * (p.H + 2) + (i + 2)) * gd.nq + l] : 0.0;
            // z_initialize!: z .= 1, z[iq2] = q
            S.x = qinit; S.y1 = 1.0; S.y2 = 1.0;
            S.residual(0.0);
            double r_vio = S.r_violation(), k_vio = S.k_violation(), reg = 0.0;
            int iters = 0;
            bool stalled = false;
            // per-solve time budget (cimpc_ip_opts::max_time): wave-uniform clock, read once per iteration when a budget is set
            const long long t_solve0 = p.budget_ticks > 0 ? (long long)wall_clock64() : 0;
            while (!stalled && !(r_vio < o.r_tol && k_vio < o.kappa_tol) && iters < o.max_iter &&
                   !(p.budget_ticks > 0 && (long long)wall_clock64() - t_solve0 >= p.budget_ticks)) {
                ++iters;
                stalled = S.iterate(o, reg, r_vio, k_vio);
            }
            const int code = (!stalled && r_vio < o.r_tol && k_vio < o.kappa_tol) ? 1 : 0;
            if (l == 0) { p.status[pi] = code; p.iters[pi] = iters; p.pflag[pi] = 0; }
            if (S.vx) p.d[pi * nd + l] = S.x - qinit;
            if (gd.mode == CIMPC_MODE_CONFIGURATIONFORCE) {
                if (l < gd.nc) p.d[pi * nd + nx + l] = S.y1 - p.gam[pi * gd.nc + l];
                else if (l < gd.nc + gd.nb) p.d[pi * nd + nx + l] = S.y1 - p.bfr[pi * gd.nb + (l - gd.nc)];
            }
            if (p.zout != nullptr) {
                double* zo = p.zout + pi * nz;
                if (S.vx) zo[l] = S.x;
                if (S.vy) { zo[nx + l] = S.y1; zo[nx + ny + l] = S.y2; }
            }
            if (l == 0) {      // (converged z* / reg are parked like the lane-group kernel does, for symmetry of the state arrays)
                double* ps = p.pstate + pi * PS;
                ps[PS - 2] = reg;
            }
            if (code == 1) {
                // differentiate_solution!: dz = -(rz^-1 rth) at z*, reg = max(reg, kappa_tol*gamma_reg); consumed block only
                S.factorize(fmax(reg, o.kappa_tol * o.gamma_reg));
                double* dzo = p.dz + pi * (size_t)(nths * nd);
                for (int c = 0; c < nths; ++c) {
                    const double u = S.vx ? S.T(L.oRthDyn, c) : 0.0;
                    const double v = S.vy ? S.T(L.oRthRst, c) : 0.0;
                    double xs;
                    const double t = S.schur_solve(u, v, xs);
                    if (S.vx) dzo[c * nd + l] = -xs;
                    if (gd.mode == CIMPC_MODE_CONFIGURATIONFORCE && l < gd.nc + gd.nb) dzo[c * nd + nx + l] = t;
                }
            }
            if (l == 0) atomicAdd(&p.Q.done_count[sb], 1);
        }
    }
}

// ---- host side --------------------------------------------------------------------------------------------------------
bool ip_generic_available(const cimpc_dims* dm) {
    const int nx = dm->nq, ny = 2 * dm->nc + dm->nb;
    return nx >= 1 && ny >= 1 && nx <= GW && ny <= GW;
}
void ip_generic_info(const cimpc_dims* dm, KernelInfo* info) {
    const LinLayout L(dm->nq, 2 * dm->nc + dm->nb, 2 * dm->nq + dm->nu + dm->nw + 2, GW);
    info->G = GW;
    info->lds_table = 0;
    info->lds_group = 0;
    info->tab_size = L.size;
}
int launch_ip_generic(const cimpc_dims* dm, const IpParams& p, hipStream_t s) {
    if (!ip_generic_available(dm)) return CIMPC_ERR_INVALID;
    const int ny = 2 * dm->nc + dm->nb, nth = 2 * dm->nq + dm->nu + dm->nw + 2, ld = ny | 1;
    const size_t lds = (size_t)(2 * ny * ld + GW + nth + 8) * sizeof(double);
    static LdsOptIn optin;
    if (lds_opt_in(optin, (const void*)ip_generic_kernel, lds) != CIMPC_OK) return CIMPC_ERR_HIP;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1)));
    const long long nprob = (long long)dm->B * CS * dm->H;
    const int grid = (int)std::max<long long>(1, std::min<long long>(nprob, 256ll * per_cu));
    const GenDims gd{dm->nq, dm->nu, dm->nw, dm->nc, dm->nb, dm->mode};
    hipLaunchKernelGGL(ip_generic_kernel, dim3(grid), dim3(64), lds, s, p, gd);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}

}  // namespace cimpc
