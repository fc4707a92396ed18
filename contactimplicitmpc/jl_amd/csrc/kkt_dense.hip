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
    const double* dzb = K.stage ? S.dz_good + (size_t)b * H * nths * nd : S.dz + (size_t)b * CS * H * nths * nd;
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

size_t kkt_dense_workspace_doubles(const NewtonDev& S) {
    return (size_t)S.dm.B * ((size_t)S.N * S.N + 2 * (size_t)S.N);
}

static int launch_kkt_dense(const NewtonDev& S, const KktArgs& K, double* ws, hipStream_t s) {
    hipLaunchKernelGGL(kkt_dense_kernel, dim3(S.nb_launch), dim3(256), 0, s, S, K, ws);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}
int launch_kkt_dense_newton(const NewtonDev& S, double* ws, hipStream_t s) {
    KktArgs K{S.res, S.delta, S.beta, 0.0, S.stage, 1};
    return launch_kkt_dense(S, K, ws, s);
}
int launch_kkt_dense_raw(const NewtonDev& S, const double* r_dev, double beta, double* delta_dev, double* ws, hipStream_t s) {
    KktArgs K{r_dev, delta_dev, nullptr, beta, nullptr, 0};
    return launch_kkt_dense(S, K, ws, s);
}

}  // namespace cimpc
