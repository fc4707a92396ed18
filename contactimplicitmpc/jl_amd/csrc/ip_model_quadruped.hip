// Interior-point sweep kernel instantiation: quadruped dimensions.
#include "ip_kernel_impl.h"
namespace cimpc {
CIMPC_DEFINE_MODEL(quadruped, 11, 8, 2, 4, 8)
}  // namespace cimpc

#ifdef CIMPC_SWEEP_PROF
// diagnostic builds only: read (and clear) the per-wave clock accounting of the quadruped sweep kernel
extern "C" int cimpc_debug_sweep_prof(unsigned long long* out16) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(cimpc::g_sweep_prof), sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(cimpc::g_sweep_prof), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

#ifdef CIMPC_UBENCH
// diagnostic builds only (-DCIMPC_UBENCH, scripts/dbg/ubench_ip.py): shader-clock cost of the interior-point primitives on `waves`
// wavefronts of ONE workgroup (4 problems per wave, the same problem in every group) - what a lone wave pays per call, i.e. the
// latency chain the small batches, the launch tails and the asynchronous solve run on.
namespace cimpc {
template <class M>
__global__ __launch_bounds__(256, 2) void ip_ubench_kernel(const double* tabs, cimpc_ip_opts o, int reps, long long* out) {
    constexpr LinLayout L(M::NX, M::NY, M::NTH, M::G, M::NTHS, M::ADJ);
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = (int)threadIdx.x, grp = tid / M::G, l = tid % M::G;
    stage_table<M>(smem, tabs, 0, tid);
    __syncthreads();
    IpSolver<M> S;
    S.bind(smem, smem + L.hot + (size_t)grp * M::LDS_GROUP, l);
    S.x = S.vx ? S.x0 : 0.0; S.y1 = 1.0; S.y2 = 1.0;
    long long t[8];
    double sink = 0.0;
    auto opaque = [&] { asm volatile("" : "+v"(S.x), "+v"(S.y1), "+v"(S.y2)); };
    t[0] = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) { S.residual(1e-3); sink += S.rdyn + S.rrst; opaque(); }
    t[1] = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) { sink += S.r_violation() + S.k_violation(); S.rdyn += 1e-30; opaque(); }
    t[2] = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) { S.factorize(1e-9); sink += S.rdinv; S.y1 += 1e-30 * S.rdinv; opaque(); }
    t[3] = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) { S.linear_solve(); sink += S.Dx_ + S.Dy1_; S.rdyn += 1e-30 * S.Dx_; opaque(); }
    t[4] = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) { sink += S.step_length(0.99); S.Dy1_ += 1e-30; opaque(); }
    t[5] = (long long)__builtin_amdgcn_s_memtime();
    double reg = 0.0, r_vio = 1.0, k_vio = 1.0;
    S.x = S.vx ? S.x0 : 0.0; S.y1 = 1.0; S.y2 = 1.0;
    S.residual(0.0); r_vio = S.r_violation(); k_vio = S.k_violation();
    int its = 0;
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        if (r_vio < o.r_tol && k_vio < o.kappa_tol) { S.x = S.vx ? S.x0 : 0.0; S.y1 = 1.0; S.y2 = 1.0; S.residual(0.0); r_vio = S.r_violation(); k_vio = S.k_violation(); }
        (void)S.iterate(o, o.kappa_tol / o.undercut, 1.0 - o.eps_min, reg, r_vio, k_vio);
        ++its;
    }
    t[6] = (long long)__builtin_amdgcn_s_memtime();
    if ((tid & 63) == 0) {
        long long* w = out + 8 * (tid >> 6);
        for (int k = 0; k < 6; ++k) w[k] = t[k + 1] - t[k];
        w[6] = its; w[7] = (long long)(sink * 1e-300);
    }
}
}  // namespace cimpc
extern "C" int cimpc_ubench_ip_quadruped(const double* tabs_dev, const cimpc_ip_opts* o, int waves, int reps, long long* out_host) {
    using M = cimpc::Model<11, 8, 2, 4, 8, 0>;
    constexpr cimpc::LinLayout L(M::NX, M::NY, M::NTH, M::G, M::NTHS, M::ADJ);
    long long* d = nullptr;
    if (hipMalloc(&d, 8 * 8 * sizeof(long long)) != hipSuccess) return -1;
    const size_t lds = (size_t)(L.hot + 16 * M::LDS_GROUP) * sizeof(double);
    (void)hipFuncSetAttribute((const void*)cimpc::ip_ubench_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((cimpc::ip_ubench_kernel<M>), dim3(1), dim3(64 * waves), lds, nullptr, tabs_dev, *o, reps, d);
    const bool ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(out_host, d, 8 * 8 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    return ok ? 0 : -1;
}
#endif
