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
// diagnostic builds only (-DCIMPC_UBENCH, scripts/dbg/ubench_ip.py): see ip_kernel_impl.h: ip_ubench_kernel
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
