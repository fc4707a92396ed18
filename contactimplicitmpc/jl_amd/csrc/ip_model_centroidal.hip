// Interior-point sweep kernel instantiation: centroidal dimensions.
#include "ip_kernel_impl.h"
namespace cimpc {
CIMPC_DEFINE_MODEL(centroidal, 18, 12, 3, 4, 16)
}  // namespace cimpc

#ifdef CIMPC_SWEEP_PROF
// diagnostic builds only: read (and clear) the per-wave clock accounting of the centroidal sweep kernels (both builds share the counters).
// Through a kernel: the counters are a TU-local device variable of the same name in every model's translation unit, and the HIP
// runtime registers a host-visible symbol of that name only once (ip_model_quadruped.hip has it).
namespace cimpc {
__global__ void sweep_prof_read_kernel(unsigned long long* out) {
    if (threadIdx.x < 16) { out[threadIdx.x] = g_sweep_prof[threadIdx.x]; g_sweep_prof[threadIdx.x] = 0ull; }
}
}  // namespace cimpc
extern "C" int cimpc_debug_sweep_prof_centroidal(unsigned long long* out16) {
    unsigned long long* d = nullptr;
    if (hipMalloc(&d, 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    hipLaunchKernelGGL(cimpc::sweep_prof_read_kernel, dim3(1), dim3(64), 0, nullptr, d);
    const bool ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(out16, d, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    return ok ? 0 : -1;
}
#endif

#ifdef CIMPC_UBENCH
// diagnostic builds only (-DCIMPC_UBENCH, scripts/dbg/ubench_ip.py): ip_kernel_impl.h: ip_ubench_kernel on the THROUGHPUT build of the 32-lane form
extern "C" int cimpc_ubench_ip_centroidal(const double* tabs_dev, const cimpc_ip_opts* o, int waves, int reps, long long* out_host) {
    using M = cimpc::Model<18, 12, 3, 4, 16, 0, 1>;
    constexpr cimpc::LinLayout L(M::NX, M::NY, M::NTH, M::G, M::NTHS, M::ADJ);
    long long* d = nullptr;
    if (hipMalloc(&d, 8 * 8 * sizeof(long long)) != hipSuccess) return -1;
    const size_t lds = (size_t)(L.hot + 16 * M::LDS_GROUP) * sizeof(double);
    if (hipFuncSetAttribute((const void*)cimpc::ip_ubench_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
    hipLaunchKernelGGL((cimpc::ip_ubench_kernel<M>), dim3(1), dim3(64 * waves), lds, nullptr, tabs_dev, *o, reps, d);
    const bool ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(out_host, d, 8 * 8 * sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    return ok ? 0 : -1;
}
#endif
