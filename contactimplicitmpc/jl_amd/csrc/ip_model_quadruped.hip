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
