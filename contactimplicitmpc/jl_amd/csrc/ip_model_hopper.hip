// Interior-point sweep kernel instantiation: hopper dimensions.
#include "ip_kernel_impl.h"
namespace cimpc {
CIMPC_DEFINE_MODEL(hopper, 4, 2, 2, 1, 2)
}  // namespace cimpc
