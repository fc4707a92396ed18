// Interior-point sweep kernel instantiation: quadruped dimensions.
#include "ip_kernel_impl.h"
namespace cimpc {
CIMPC_DEFINE_MODEL(quadruped, 11, 8, 2, 4, 8)
}  // namespace cimpc
