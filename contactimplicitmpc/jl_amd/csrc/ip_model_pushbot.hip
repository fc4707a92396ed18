// Interior-point sweep kernel instantiation: pushbot dimensions.
#include "ip_kernel_impl.h"
namespace cimpc {
CIMPC_DEFINE_MODEL(pushbot, 2, 2, 2, 2, 4)
}  // namespace cimpc
