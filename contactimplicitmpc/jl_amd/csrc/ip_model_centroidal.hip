// Interior-point sweep kernel instantiation: centroidal dimensions.
#include "ip_kernel_impl.h"
namespace cimpc {
CIMPC_DEFINE_MODEL(centroidal, 18, 12, 3, 4, 16)
}  // namespace cimpc
