// Interior-point sweep kernel instantiation: flamingo dimensions (src/dynamics/flamingo/model.jl:458-462).
#include "ip_kernel_impl.h"
namespace cimpc {
CIMPC_DEFINE_MODEL(flamingo, 9, 6, 2, 4, 8)
}  // namespace cimpc
