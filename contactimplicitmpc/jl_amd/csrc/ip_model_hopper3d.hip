// Interior-point sweep kernel instantiation: hopper_3D (src/dynamics/hopper_3D/model.jl:96-99) dimensions (lock-step rounds; no single-launch kernel).
#include "ip_kernel_impl.h"
namespace cimpc {
CIMPC_DEFINE_MODEL(hopper3d, 7, 3, 3, 1, 4)
}  // namespace cimpc
