// Interior-point sweep kernel instantiation: particle (src/dynamics/particle/model.jl:114-117) dimensions (lock-step rounds; no single-launch kernel).
#include "ip_kernel_impl.h"
namespace cimpc {
CIMPC_DEFINE_MODEL(particle, 3, 3, 3, 1, 4)
}  // namespace cimpc
