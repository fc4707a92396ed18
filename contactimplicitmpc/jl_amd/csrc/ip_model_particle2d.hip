// Interior-point sweep kernel instantiation: particle_2D (src/dynamics/particle_2D/model.jl) dimensions (lock-step rounds; no single-launch kernel).
#include "ip_kernel_impl.h"
namespace cimpc {
CIMPC_DEFINE_MODEL(particle2d, 2, 2, 2, 1, 2)
}  // namespace cimpc
