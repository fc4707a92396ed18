// Interior-point sweep kernel instantiation: centroidal_quadruped_wall (src/dynamics/centroidal_quadruped_wall/model.jl:204-207:
// nc = 8 contacts, ny = 2 nc + nb = 48 > 32) - 64-lane groups, ONE problem per wavefront (round 6), :configuration mode.
#include "ip_kernel_impl.h"
namespace cimpc {
CIMPC_DEFINE_MODEL64(centroidal_wall, 18, 12, 3, 8, 32)
}  // namespace cimpc
