// Interior-point sweep kernel instantiation: walledcartpole (src/dynamics/walledcartpole/model.jl:143-147) dimensions (lock-step rounds; no single-launch kernel).
#include "ip_kernel_impl.h"
namespace cimpc {
CIMPC_DEFINE_MODEL(walledcartpole, 4, 1, 4, 2, 4)
}  // namespace cimpc
