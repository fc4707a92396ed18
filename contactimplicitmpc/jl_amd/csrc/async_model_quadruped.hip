// Asynchronous single-launch Newton solve, quadruped dimensions.
#include "newton_async_impl.h"
namespace cimpc {
CIMPC_DEFINE_ASYNC_MODEL(quadruped, 11, 8, 2, 4, 8)
}  // namespace cimpc
