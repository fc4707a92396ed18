// Asynchronous single-launch Newton solve, hopper dimensions.
#include "newton_async_impl.h"
namespace cimpc {
CIMPC_DEFINE_ASYNC_MODEL(hopper, 4, 2, 2, 1, 2)
}  // namespace cimpc
