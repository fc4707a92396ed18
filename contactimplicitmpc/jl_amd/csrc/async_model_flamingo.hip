// Asynchronous single-launch Newton solve, flamingo dimensions.
#include "newton_async_impl.h"
namespace cimpc {
CIMPC_DEFINE_ASYNC_MODEL(flamingo, 9, 6, 2, 4, 8)
}  // namespace cimpc
