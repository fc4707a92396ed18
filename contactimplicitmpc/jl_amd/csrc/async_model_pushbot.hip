// Asynchronous single-launch Newton solve, pushbot dimensions.
#include "newton_async_impl.h"
namespace cimpc {
CIMPC_DEFINE_ASYNC_MODEL(pushbot, 2, 2, 2, 2, 4)
}  // namespace cimpc
