// Asynchronous single-launch Newton solve, centroidal dimensions: 32-lane
// interior-point groups, 24 x 24 KKT tiles; one workgroup per CU (512-register budget, 128 KB of LDS).
#include "newton_async_impl.h"
namespace cimpc {
CIMPC_DEFINE_ASYNC_MODEL(centroidal, 18, 12, 3, 4, 16)
}  // namespace cimpc
