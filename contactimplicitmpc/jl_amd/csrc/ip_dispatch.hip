// Dimension dispatch of the interior-point sweep kernel (one instantiation per model the
// reference ships dimensions for; SURVEY.md section 2 "Problem dimensions").
#include "cimpc_internal.h"

namespace cimpc {

#define CIMPC_MODELS(X)                      \
    X(pushbot, 2, 2, 2, 2, 4)                \
    X(hopper, 4, 2, 2, 1, 2)                 \
    X(quadruped, 11, 8, 2, 4, 8)             \
    X(centroidal, 18, 12, 3, 4, 16)

#define X(name, q, u, w, c, b)                                                               \
    int ip_launch_##name(int mode, const IpParams& p, int waves, hipStream_t s);   \
    void ip_info_##name(int mode, KernelInfo* info);
CIMPC_MODELS(X)
#undef X

int ip_kernel_info(const cimpc_dims* dm, KernelInfo* info) {
#define X(name, q, u, w, c, b)                                                          \
    if (dm->nq == q && dm->nu == u && dm->nw == w && dm->nc == c && dm->nb == b) {      \
        ip_info_##name(dm->mode, info);                                                 \
        return CIMPC_OK;                                                                \
    }
    CIMPC_MODELS(X)
#undef X
    return CIMPC_ERR_INVALID;
}

int launch_ip_sweep(const cimpc_dims* dm, const IpParams& p, int waves, hipStream_t s) {
#define X(name, q, u, w, c, b)                                                          \
    if (dm->nq == q && dm->nu == u && dm->nw == w && dm->nc == c && dm->nb == b)        \
        return ip_launch_##name(dm->mode, p, waves, s);
    CIMPC_MODELS(X)
#undef X
    return CIMPC_ERR_INVALID;
}

}  // namespace cimpc
