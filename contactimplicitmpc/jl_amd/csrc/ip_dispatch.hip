// Dimension dispatch of the interior-point sweep kernel: one instantiation per model of the reference
// (src/dynamics/*/model.jl; point_foot_quadruped and centroidal_quadruped_box share the centroidal dimensions).
// A further model is ONE line here + a five-line ip_model_<name>.hip + its (nq, nu) pair in newton_kernels.hip
// (CIMPC_NQNU).  centroidal_quadruped_wall (ny = 48 exceeds the 32-lane group) has a 64-lane instantiation - one problem per
// wavefront, :configuration mode only (CIMPC_MODELS64, round 6); its :configurationforce mode keeps the runtime-dimension kernel.
#include "newton_state.h"

namespace cimpc {

#define CIMPC_MODELS(X)                      \
    X(pushbot, 2, 2, 2, 2, 4)                \
    X(hopper, 4, 2, 2, 1, 2)                 \
    X(quadruped, 11, 8, 2, 4, 8)             \
    X(flamingo, 9, 6, 2, 4, 8)               \
    X(centroidal, 18, 12, 3, 4, 16)          \
    X(hopper3d, 7, 3, 3, 1, 4)               \
    X(walledcartpole, 4, 1, 4, 2, 4)         \
    X(particle, 3, 3, 3, 1, 4)               \
    X(particle2d, 2, 2, 2, 1, 2)

#define CIMPC_MODELS64(X)                    \
    X(centroidal_wall, 18, 12, 3, 8, 32)

#define X(name, q, u, w, c, b)                                                               \
    int ip_launch_##name(int mode, const IpParams& p, int waves, hipStream_t s);   \
    int ip_callback_##name(int mode, const IpCallbackArgs& a, hipStream_t s);      \
    void ip_info_##name(int mode, KernelInfo* info);
CIMPC_MODELS(X)
#undef X
#define X(name, q, u, w, c, b)                                                               \
    int ip_launch_##name(int mode, const IpParams& p, int waves, hipStream_t s);   \
    void ip_info_##name(int mode, KernelInfo* info);
CIMPC_MODELS64(X)
#undef X
#define CIMPC_IS64(dm, q, u, w, c, b) ((dm)->mode == CIMPC_MODE_CONFIGURATION && (dm)->nq == q && (dm)->nu == u && (dm)->nw == w && (dm)->nc == c && (dm)->nb == b)

int ip_kernel_info(const cimpc_dims* dm, KernelInfo* info) {
#define X(name, q, u, w, c, b)                                                          \
    if (dm->nq == q && dm->nu == u && dm->nw == w && dm->nc == c && dm->nb == b) {      \
        ip_info_##name(dm->mode, info);                                                 \
        info->generic = 0;                                                              \
        return CIMPC_OK;                                                                \
    }
    CIMPC_MODELS(X)
#undef X
#define X(name, q, u, w, c, b)                                                          \
    if (CIMPC_IS64(dm, q, u, w, c, b)) { ip_info_##name(dm->mode, info); info->generic = 0; return CIMPC_OK; }
    CIMPC_MODELS64(X)
#undef X
    if (ip_generic_available(dm)) {      // any other model with nx, ny <= 64: runtime-dimension kernel
        ip_generic_info(dm, info);
        info->generic = 1;
        return CIMPC_OK;
    }
    return CIMPC_ERR_INVALID;
}

int launch_ip_sweep(const cimpc_dims* dm, const IpParams& p, int waves, hipStream_t s) {
#define X(name, q, u, w, c, b)                                                          \
    if (dm->nq == q && dm->nu == u && dm->nw == w && dm->nc == c && dm->nb == b)        \
        return ip_launch_##name(dm->mode, p, waves, s);
    CIMPC_MODELS(X)
#undef X
#define X(name, q, u, w, c, b) if (CIMPC_IS64(dm, q, u, w, c, b)) return ip_launch_##name(dm->mode, p, waves, s);
    CIMPC_MODELS64(X)
#undef X
    return launch_ip_generic(dm, p, s);
}

int launch_ip_callback(const cimpc_dims* dm, const IpCallbackArgs& a, hipStream_t s) {
#define X(name, q, u, w, c, b)                                                          \
    if (dm->nq == q && dm->nu == u && dm->nw == w && dm->nc == c && dm->nb == b)        \
        return ip_callback_##name(dm->mode, a, s);
    CIMPC_MODELS(X)
#undef X
    return CIMPC_ERR_INVALID;
}

// asynchronous single-launch Newton solve (newton_async_impl.h): :configuration mode, nq, nu <= 16
int async_launch_pushbot(const IpParams& p, const NewtonDev& S, int waves, int grid, hipStream_t s);
int async_launch_hopper(const IpParams& p, const NewtonDev& S, int waves, int grid, hipStream_t s);
int async_launch_quadruped(const IpParams& p, const NewtonDev& S, int waves, int grid, hipStream_t s);
int async_launch_flamingo(const IpParams& p, const NewtonDev& S, int waves, int grid, hipStream_t s);
int async_launch_centroidal(const IpParams& p, const NewtonDev& S, int waves, int grid, hipStream_t s);

bool newton_async_available(const cimpc_dims* dm) {
    if (dm->mode != CIMPC_MODE_CONFIGURATION || dm->H > 96) return false;
    return (dm->nq == 2 && dm->nu == 2 && dm->nw == 2 && dm->nc == 2 && dm->nb == 4) ||
           (dm->nq == 4 && dm->nu == 2 && dm->nw == 2 && dm->nc == 1 && dm->nb == 2) ||
           (dm->nq == 11 && dm->nu == 8 && dm->nw == 2 && dm->nc == 4 && dm->nb == 8) ||
           (dm->nq == 9 && dm->nu == 6 && dm->nw == 2 && dm->nc == 4 && dm->nb == 8) ||
           (dm->nq == 18 && dm->nu == 12 && dm->nw == 3 && dm->nc == 4 && dm->nb == 16);
}

int launch_newton_async(const cimpc_dims* dm, const IpParams& p, const NewtonDev& S, int waves, int grid, hipStream_t s) {
    if (!newton_async_available(dm)) return CIMPC_ERR_INVALID;
    if (dm->nq == 2) return async_launch_pushbot(p, S, waves, grid, s);
    if (dm->nq == 4) return async_launch_hopper(p, S, waves, grid, s);
    if (dm->nq == 18) return async_launch_centroidal(p, S, waves, grid, s);
    if (dm->nq == 9) return async_launch_flamingo(p, S, waves, grid, s);
    return async_launch_quadruped(p, S, waves, grid, s);
}

}  // namespace cimpc
