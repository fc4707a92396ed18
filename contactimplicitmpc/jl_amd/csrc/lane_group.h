// Lane-group primitives for gfx950 (CDNA4, wave64).
//
// The small dense kernels of this library map ONE problem to a group of G lanes
// (G = 16 or 32) of a 64-wide wavefront, i.e. 4 or 2 problems per wave.  Vectors are
// "lane indexed" (element l lives in lane l of the group), matrices are either held
// one column / one row per lane in registers or streamed from LDS.  The only
// cross-lane traffic is (a) broadcast of one lane's value to its group and
// (b) all-reduce over the group.  For G = 16 a group is exactly one DPP row, so the
// broadcast is a single `v_mov_b64_dpp ... row_newbcast:K` and reductions use
// quad_perm / row_ror DPP moves; for G = 32 the cross-row step uses ds_bpermute.
//
// All lanes of a group must be active (EXEC) whenever these primitives execute:
// a DPP / bpermute read of a disabled lane does not return that lane's register.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <utility>

namespace cimpc {

template <int K, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (K < N) {
        f(std::integral_constant<int, K>{});
        static_for<K + 1, N>(f);
    }
}
// descending: K = N-1 ... 0
template <int K, class F>
__device__ __forceinline__ void static_rfor(F&& f) {
    if constexpr (K >= 0) {
        f(std::integral_constant<int, K>{});
        static_rfor<K - 1>(f);
    }
}

template <int G>
struct LaneGroup {
    static_assert(G == 16 || G == 32, "lane group is one or two DPP rows");

    // value of `v` held by lane K of the caller's group
    template <int K>
    static __device__ __forceinline__ double bcast(double v) {
        if constexpr (G == 16) {
            return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xF, 0xF, true);  // row_newbcast:K
        } else {
            // (measured alternatives for the 32-lane model: v_readlane + select is 50 % SLOWER - VALU -> SGPR -> VALU
            //  hazards; DPP row_newbcast + gfx950 v_permlane16_swap gives the same sweep time as this ds_bpermute)
            const int lane = (int)(threadIdx.x & 63);
            return __shfl(v, (lane & ~31) | K, 64);
        }
    }

    // DPP controls: quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_ror:n = 0x120+n
    static __device__ __forceinline__ double dpp_xor1(double v) {
        return __builtin_amdgcn_update_dpp(0.0, v, 0xB1, 0xF, 0xF, true);
    }
    static __device__ __forceinline__ double dpp_xor2(double v) {
        return __builtin_amdgcn_update_dpp(0.0, v, 0x4E, 0xF, 0xF, true);
    }
    static __device__ __forceinline__ double dpp_ror4(double v) {
        return __builtin_amdgcn_update_dpp(0.0, v, 0x124, 0xF, 0xF, true);
    }
    static __device__ __forceinline__ double dpp_ror8(double v) {
        return __builtin_amdgcn_update_dpp(0.0, v, 0x128, 0xF, 0xF, true);
    }

    static __device__ __forceinline__ double all_max(double v) {
        v = fmax(v, dpp_xor1(v));
        v = fmax(v, dpp_xor2(v));
        v = fmax(v, dpp_ror4(v));
        v = fmax(v, dpp_ror8(v));
        if constexpr (G == 32) v = fmax(v, __shfl_xor(v, 16, 64));
        return v;
    }
    static __device__ __forceinline__ double all_min(double v) {
        v = fmin(v, dpp_xor1(v));
        v = fmin(v, dpp_xor2(v));
        v = fmin(v, dpp_ror4(v));
        v = fmin(v, dpp_ror8(v));
        if constexpr (G == 32) v = fmin(v, __shfl_xor(v, 16, 64));
        return v;
    }
    // sum over the group; the result of lane 0 is re-broadcast so that every lane of the
    // group holds bit-identical data (rotation-based reductions associate differently per lane)
    static __device__ __forceinline__ double all_sum(double v) {
        v += dpp_xor1(v);
        v += dpp_xor2(v);
        v += dpp_ror4(v);
        v += dpp_ror8(v);
        if constexpr (G == 32) v += __shfl_xor(v, 16, 64);
        return bcast<0>(v);
    }
};

// 1/sqrt(x) and 1/x to ~1 ulp from the hardware seeds (v_rsq_f64 / v_rcp_f64) and two
// Newton steps - an order of magnitude fewer instructions than the IEEE sqrt + divide
// sequences hipcc emits (v_div_scale / v_div_fmas / v_div_fixup).
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    double e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    return y;
}
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}

}  // namespace cimpc
