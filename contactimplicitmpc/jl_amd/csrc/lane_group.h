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

// Pins N values in registers at this point of the program with ONE empty statement (a barrier for code motion only: loads issued
// before it are waited for once, together, instead of one s_waitcnt per value).  Chunks of 16: an inline-assembly statement takes
// at most 30 operands.
template <int I0, int N, int... I>
__device__ __forceinline__ void pin_values_c(double (&a)[N], std::integer_sequence<int, I...>) {
    if constexpr (sizeof...(I) == 16) asm volatile("" : "+v"(a[I0 + 0]), "+v"(a[I0 + 1]), "+v"(a[I0 + 2]), "+v"(a[I0 + 3]), "+v"(a[I0 + 4]), "+v"(a[I0 + 5]), "+v"(a[I0 + 6]), "+v"(a[I0 + 7]), "+v"(a[I0 + 8]), "+v"(a[I0 + 9]), "+v"(a[I0 + 10]), "+v"(a[I0 + 11]), "+v"(a[I0 + 12]), "+v"(a[I0 + 13]), "+v"(a[I0 + 14]), "+v"(a[I0 + 15]));
    else if constexpr (sizeof...(I) == 8) asm volatile("" : "+v"(a[I0 + 0]), "+v"(a[I0 + 1]), "+v"(a[I0 + 2]), "+v"(a[I0 + 3]), "+v"(a[I0 + 4]), "+v"(a[I0 + 5]), "+v"(a[I0 + 6]), "+v"(a[I0 + 7]));
    else if constexpr (sizeof...(I) == 4) asm volatile("" : "+v"(a[I0 + 0]), "+v"(a[I0 + 1]), "+v"(a[I0 + 2]), "+v"(a[I0 + 3]));
    else if constexpr (sizeof...(I) == 2) asm volatile("" : "+v"(a[I0 + 0]), "+v"(a[I0 + 1]));
    else asm volatile("" : "+v"(a[I0 + 0]));
}
template <int I0 = 0, int N>
__device__ __forceinline__ void pin_values(double (&a)[N]) {
    if constexpr (I0 < N) {
        constexpr int R = N - I0, C = R >= 16 ? 16 : R >= 8 ? 8 : R >= 4 ? 4 : R >= 2 ? 2 : 1;
        pin_values_c<I0>(a, std::make_integer_sequence<int, C>{});
        pin_values<I0 + C>(a);
    }
}

template <int G>
struct LaneGroup {
    static_assert(G == 16 || G == 32 || G == 64, "lane group is one, two or four DPP rows (64: the whole wavefront, one problem per wave)");

    // value of `v` held by lane K of the caller's group
    template <int K>
    static __device__ __forceinline__ double bcast(double v) {
        if constexpr (G == 16) {
            return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + K, 0xF, 0xF, true);  // row_newbcast:K
        } else if constexpr (G == 64) {
            // one problem per wavefront (round 6, ny > 32): the source lane is wave-uniform - two v_readlane_b32, and the value
            // arrives in SCALAR registers (a free operand of the multiply-add that consumes it)
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane(__double2loint(v), K);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane(__double2hiint(v), K);
            return __hiloint2double((int)hi, (int)lo);
        } else {
            // 32-lane group = two DPP rows.  row_newbcast:(K & 15) puts lane (K & 15) of EVERY row into its own row; gfx950's
            // v_permlane16_swap then exchanges the odd rows of one operand with the even rows of the other: with both operands
            // = that value, the first result holds the even rows' value in both rows of a group, the second the odd rows' - the
            // broadcast of lane K for K < 16 / K >= 16.  Three VALU instructions and NO LDS round trip: what is left on this path
            // after round 4 are the scalar broadcasts inside dependency chains (back-substitution, 1/|a_k|, reductions), where the
            // ~100-cycle latency of ds_bpermute was the chain.  (Vector broadcasts - matrix-vector products, the column of the
            // MGS step - go through an LDS staging vector instead: IpSolver::stage.  Round 1-3 history: ds_bpermute for
            // everything; v_readlane + select 50 % slower; this DPP + swap form for EVERYTHING the same sweep time as ds_bpermute.)
            const double t = __builtin_amdgcn_update_dpp(0.0, v, 0x150 + (K & 15), 0xF, 0xF, true);
            const unsigned lo = (unsigned)__double2loint(t), hi = (unsigned)__double2hiint(t);
            const auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
            constexpr int S = K < 16 ? 0 : 1;
            return __hiloint2double((int)r1[S], (int)r0[S]);
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

    // max / min of two doubles as ONE instruction.  fmax / fmin compile to v_max_f64 / v_min_f64 PLUS a canonicalising
    // v_max_f64 x, x, x of every operand the compiler cannot prove quiet (IEEE mode: signalling NaNs) - in the butterfly below
    // that is the freshly moved operand of every step, 5 extra instructions per reduction, 30 per interior-point iteration.
    // No signalling NaN exists in these kernels (nothing loads raw bit patterns as doubles), and quiet NaNs behave the same.
    static __device__ __forceinline__ double max2(double a, double b) {
        double r;
        asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static __device__ __forceinline__ double min2(double a, double b) {
        double r;
        asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static __device__ __forceinline__ double all_max(double v) {
        v = max2(v, dpp_xor1(v));
        v = max2(v, dpp_xor2(v));
        v = max2(v, dpp_ror4(v));
        v = max2(v, dpp_ror8(v));
        if constexpr (G >= 32) v = max2(v, __shfl_xor(v, 16, 64));
        if constexpr (G == 64) v = max2(v, __shfl_xor(v, 32, 64));
        return v;
    }
    static __device__ __forceinline__ double all_min(double v) {
        v = min2(v, dpp_xor1(v));
        v = min2(v, dpp_xor2(v));
        v = min2(v, dpp_ror4(v));
        v = min2(v, dpp_ror8(v));
        if constexpr (G >= 32) v = min2(v, __shfl_xor(v, 16, 64));
        if constexpr (G == 64) v = min2(v, __shfl_xor(v, 32, 64));
        return v;
    }
    // sum over the group WITHOUT the re-broadcast: every lane holds the sum in its own association (rotation-based reductions
    // associate differently per lane) - for results that ONE designated lane keeps
    static __device__ __forceinline__ double group_sum(double v) {
        v += dpp_xor1(v);
        v += dpp_xor2(v);
        v += dpp_ror4(v);
        v += dpp_ror8(v);
        if constexpr (G >= 32) v += __shfl_xor(v, 16, 64);
        if constexpr (G == 64) v += __shfl_xor(v, 32, 64);
        return v;
    }
    // sum over the group; the result of lane 0 is re-broadcast so that every lane of the
    // group holds bit-identical data (rotation-based reductions associate differently per lane)
    static __device__ __forceinline__ double all_sum(double v) {
        v += dpp_xor1(v);
        v += dpp_xor2(v);
        v += dpp_ror4(v);
        v += dpp_ror8(v);
        if constexpr (G >= 32) v += __shfl_xor(v, 16, 64);
        if constexpr (G == 64) v += __shfl_xor(v, 32, 64);
        return bcast<0>(v);
    }
};

// ------------------------------------------------------------------------------------------------------------------
// Fused broadcast-FMA for 16-lane groups:  acc += bcast<K>(s) * t  in ONE instruction.
// gfx90a+ DP-ALU instructions accept a DPP source with the row_newbcast control (and only that one): the broadcast of
// lane K of every 16-lane row rides on src0 of `v_fmac_f64_dpp`.  hipcc does not form it from `update_dpp` + `fma`
// (it emits v_mov_b64_dpp + v_fmac_f64: two DP-ALU issue slots per multiply-add), hence the inline assembly.
//   * DPP hazard: a VGPR written by a VALU instruction may be read through DPP only two wait states later.  The
//     hazard recogniser does not look inside inline assembly (and register allocation may place a copy right in
//     front of a statement), so every statement opens with `s_nop 1`; statements carry 2..4 multiply-adds.
//   * EXEC: every lane of the row must be active (same rule as for the broadcasts above).
//   * Arithmetic: fmac(acc, a, b) = fma(a, b, acc) - the same roundings as the unfused form.
#define CIMPC_FMAC_DPP(acc, src, t, k) "v_fmac_f64_dpp %[" #acc "], %[" #src "], %[" #t "] row_newbcast:%[" #k "] row_mask:0xf bank_mask:0xf\n\t"

}  // namespace cimpc
#include "dpp16_gen.h"     // Dpp16Gen: statements of 1, 2, 3, 4, 8, 16 fused multiply-adds (scripts/gen_dpp16.py)
namespace cimpc {

// largest generated statement size that fits into N remaining multiply-adds (earlier chunks are multiples of four,
// so accumulator rotation k & 1 / k & 3 stays aligned)
template <int N, int MAX = 16>
constexpr int dpp_chunk() { return (N >= 16 && MAX >= 16) ? 16 : (N >= 8 && MAX >= 8) ? 8 : N >= 4 ? 4 : N; }

struct Dpp16 {
    // ---- a += bcast<K>(Q[r]) * Q[r], r = R0 .. R0+N-1 in ONE chain (see scripts/gen_dpp16.py: dotsN) -------------------------
    template <int K, int R0, int NQ, int... I>
    static __device__ __forceinline__ void dots_c(double& a, const double (&Q)[NQ], std::integer_sequence<int, I...>) {
        constexpr int C = sizeof...(I);
        if constexpr (C == 16) Dpp16Gen::dots16<K>(a, Q[R0 + I]...);
        else if constexpr (C == 8) Dpp16Gen::dots8<K>(a, Q[R0 + I]...);
        else if constexpr (C == 4) Dpp16Gen::dots4<K>(a, Q[R0 + I]...);
        else if constexpr (C == 3) Dpp16Gen::dots3<K>(a, Q[R0 + I]...);
        else if constexpr (C == 2) Dpp16Gen::dots2<K>(a, Q[R0 + I]...);
        else Dpp16Gen::dots1<K>(a, Q[R0 + I]...);
    }
    template <int K, int N, int R0 = 0, int NQ>
    static __device__ __forceinline__ void dots(double& a, const double (&Q)[NQ]) {
        if constexpr (N > 0) {
            constexpr int C = dpp_chunk<N>();
            dots_c<K, R0>(a, Q, std::make_integer_sequence<int, C>{});
            dots<K, N - C, R0 + C>(a, Q);
        }
    }
    // ---- Q[r] += bcast<K>(Q[r]) * c, then a += bcast<K+1>(Q[r]) * Q[r]: update of MGS step K + dot products of step K + 1 in one
    //      statement (N = 4, 8, 16: the whole column in one chunk)
    template <int K, int N, int... I>
    static __device__ __forceinline__ void selfdot_c(double& a, double (&Q)[N], double c, std::integer_sequence<int, I...>) {
        if constexpr (N == 16) Dpp16Gen::selfdot16<K>(a, Q[I]..., c);
        else if constexpr (N == 8) Dpp16Gen::selfdot8<K>(a, Q[I]..., c);
        else Dpp16Gen::selfdot4<K>(a, Q[I]..., c);
    }
    template <int N>
    static constexpr bool can_selfdot() { return N == 4 || N == 8 || N == 16; }
    template <int K, int N>
    static __device__ __forceinline__ void selfdot(double& a, double (&Q)[N], double c) {
        static_assert(can_selfdot<N>(), "one chunk");
        selfdot_c<K>(a, Q, c, std::make_integer_sequence<int, N>{});
    }
    // ---- Q[r] += bcast<K>(Q[r]) * c: rank-1 update of the columns by column K --------------------------------------------
    template <int K, int R0, int NQ, int... I>
    static __device__ __forceinline__ void self_c(double (&Q)[NQ], double c, std::integer_sequence<int, I...>) {
        constexpr int C = sizeof...(I);
        if constexpr (C == 16) Dpp16Gen::self16<K>(Q[R0 + I]..., c);
        else if constexpr (C == 8) Dpp16Gen::self8<K>(Q[R0 + I]..., c);
        else if constexpr (C == 4) Dpp16Gen::self4<K>(Q[R0 + I]..., c);
        else if constexpr (C == 3) Dpp16Gen::self3<K>(Q[R0 + I]..., c);
        else if constexpr (C == 2) Dpp16Gen::self2<K>(Q[R0 + I]..., c);
        else Dpp16Gen::self1<K>(Q[R0 + I]..., c);
    }
    template <int K, int N, int R0 = 0, int NQ>
    static __device__ __forceinline__ void self(double (&Q)[NQ], double c) {
        if constexpr (N > 0) {
            constexpr int C = dpp_chunk<N>();
            self_c<K, R0>(Q, c, std::make_integer_sequence<int, C>{});
            self<K, N - C, R0 + C>(Q, c);
        }
    }
    // ---- a[i] += bcast<i>(v) * t, i = 0..N-1: one entry t of this lane against N rows of v (outer-product update) ----------------
    template <int I0, int NA, int... I>
    static __device__ __forceinline__ void outer_c(double (&a)[NA], double v, double t, std::integer_sequence<int, I...>) {
        constexpr int C = sizeof...(I);
        if constexpr (C == 16) Dpp16Gen::outer16<I0>(a[I0 + I]..., v, t);
        else if constexpr (C == 8) Dpp16Gen::outer8<I0>(a[I0 + I]..., v, t);
        else if constexpr (C == 4) Dpp16Gen::outer4<I0>(a[I0 + I]..., v, t);
        else if constexpr (C == 3) Dpp16Gen::outer3<I0>(a[I0 + I]..., v, t);
        else if constexpr (C == 2) Dpp16Gen::outer2<I0>(a[I0 + I]..., v, t);
        else Dpp16Gen::outer1<I0>(a[I0 + I]..., v, t);
    }
    template <int N, int I0 = 0, int NA>
    static __device__ __forceinline__ void outer(double (&a)[NA], double v, double t) {
        if constexpr (N > 0) {
            constexpr int C = dpp_chunk<N>();
            outer_c<I0>(a, v, t, std::make_integer_sequence<int, C>{});
            outer<N - C, I0 + C>(a, v, t);
        }
    }
    // ---- acc[k & 1] += bcast<k>(v) * T(k), k = K0 .. K0+N-1: even / odd partial sums (the sensitivity product of the wide lane groups) ----
    template <int K0, class TF, int... I>
    static __device__ __forceinline__ void chain2_c(double (&a)[2], double v, TF& T, std::integer_sequence<int, I...>) {
        static_assert((K0 & 1) == 0, "chunks start on an even index");
        constexpr int C = sizeof...(I);
        if constexpr (C == 16) Dpp16Gen::lanes16_acc2<K0>(a[0], a[1], v, T(std::integral_constant<int, K0 + I>{})...);
        else if constexpr (C == 8) Dpp16Gen::lanes8_acc2<K0>(a[0], a[1], v, T(std::integral_constant<int, K0 + I>{})...);
        else if constexpr (C == 4) Dpp16Gen::lanes4_acc2<K0>(a[0], a[1], v, T(std::integral_constant<int, K0 + I>{})...);
        else if constexpr (C == 2) Dpp16Gen::lanes2_acc2<K0>(a[0], a[1], v, T(std::integral_constant<int, K0 + I>{})...);
        else static_assert(C == 16 || C == 8 || C == 4 || C == 2, "even chunk sizes");
    }
    template <int N, int K0 = 0, int MAX = 16, class TF>
    static __device__ __forceinline__ void chain2(double (&a)[2], double v, TF&& T) {
        if constexpr (N > 0) {
            constexpr int C0 = dpp_chunk<N, MAX>(), C = C0 == 3 ? 2 : C0;
            static_assert(C != 1, "an even number of terms per row-spread chunk");
            chain2_c<K0>(a, v, T, std::make_integer_sequence<int, C>{});
            chain2<N - C, K0 + C, MAX>(a, v, T);
        }
    }
    // ---- a[A0 + i] += bcast<B0 + i>(v) * t, i = 0..N-1: `outer` on a window of a longer register array (wide lane groups: chunk j of a column
    //      that every 16-lane row holds spread over its lanes - IpSolver::factorize)
    template <int A0, int B0, int NA, int... I>
    static __device__ __forceinline__ void outer_at_c(double (&a)[NA], double v, double t, std::integer_sequence<int, I...>) {
        constexpr int C = sizeof...(I);
        if constexpr (C == 16) Dpp16Gen::outer16<B0>(a[A0 + I]..., v, t);
        else if constexpr (C == 8) Dpp16Gen::outer8<B0>(a[A0 + I]..., v, t);
        else if constexpr (C == 4) Dpp16Gen::outer4<B0>(a[A0 + I]..., v, t);
        else if constexpr (C == 3) Dpp16Gen::outer3<B0>(a[A0 + I]..., v, t);
        else if constexpr (C == 2) Dpp16Gen::outer2<B0>(a[A0 + I]..., v, t);
        else Dpp16Gen::outer1<B0>(a[A0 + I]..., v, t);
    }
    template <int N, int A0, int B0 = 0, int NA>
    static __device__ __forceinline__ void outer_at(double (&a)[NA], double v, double t) {
        if constexpr (N > 0) {
            constexpr int C = dpp_chunk<N>();
            outer_at_c<A0, B0>(a, v, t, std::make_integer_sequence<int, C>{});
            outer_at<N - C, A0 + C, B0 + C>(a, v, t);
        }
    }
    // ---- s += bcast<k>(v) * T(k), k = 0..N-1 in this order: ONE chain, the association of a plain loop -----------------------
    template <int K0, class TF, int... I>
    static __device__ __forceinline__ void chain_c(double& s, double v, TF& T, std::integer_sequence<int, I...>) {
        constexpr int C = sizeof...(I);
        if constexpr (C == 16) Dpp16Gen::lanes16_acc1<K0>(s, v, T(std::integral_constant<int, K0 + I>{})...);
        else if constexpr (C == 8) Dpp16Gen::lanes8_acc1<K0>(s, v, T(std::integral_constant<int, K0 + I>{})...);
        else if constexpr (C == 4) Dpp16Gen::lanes4_acc1<K0>(s, v, T(std::integral_constant<int, K0 + I>{})...);
        else if constexpr (C == 3) Dpp16Gen::lanes3_acc1<K0>(s, v, T(std::integral_constant<int, K0 + I>{})...);
        else if constexpr (C == 2) Dpp16Gen::lanes2_acc1<K0>(s, v, T(std::integral_constant<int, K0 + I>{})...);
        else Dpp16Gen::lanes1_acc1<K0>(s, v, T(std::integral_constant<int, K0 + I>{})...);
    }
    // (MAX: largest statement - a statement holds all its operands in registers at once: 16 doubles are 32 VGPRs)
    template <int N, int K0 = 0, int MAX = 16, class TF>
    static __device__ __forceinline__ void chain(double& s, double v, TF&& T) {
        if constexpr (N > 0) {
            constexpr int C = dpp_chunk<N, MAX>();
            chain_c<K0>(s, v, T, std::make_integer_sequence<int, C>{});
            chain<N - C, K0 + C, MAX>(s, v, T);
        }
    }
    // ---- a += bcast<k>(v) * TA(k), c += bcast<k>(v) * TC(k), k = 0..N-1: two operators on one vector, sequential chains ---
    template <int K0, class TA, class TC, int... I>
    static __device__ __forceinline__ void pair_c(double& a, double& c, double v, TA& ta, TC& tc, std::integer_sequence<int, I...>) {
        constexpr int C = sizeof...(I);
        // (argument order of the generated statements: ta0, tc0, ta1, tc1, ...)
        if constexpr (C == 8) Dpp16Gen::pair8<K0>(a, c, v, ta(std::integral_constant<int, K0 + 0>{}), tc(std::integral_constant<int, K0 + 0>{}), ta(std::integral_constant<int, K0 + 1>{}), tc(std::integral_constant<int, K0 + 1>{}), ta(std::integral_constant<int, K0 + 2>{}), tc(std::integral_constant<int, K0 + 2>{}), ta(std::integral_constant<int, K0 + 3>{}), tc(std::integral_constant<int, K0 + 3>{}), ta(std::integral_constant<int, K0 + 4>{}), tc(std::integral_constant<int, K0 + 4>{}), ta(std::integral_constant<int, K0 + 5>{}), tc(std::integral_constant<int, K0 + 5>{}), ta(std::integral_constant<int, K0 + 6>{}), tc(std::integral_constant<int, K0 + 6>{}), ta(std::integral_constant<int, K0 + 7>{}), tc(std::integral_constant<int, K0 + 7>{}));
        else if constexpr (C == 4) Dpp16Gen::pair4<K0>(a, c, v, ta(std::integral_constant<int, K0 + 0>{}), tc(std::integral_constant<int, K0 + 0>{}), ta(std::integral_constant<int, K0 + 1>{}), tc(std::integral_constant<int, K0 + 1>{}), ta(std::integral_constant<int, K0 + 2>{}), tc(std::integral_constant<int, K0 + 2>{}), ta(std::integral_constant<int, K0 + 3>{}), tc(std::integral_constant<int, K0 + 3>{}));
        else if constexpr (C == 3) Dpp16Gen::pair3<K0>(a, c, v, ta(std::integral_constant<int, K0 + 0>{}), tc(std::integral_constant<int, K0 + 0>{}), ta(std::integral_constant<int, K0 + 1>{}), tc(std::integral_constant<int, K0 + 1>{}), ta(std::integral_constant<int, K0 + 2>{}), tc(std::integral_constant<int, K0 + 2>{}));
        else if constexpr (C == 2) Dpp16Gen::pair2<K0>(a, c, v, ta(std::integral_constant<int, K0 + 0>{}), tc(std::integral_constant<int, K0 + 0>{}), ta(std::integral_constant<int, K0 + 1>{}), tc(std::integral_constant<int, K0 + 1>{}));
        else Dpp16Gen::pair1<K0>(a, c, v, ta(std::integral_constant<int, K0>{}), tc(std::integral_constant<int, K0>{}));
    }
    template <int N, int K0 = 0, int MAX = 8, class TA, class TC>
    static __device__ __forceinline__ void pair(double& a, double& c, double v, TA&& ta, TC&& tc) {
        if constexpr (N > 0) {
            constexpr int C = dpp_chunk<N, MAX>();
            pair_c<K0>(a, c, v, ta, tc, std::make_integer_sequence<int, C>{});
            pair<N - C, K0 + C, MAX>(a, c, v, ta, tc);
        }
    }
    // back-substitution step K of an upper-triangular solve kept one ROW per lane (nr = -R[l,K], zero for K <= l):
    //   x_K = c_K * rdinv_K  (formed in every lane, lane K's value is the solution component);  c += bcast<K>(x) * nr
    template <int K>
    static __device__ __forceinline__ void backsub(double& c, double rdinv, double nr) {
        double x;
        asm("v_mul_f64 %[x], %[c], %[ri]\n\ts_nop 1\n\t" CIMPC_FMAC_DPP(c, x, nr, k) : [c] "+v"(c), [x] "=&v"(x) : [ri] "v"(rdinv), [nr] "v"(nr), [k] "n"(K));
    }
    // N (= 2, 3, 4) independent back-substitutions side by side: the N products first, then the N multiply-adds - from N = 3 on every
    // x_j is two instructions old when its multiply-add reads it through DPP, so the statement needs no wait states between them (one
    // chain alone: 17 clocks per step, four interleaved: 9 - scripts/ubench/valu_lat.hip).  Per chain the arithmetic of `backsub`.
    template <int K>
    static __device__ __forceinline__ void backsub4(double& c0, double& c1, double& c2, double& c3, double rdinv, double nr) {
        double x0, x1, x2, x3;
        asm("v_mul_f64 %[x0], %[c0], %[ri]\n\tv_mul_f64 %[x1], %[c1], %[ri]\n\tv_mul_f64 %[x2], %[c2], %[ri]\n\tv_mul_f64 %[x3], %[c3], %[ri]\n\t"
            CIMPC_FMAC_DPP(c0, x0, nr, k) CIMPC_FMAC_DPP(c1, x1, nr, k) CIMPC_FMAC_DPP(c2, x2, nr, k) CIMPC_FMAC_DPP(c3, x3, nr, k)
            : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2), [x3] "=&v"(x3)
            : [ri] "v"(rdinv), [nr] "v"(nr), [k] "n"(K));
    }
    template <int K>
    static __device__ __forceinline__ void backsub3(double& c0, double& c1, double& c2, double rdinv, double nr) {
        double x0, x1, x2;
        asm("v_mul_f64 %[x0], %[c0], %[ri]\n\tv_mul_f64 %[x1], %[c1], %[ri]\n\tv_mul_f64 %[x2], %[c2], %[ri]\n\t"
            CIMPC_FMAC_DPP(c0, x0, nr, k) CIMPC_FMAC_DPP(c1, x1, nr, k) CIMPC_FMAC_DPP(c2, x2, nr, k)
            : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2)
            : [ri] "v"(rdinv), [nr] "v"(nr), [k] "n"(K));
    }
    template <int K>
    static __device__ __forceinline__ void backsub2(double& c0, double& c1, double rdinv, double nr) {
        double x0, x1;
        asm("v_mul_f64 %[x0], %[c0], %[ri]\n\tv_mul_f64 %[x1], %[c1], %[ri]\n\ts_nop 0\n\t"
            CIMPC_FMAC_DPP(c0, x0, nr, k) CIMPC_FMAC_DPP(c1, x1, nr, k)
            : [c0] "+v"(c0), [c1] "+v"(c1), [x0] "=&v"(x0), [x1] "=&v"(x1)
            : [ri] "v"(rdinv), [nr] "v"(nr), [k] "n"(K));
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
// (Measured and NOT taken, scripts/ubench/rsq_check.hip: ONE third-order step y (1 + e (1/2 + 3/8 e)), e = 1 - x y^2, instead of the two
//  Newton steps - two instructions fewer per MGS step at the same 0.62-ulp error bound, but its results differ from the two-step
//  value in the last place on 17 % of the arguments, and with it the device leaves the CPU checker's discrete path on 1.6 % of the
//  20 480 full-size interior-point solves instead of 0.005 %: the two-step value is what keeps the iterates on the checker's path.)
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}

}  // namespace cimpc
