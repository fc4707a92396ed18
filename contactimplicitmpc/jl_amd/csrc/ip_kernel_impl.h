// Batched interior-point solve of the linearized contact-LCP knots: the data-parallel hot
// loop `implicit_dynamics!` (/root/reference/src/controller/implicit_dynamics.jl:156-192)
// with every callback the reference supplies to RoboDojo's interior_point_solve! fused
// into ONE gfx950 kernel:
//   rlin!                      linearized_solver.jl:364-373
//   rzlin! + schur_factorize!  linearized_solver.jl:378-399, schur.jl:80-88
//   MGS-QR factorize!/qr_solve!  qr.jl:113-158
//   linear_solve! (vector)     linearized_solver.jl:424-444, schur_solve! schur.jl:93-110
//   general_correction_term!, residual_violation, bilinear_violation   :401-418
//   linear_solve! (matrix rhs, sensitivities)                          :451-479
// and the IP iteration control frozen in DESIGN.md ("IP iteration spec"; RoboDojo 0.1.3
// owns it upstream and its source is not available).
//
// Mapping (CDNA4, wave64): one problem (rollout b, horizon position i) per group of
// G = 16 (ny <= 16) or 32 lanes -> 4 / 2 problems per wavefront.  A workgroup serves
// problems that share a reference knot: the knot's packed linearization table is
// staged once into LDS with coalesced 16-byte loads.  Per lane: lane-indexed vectors
// x, y1, y2, r, Delta; column l of the ny x ny Schur matrix / Q factor in registers;
// the R factor goes through a padded LDS tile (transpose) into one row per lane.
// Broadcasts are `v_mov_b64_dpp row_newbcast`, reductions DPP quad_perm/row_ror.
#pragma once
#include "cimpc_internal.h"
#include "lane_group.h"
#include "lin_table.h"

// wave priorities: the KKT recursion that runs next to the sweep (second stream, one wave per rollout) is the longer leg of
// most rounds since the sweep lost a third of its instructions: it gets the issue slots first (KKT 2 / sweep 0: 11.9 -> 11.6 ms)
#define CIMPC_SWEEP_PRIO 0      // (a constant of the build: the -D override is gone with the experiment it served)

namespace cimpc {

// -DCIMPC_SWEEP_PROF: per-wave shader-clock accounting of the lock-step sweep (diagnostic builds only; the timers cost ~10 %):
//   [0] wave lifetime  [1] barriers + knot pick  [2] table staging  [3] end-of-solve section  [4] pull section
//   [5] interior-point trips  [6] sensitivity trips  [7] waves  [8] pulls  [9] ip trips  [10] sens trips
#ifdef CIMPC_SWEEP_PROF
static __device__ unsigned long long g_sweep_prof[16];
#define SPROF_T() ((long long)__builtin_amdgcn_s_memtime())
#else
#define SPROF_T() 0ll
#endif

// WIDE_ (32-lane models only): the throughput build - see ip_queue_kernel
template <int NQ_, int NU_, int NW_, int NC_, int NB_, int MODE_, int WIDE_ = 0>
struct Model {
    static constexpr int NQ = NQ_, NU = NU_, NW = NW_, NC = NC_, NB = NB_, MODE = MODE_;
    static constexpr bool WIDE = WIDE_ != 0;
    using Wide = Model<NQ_, NU_, NW_, NC_, NB_, MODE_, 1>;
    static constexpr int NX = NQ, NY = 2 * NC + NB, NZ = NQ + 4 * NC + 2 * NB;
    static constexpr int NTH = 2 * NQ + NU + NW + 2, NTHS = 2 * NQ + NU;
    static constexpr int ND = MODE ? NQ + NC + NB : NQ;
    static constexpr int G = (NX <= 16 && NY <= 16) ? 16 : (NX <= 32 && NY <= 32) ? 32 : 64;
    static_assert(NX <= 64 && NY <= 64, "lane group holds at most 64 rows");
    // R-factor transposition buffer: lane = column produces row k of R at step k, lane l needs row l in its registers.
    // The whole [NY][G + 1] tile (conflict-free at stride G + 1), read back once at the end - except in the throughput build of the
    // 32-lane models (WIDE, round 4): a window of RROWS = 8 rows at stride 34 (16-byte aligned rows for ds_read_b128), read back by
    // its eight owner lanes every eight steps - 2.2 KB instead of 8.4 KB per problem, which is what lets a workgroup hold 16
    // problems (8 waves = two per SIMD) next to the 85 KB table of the centroidal model.
    static constexpr bool FULL_TILE = !(WIDE && !(NX <= 16 && NY <= 16)) && G != 64;      // (64-lane groups: R packed in LDS, see RLDS)
    static constexpr int RROWS = FULL_TILE ? NY : 8;
    static constexpr int RST_LD = FULL_TILE ? G + 1 : G + 2;   // padded row stride of the R tile
    static constexpr int DTN_LD = ((NTHS + G - 1) / G) * G;    // leading dimension of the delta^T nu products (IpParams::dtn)
    // Adjoint form of the sensitivity pass (round 5; configuration mode, where only the x rows of dz/dtheta are wanted):
    // nx transposed solves M^T a_i = (A^-1 B)[i, :]^T and one small product instead of nths solves with M - `sensitivities`
#ifndef CIMPC_SENS_ADJOINT
#define CIMPC_SENS_ADJOINT 1
#endif
    static constexpr int ADJ = (CIMPC_SENS_ADJOINT && MODE == 0) ? 1 : 0;
#define CIMPC_SENS_MAX 16      // (a constant of the build: the -D override is gone with the experiment it served)
    static constexpr int SENS_MAX = CIMPC_SENS_MAX;             // converged problems a group may defer
// (measured dead ends: C A^-1, A^-1, Dy1 rows register-resident per knot - the 76 extra VGPRs spill, sweep launch 0.30 -> 0.47 ms;
//  rows of R left in the LDS tile instead of registers for a third wave per SIMD - the spills stay, 0.35 -> 0.63 ms)
#define CIMPC_SENS_ILP 5      // (a constant of the build: the -D override is gone with the experiment it served)
#ifndef CIMPC_SENS_ILP32
#define CIMPC_SENS_ILP32 3
#endif
    static constexpr int SENS_ILP = (NX <= 16 && NY <= 16) ? CIMPC_SENS_ILP : CIMPC_SENS_ILP32;   // sensitivity columns solved side by side
    // per-problem LDS: the R tile and theta - theta0 SHARE their space (theta - theta0 lives from the pull of a problem to the two
    // dot products a few lines below it; the tile is scratch inside factorize), then the backlog of deferred sensitivities
    // Round 6: the throughput build of the 32-lane models keeps R IN LDS for the whole solve - the strict upper triangle
    // packed by column (entry (r, c), r < c, at c (c - 1) / 2 + r: NY (NY - 1) / 2 = 276 doubles for ny = 24, the size of the row window
    // it replaces) - and the back-substitutions read their row entry per step from there.  The 24 row registers per lane were what
    // the 256-register budget spilled: 466 MB of scratch writes per launch, 8-10 x the algorithmic traffic.  Adjoint models only
    // (:configuration): there the tile is free for R during the solves (the transposed-solve results travel through the staging
    // vectors, the columns of the product phase are parked in the tile AFTER the last solve).
    static constexpr bool RLDS = !FULL_TILE && ADJ != 0;
    static_assert(G != 64 || RLDS, "64-lane groups: :configuration mode only (the adjoint sensitivity pass keeps the tile free for R)");
    static constexpr int RTRI = NY * (NY - 1) / 2;
    static constexpr int TILE0 = RLDS ? RTRI : RROWS * RST_LD;
    static constexpr int TILE = TILE0 > NTH ? TILE0 : NTH;
    // ... then one 64-bit word per group: the clock value the running solve's time budget counts from (IpParams::budget_ticks)
    // ... then (32-lane groups) three staging vectors of G doubles: one for lane-indexed vectors that feed a matrix-vector
    // product, two (by step parity) for the column the MGS step broadcasts - IpSolver::stage / factorize
    static constexpr int OFF_BV = ((TILE + SENS_MAX / 2 + 1) + 1) & ~1;
#ifndef CIMPC_ADJ_ILP32
#define CIMPC_ADJ_ILP32 3
#endif
#ifndef CIMPC_ADJ_CB
#define CIMPC_ADJ_CB 4
#endif
#ifndef CIMPC_ADJ_ILP64
#define CIMPC_ADJ_ILP64 3
#endif
#ifndef CIMPC_ADJ_CB64
#define CIMPC_ADJ_CB64 2      // (four columns side by side: 105 spilled VGPRs, 368 B of scratch per lane; two: none)
#endif
    static constexpr int ADJ_ILP = (NX <= 16 && NY <= 16) ? 4 : G == 64 ? CIMPC_ADJ_ILP64 : CIMPC_ADJ_ILP32;      // transposed solves of the adjoint pass side by side
    static constexpr int NBV0 = CIMPC_SENS_ILP32 > 3 ? CIMPC_SENS_ILP32 : 3;
    static constexpr int NBV = (ADJ && ADJ_ILP > NBV0) ? ADJ_ILP : NBV0;      // staging vectors per 32-lane group
    static constexpr int BVEC = (G == 16) ? 0 : NBV * G;
    static constexpr int LDS_GROUP = OFF_BV + BVEC;  // doubles / problem (even)
};

__device__ __forceinline__ void wave_lds_fence() {
    // LDS traffic of one wavefront is processed in issue order; this only stops the
    // compiler from reordering across the hand-off between lanes of the same wave.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Per-lane state and operators of one interior-point problem (one lane group).
template <class M>
struct IpSolver {
    static constexpr int NX = M::NX, NY = M::NY, NTH = M::NTH, G = M::G;
    static constexpr LinLayout L{NX, NY, NTH, G, M::NTHS, M::ADJ};
    using LG = LaneGroup<G>;

    const double* tab;   // LDS: staged linearization table
    const double* ctab;  // the blocks a solve touches once (LinLayout::hot): = tab, or (64-lane models) the table in global memory
    double* Rst;         // LDS: [NY][G+1] R-factor transpose tile of this problem
    double* bv;          // LDS (32-lane groups): staging vectors [3][G], see stage() / factorize()
    int l;               // lane within the group
    int csl;             // (RLDS) start of column l of the packed R triangle: l (l - 1) / 2
    bool vx, vy;
    // per-lane constants of the knot
    double ry2, ry1d, caibd, rdyn0, rrst0, x0, y10, y20;
    // per-solve constants
    double tthdyn, tthrst, altl;
    // iterate, residual, direction
    double x, y1, y2, rdyn, rrst, rbil, Dx_, Dy1_, Dy2_;
    // factorization: column l of Q, row l of -R (strict upper part), 1/R[l,l], regularised y, 1/y1r
    double Qc[NY], Rr[NY], rdinv, y1r, y2r, iy1r;

    __device__ __forceinline__ void bind(const double* tab_, double* Rst_, int l_, const double* ctab_ = nullptr) {
        tab = tab_; Rst = Rst_; l = l_;
        ctab = ctab_ != nullptr ? ctab_ : tab_;
        bv = Rst_ + M::OFF_BV;
        csl = l_ * (l_ - 1) / 2;
        vx = l < NX; vy = l < NY;
        const double* tVec = tab + L.oVecL;
        ry2 = tVec[LinLayout::V_RY2 * G + l];
        ry1d = tVec[LinLayout::V_RY1D * G + l];
        caibd = tVec[LinLayout::V_CAIBD * G + l];
        rdyn0 = tVec[LinLayout::V_RDYN0 * G + l];
        rrst0 = tVec[LinLayout::V_RRST0 * G + l];
        x0 = tVec[LinLayout::V_X0 * G + l];
        y10 = tVec[LinLayout::V_Y10 * G + l];
        y20 = tVec[LinLayout::V_Y20 * G + l];
        rdinv = 0.0; y1r = y2r = iy1r = 1.0;
        tthdyn = tthrst = altl = 0.0;
    }

    // 32-lane groups: a lane-indexed vector that feeds a matrix-vector product is STAGED in LDS - one store per lane - and read
    // back with group-uniform addresses (broadcast reads, two entries per ds_read_b128): 1 + n/2 LDS instructions instead of the
    // 2 n ds_bpermute_b32 of rounds 1-3 (the centroidal sweep was LDS-bound: SQ_INSTS_LDS 64 M per launch against 68 M VALU,
    // the LDS pipe 40 % busy over the whole launch - profiles/r04/cent_pmc_before.csv).  Same operands, same summation order.
    __device__ __forceinline__ void stage(double v) const {
        wave_lds_fence();            // (LDS operations of a wavefront execute in order: earlier reads of bv are done)
        bv[l] = v;
        wave_lds_fence();
    }

    // 32-lane groups (round 6): a lane-indexed vector as two ROW-SPREAD chunks - entries 0..15 and 16..31 in the lanes of EVERY 16-lane row
    // of the group (two v_permlane16_swap: the even row's and the odd row's registers in both rows) - so that a matrix-vector product
    // takes its vector entry through DPP row_newbcast like the 16-lane form, without the LDS staging vector (a broadcast ds_read_b128
    // costs the LDS pipe what a full-width read costs, and the pipe is what bounds this form).  Same products in the same order.
    __device__ __forceinline__ void spread32(double v, double& c0, double& c1) const {
        const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
        const auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        c0 = __hiloint2double((int)r1[0], (int)r0[0]);
        c1 = __hiloint2double((int)r1[1], (int)r0[1]);
    }
    // acc += sum_k v_k T(k), k = 0 .. N-1 ascending (T: callable on std::integral_constant<int, k>)
    template <int N, class TF>
    __device__ __forceinline__ void mv32(double& acc, double v, TF&& T) const {
        static_assert(G == 32 && N <= 32, "two row-spread chunks");
        double c0, c1;
        spread32(v, c0, c1);
        // (statements of four: the throughput build runs at 256 registers - sixteen table operands at once spilled 333 of them)
        Dpp16::chain<(N < 16 ? N : 16), 0, 4>(acc, c0, [&](auto kc) { return T(kc); });
        if constexpr (N > 16) Dpp16::chain<N - 16, 0, 4>(acc, c1, [&](auto kc) { return T(std::integral_constant<int, 16 + decltype(kc)::value>{}); });
    }
    // a += sum_k v_k TA(k), c += sum_k v_k TC(k): two operators on one vector
    template <int N, class TA, class TC>
    __device__ __forceinline__ void pair32(double& a, double& c, double v, TA&& ta, TC&& tc) const {
        static_assert(G == 32 && N <= 32, "two row-spread chunks");
        double c0, c1;
        spread32(v, c0, c1);
        Dpp16::pair<(N < 16 ? N : 16), 0, 4>(a, c, c0, [&](auto kc) { return ta(kc); }, [&](auto kc) { return tc(kc); });
        if constexpr (N > 16)
            Dpp16::pair<N - 16, 0, 4>(a, c, c1, [&](auto kc) { return ta(std::integral_constant<int, 16 + decltype(kc)::value>{}); },
                                [&](auto kc) { return tc(std::integral_constant<int, 16 + decltype(kc)::value>{}); });
    }

    // rlin! (linearized_solver.jl:364-373), same association as the reference expression
    __device__ __forceinline__ void residual(double kappa) {
        const double* tDx = tab + L.oDx; const double* tRx = tab + L.oRx;
        const double* tDy1 = tab + L.oDy1; const double* tRy1 = tab + L.oRy1;
        const double dx = x - x0, dy1 = y1 - y10, dy2 = y2 - y20;
        double a = 0.0, c = 0.0, bb = 0.0, e = 0.0;
        if constexpr (G == 16) {       // broadcast fused into the multiply-add (lane_group.h: Dpp16), same summation order
            Dpp16::pair<NX>(a, c, dx, [&](auto kc) { return tDx[decltype(kc)::value * G + l]; }, [&](auto kc) { return tRx[decltype(kc)::value * G + l]; });
            Dpp16::pair<NY>(bb, e, dy1, [&](auto kc) { return tDy1[decltype(kc)::value * G + l]; }, [&](auto kc) { return tRy1[decltype(kc)::value * G + l]; });
        } else if constexpr (G == 32) {
            pair32<NX>(a, c, dx, [&](auto kc) { return tDx[decltype(kc)::value * G + l]; }, [&](auto kc) { return tRx[decltype(kc)::value * G + l]; });
            pair32<NY>(bb, e, dy1, [&](auto kc) { return tDy1[decltype(kc)::value * G + l]; }, [&](auto kc) { return tRy1[decltype(kc)::value * G + l]; });
        } else {
            stage(dx);
            static_for<0, NX>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const double v = bv[k];
                a = fma(tDx[k * G + l], v, a);
                c = fma(tRx[k * G + l], v, c);
            });
            stage(dy1);
            static_for<0, NY>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const double v = bv[k];
                bb = fma(tDy1[k * G + l], v, bb);
                e = fma(tRy1[k * G + l], v, e);
            });
        }
        rdyn = ((rdyn0 + a) + bb) + tthdyn;
        rrst = ((((rrst0 + c) + e) + ry2 * dy2) + tthrst) + altl;
        rbil = vy ? (y1 * y2 - kappa) : 0.0;
    }
    __device__ __forceinline__ double r_violation() const { return LG::all_max(fmax(fabs(rdyn), fabs(rrst))); }
    __device__ __forceinline__ double k_violation() const { return LG::all_max(fabs(rbil)); }

    // rzlin! + schur_factorize! + MGS factorize!.  Right-looking order: per column exactly
    // the arithmetic of the reference's left-looking loop (qr.jl:113-137); dot products use
    // one running sum (the order of the reference's loop), 1/|a_k| comes from v_rsq_f64 + two Newton steps on the broadcast
    // self-product of column k.
    // TR (adjoint sensitivity pass, Model::ADJ): factorizes the TRANSPOSE of the Schur matrix (lane l reads row l of the block
    // instead of column l; its padded leading dimension keeps both free of bank conflicts) - qr_solve then solves with M^T.
    template <bool TR = false>
    __device__ __forceinline__ void factorize(double reg) {
        const double* tW = tab + L.oW;
        // lane index of the 3 NY lane compares below (l == r, l == k, l > k), opaque per call: as loop invariants the 48 masks were
        // hoisted out of every loop, overflowed the scalar register file and came back through v_readlane pairs at every use -
        // one v_cmp where it is needed is cheaper than two v_readlane
        int lq = l;
        asm volatile("" : "+v"(lq));
        y1r = fmax(y1, reg);
        y2r = fmax(y2, reg);
        iy1r = fast_rcp(y1r);
        const double dd = ry2 * y2r * iy1r;
        // (D - CAiB)[r, l]: the column of the table block, its diagonal entry replaced.  Every entry is LOADED first and selected
        // afterwards: written as one conditional expression per entry, the compiler turned the select into control flow around the
        // load - per entry two exec-mask flips, a branch and a wait for ITS load: 16 exposed LDS round trips and ~160 instructions, a
        // quarter of the factorization's time on a lone wavefront (scripts/dbg/ubench_ip.py: 5.55 k -> 4.2 k clocks)
        const double dval = (ry1d - dd) - caibd;
        double wcol[NY];
        static_for<0, NY>([&](auto ic) {
            constexpr int r = decltype(ic)::value;
            wcol[r] = TR ? tW[(vy ? l : 0) * L.ldw + r] : tW[r * L.ldw + l];      // Ry1[r,l] - CAiB[r,l]   (r != l)
        });
        pin_values(wcol);      // (the loads stay where they are)
        static_for<0, NY>([&](auto ic) {
            constexpr int r = decltype(ic)::value;
            Qc[r] = (lq == r) ? dval : wcol[r];
        });
        // Column l stays UNNORMALISED in Qc (q_l = Qc * rdinv); the projection coefficients are
        // r_kj = (a_k . a_j) / |a_k| and the update a_j -= (r_kj / |a_k|) a_k - the same numbers as
        // q_k = a_k/|a_k|, r_kj = q_k . a_j, a_j -= r_kj q_k up to one rounding, 16 multiplies per
        // step cheaper and without a lane-divergent branch.
        // 32-lane groups: the column the step broadcasts goes through LDS - its owner stores it (one lane per group), every lane
        // reads it back with group-uniform addresses.  Column k + 1 is final once step k has updated it, so its owner stores it
        // at the END of step k into the other of two buffers: the store's latency hides behind the rest of the step.
        [[maybe_unused]] double* bc = bv + G;      // [2][G]
        if constexpr (G != 16) {
            wave_lds_fence();
            if (lq == 0) static_for<0, NY>([&](auto ic) { constexpr int r = decltype(ic)::value; bc[r] = Qc[r]; });
        }
        // (16-lane groups with the column in one DPP chunk: the dot products of step k + 1 ride in the statement of step k's update -
        //  `dnext` carries them into the next step)
        constexpr bool FUSED = G == 16 && Dpp16::can_selfdot<NY>();
        [[maybe_unused]] double dnext = 0.0;
        if constexpr (FUSED) Dpp16::dots<0, NY>(dnext, Qc);
        static_for<0, NY>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            double acc[1] = {0.0};
            constexpr int NCH = (NY + 15) / 16;
            [[maybe_unused]] double akc[NCH];      // wide groups: this lane's share of column k (entry (l & 15) of every chunk of 16 rows)
            if constexpr (FUSED) {
                acc[0] = dnext;
            } else if constexpr (G == 16) {
                // a_k . a_l: column k rides on the DPP source of the multiply-add (no broadcast registers).  ONE chain (round 6): a
                // dependent multiply-add issues as fast as an independent one on this pipe, the four partial sums of rounds 1-5 only
                // cost three zero-initialisations and three adds per step (lane_group.h: Dpp16::dots) - and a plain running sum is
                // what the reference's loop forms (qr.jl:113-137)
                Dpp16::dots<k, NY>(acc[0], Qc);
            } else {
                wave_lds_fence();
                // Round 6: every 16-lane ROW picks the column up spread over its lanes - lane i of a row holds entries i, 16 + i, ... (one
                // ds_read_b64 per chunk of 16 rows) - and the multiply-adds take their entry through DPP row_newbcast, as the 16-lane form
                // does: 2-3 LDS reads per step and wave instead of NY / 2 broadcast ds_read_b128 (a broadcast costs the LDS pipe what a
                // full-width read costs: the 32-lane factorization slowed down with every wave added to the CU, scripts/dbg/ubench_ip.py).
                // Same products, same order: bit-identical.
                const double* src = bc + (k & 1) * G;
                static_for<0, NCH>([&](auto jc) {
                    constexpr int j = decltype(jc)::value, n = NY - 16 * j < 16 ? NY - 16 * j : 16;
                    akc[j] = src[16 * j + ((lq & 15) < n ? (lq & 15) : 0)];
                });
                static_for<0, NCH>([&](auto jc) {
                    constexpr int j = decltype(jc)::value, n = NY - 16 * j < 16 ? NY - 16 * j : 16;
                    Dpp16::chain<n>(acc[0], akc[j], [&](auto ic) { return Qc[16 * j + decltype(ic)::value]; });      // (one running sum here too: the reference loop's order)
                });
            }
            const double dot = acc[0];     // a_k . a_l
            // |a_k|^2 is lane k's own dot product (there a_k[r] * a_k[r], the same sum a separate
            // norm loop would form): one broadcast instead of 16 multiply-adds per step
            const double invk = fast_rsqrt(LG::template bcast<k>(dot));
            rdinv = (lq == k) ? invk : rdinv;
            // -r_kl = -(dot * invk), formed with the sign on the multiplier (exact) and kept NEGATED from here on: the tile then holds
            // -R, which is what the back-substitution adds - the 16 sign flips per factorization after the transposed read are gone
            double nrk = dot * -invk;
            nrk = ((lq > k) && vy) ? nrk : -0.0;       // (-0.0: the masked entries keep the sign they had as -(+0.0))
            const double ncoef = nrk * invk;            // -(r_kl / |a_k|); zero in lanes <= k: column k itself stays as it is during the update
            if constexpr (FUSED) {
                if constexpr (k + 1 < NY) { dnext = 0.0; Dpp16::selfdot<k, NY>(dnext, Qc, ncoef); }      // a_l -= (r_kl / |a_k|) a_k ; a_{k+1} . a_l
                // (the last column's update would change nothing: every coefficient is masked to zero at k = NY - 1)
            } else if constexpr (G == 16) {
                Dpp16::self<k, NY>(Qc, ncoef);       // a_l -= (r_kl / |a_k|) a_k
            } else {
                static_for<0, NCH>([&](auto jc) {
                    constexpr int j = decltype(jc)::value, n = NY - 16 * j < 16 ? NY - 16 * j : 16;
                    Dpp16::outer_at<n, 16 * j>(Qc, akc[j], ncoef);
                });
                if constexpr (k + 1 < NY) {       // column k + 1 is final: its owner publishes it for the next step
                    double* dst = bc + ((k + 1) & 1) * G;
                    if (lq == k + 1) static_for<0, NY>([&](auto ic) { constexpr int r = decltype(ic)::value; dst[r] = Qc[r]; });
                }
            }
            if constexpr (M::RLDS) {
                if ((lq > k) && vy) Rst[csl + k] = nrk;      // -R[k,l] -> column l, position k of the packed triangle (stays in LDS)
            } else if constexpr (M::FULL_TILE) {
                Rst[k * M::RST_LD + l] = nrk;  // -R[k,l], l > k (zeros elsewhere)
            } else {
                // window of RROWS rows: row k goes to slot k % RROWS; when the window is full (or the factorization ends) the
                // owner lanes of its rows read them back - -R[l,j] (stored negated): the back-substitution adds
                Rst[(k % M::RROWS) * M::RST_LD + l] = nrk;
                if constexpr ((k % M::RROWS) == M::RROWS - 1 || k == NY - 1) {
                    constexpr int k0 = (k / M::RROWS) * M::RROWS;
                    wave_lds_fence();
                    if (lq >= k0 && lq <= k) {
                        const double* rrow = Rst + (lq - k0) * M::RST_LD;
                        static_for<0, NY>([&](auto jc) { constexpr int j = decltype(jc)::value; Rr[j] = rrow[j]; });
                    }
                    wave_lds_fence();
                }
            }
        });
        if constexpr (M::RLDS) {
            wave_lds_fence();                  // the triangle is complete: the back-substitutions read it (r_entry)
        } else if constexpr (M::FULL_TILE) {
            wave_lds_fence();
            static_for<0, NY>([&](auto kc) {   // row l of R (transpose through the LDS tile)
                constexpr int k = decltype(kc)::value;
                Rr[k] = vy ? Rst[l * M::RST_LD + k] : 0.0;       // -R[l,k] (stored negated): the back-substitution adds
            });
            wave_lds_fence();
        } else {
            if (!vy) static_for<0, NY>([&](auto kc) { Rr[decltype(kc)::value] = 0.0; });     // (lanes beyond NY own no row)
        }
    }

    // -R[l, K] as the back-substitution adds it (zero for K <= l): the row register, or (RLDS) the packed triangle in LDS - lanes
    // l < K read K consecutive doubles of column K
    template <int K>
    __device__ __forceinline__ double r_entry() const {
        if constexpr (M::RLDS) {
            const double v = Rst[K * (K - 1) / 2 + ((l < K) ? l : 0)];
            return (l < K) ? v : -0.0;
        } else return Rr[K];
    }

    // t = R^-1 Q^T rhs  (qr_solve!, qr.jl:142-158); rhs lane-indexed
    __device__ __forceinline__ double qr_solve(double rhs) const {
        double acc = 0.0;
        if constexpr (G == 16) {
            Dpp16::chain<NY>(acc, rhs, [&](auto rc) { return Qc[decltype(rc)::value]; });      // (one chain: see factorize)
        } else {
            // (32-lane groups: this product keeps the staging vector - with the row-spread form here AND in schur_solve the throughput
            //  build spilled 320 registers; of the two placements this one measured better: iteration 40.2 k -> 36.7 k clocks at eight waves)
            stage(rhs);
            static_for<0, NY>([&](auto ic) {
                constexpr int r = decltype(ic)::value;
                acc = fma(Qc[r], bv[r], acc);
            });
        }
        double c = acc * rdinv;   // (Q^T rhs)_l, Q = Qc * rdinv
        // back-substitution, row l of R in this lane: x_k = c_k / R[k,k] broadcast, c_l -= R[l,k] x_k (k > l).  Lane l's c is
        // final once step k = l + 1 is done (R[l,k] = 0 for k <= l), so x_l = c * rdinv after the loop - the same product
        // the reference forms at step l (qr.jl:150-157).
        if constexpr (G == 16) {
            // (step 0 would add R[l, 0] x_0 = 0 to every lane: skipped)
            static_rfor<NY - 1>([&](auto kc) { if constexpr (decltype(kc)::value > 0) Dpp16::backsub<decltype(kc)::value>(c, rdinv, Rr[decltype(kc)::value]); });
        } else {
            static_rfor<NY - 1>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (k > 0) {
                    const double xk = LG::template bcast<k>(c * rdinv);
                    c = fma(r_entry<k>(), xk, c);
                }
            });
        }
        return c * rdinv;
    }

    // 16-lane groups: N = 2 .. 4 right-hand sides side by side (the transposed solves of the adjoint sensitivity pass) - per right-hand
    // side the arithmetic of qr_solve, the back-substitution chains interleaved (Dpp16::backsubN)
    template <int N>
    __device__ __forceinline__ void qr_solve_n16(const double (&rhs)[N], double (&t)[N]) const {
        static_assert(G == 16 && N >= 2 && N <= 4, "two to four chains");
        double c[N];
        static_for<0, N>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            double a = 0.0;
            Dpp16::chain<NY>(a, rhs[j], [&](auto rc) { return Qc[decltype(rc)::value]; });
            c[j] = a * rdinv;
        });
        static_rfor<NY - 1>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k == 0) return;      // (R[l, 0] = 0 in every lane)
            else if constexpr (N == 4) Dpp16::backsub4<k>(c[0], c[1], c[2], c[3], rdinv, Rr[k]);
            else if constexpr (N == 3) Dpp16::backsub3<k>(c[0], c[1], c[2], rdinv, Rr[k]);
            else Dpp16::backsub2<k>(c[0], c[1], rdinv, Rr[k]);
        });
        static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; t[j] = c[j] * rdinv; });
    }

    // schur_solve! (schur.jl:93-110): returns temp; x = Ai*(u + B*temp), y = -temp
    // PRE: `v` already is the right-hand side of the QR, CAi*u - v (the sensitivity pass reads it from the table: LinLayout::oGs)
    template <bool PRE = false>
    __device__ __forceinline__ double schur_solve(double u, double v, double& xs) const {
        const double* tCAi = tab + L.oCAi; const double* tAi = tab + L.oAi; const double* tDy1 = tab + L.oDy1;
        double bq[1] = {0.0}, w[1] = {0.0}, xx[1] = {0.0};
        if constexpr (G == 16) {      // (one chain per operator: see factorize)
            // (measured and not taken: every operand of the three products requested ahead of the triangular solve / of the residual's
            //  four products - the extra live registers cost more in copies than the hidden LDS round trips give: iteration of a lone
            //  wave 7.94 k -> 8.44 k clocks)
            if constexpr (!PRE) Dpp16::chain<NX>(bq[0], u, [&](auto kc) { return tCAi[decltype(kc)::value * G + l]; });
            const double t = qr_solve(PRE ? v : bq[0] - v);
            Dpp16::chain<NY>(w[0], t, [&](auto kc) { return tDy1[decltype(kc)::value * G + l]; });
            const double ww = u + w[0];
            Dpp16::chain<NX>(xx[0], ww, [&](auto kc) { return tAi[decltype(kc)::value * G + l]; });
            xs = xx[0];
            return t;
        }
        if constexpr (G == 32) {
            if constexpr (!PRE) mv32<NX>(bq[0], u, [&](auto kc) { return tCAi[decltype(kc)::value * G + l]; });
            const double t = qr_solve(PRE ? v : bq[0] - v);
            mv32<NY>(w[0], t, [&](auto kc) { return tDy1[decltype(kc)::value * G + l]; });
            const double ww = u + w[0];
            mv32<NX>(xx[0], ww, [&](auto kc) { return tAi[decltype(kc)::value * G + l]; });
            xs = xx[0];
            return t;
        }
        if constexpr (!PRE) {
            stage(u);
            static_for<0, NX>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                bq[0] = fma(tCAi[k * G + l], bv[k], bq[0]);
            });
        }
        const double t = qr_solve(PRE ? v : bq[0] - v);
        stage(t);
        static_for<0, NY>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            w[0] = fma(tDy1[k * G + l], bv[k], w[0]);
        });
        const double ww = u + w[0];
        stage(ww);
        static_for<0, NX>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            xx[0] = fma(tAi[k * G + l], bv[k], xx[0]);
        });
        xs = xx[0];
        return t;
    }

    // ---- 32-lane groups, sensitivity pass: N right-hand sides side by side (N <= 3: the staging vector and the two MGS column
    // buffers, idle here).  The columns of r_theta are independent; solving them together reads every operator entry (Q column,
    // Dy1, Ai, the row of R) once for all, stages the vectors with one hand-over and interleaves the back-substitution chains.
    // Per column the arithmetic is that of schur_solve<true> / qr_solve - same operands, same order: bit-identical columns.
    template <int N>
    __device__ __forceinline__ void stage_n(const double (&v)[N]) const {
        static_assert(N <= M::NBV, "one staging vector per right-hand side");
        wave_lds_fence();
        static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; bv[j * G + l] = v[j]; });
        wave_lds_fence();
    }
    template <int N>
    __device__ __forceinline__ void qr_solve_n(const double (&rhs)[N], double (&t)[N]) const {
        double a[N];
        static_for<0, N>([&](auto jc) { a[decltype(jc)::value] = 0.0; });
        if constexpr (G == 32) {
            static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; mv32<NY>(a[j], rhs[j], [&](auto rc) { return Qc[decltype(rc)::value]; }); });
        } else {
            stage_n<N>(rhs);
            static_for<0, NY>([&](auto ic) {
                constexpr int r = decltype(ic)::value;
                static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; a[j] = fma(Qc[r], bv[j * G + r], a[j]); });
            });
        }
        double c[N];
        static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; c[j] = a[j] * rdinv; });
        static_rfor<NY - 1>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if constexpr (k == 0) return;      // (R[l, 0] = 0 in every lane)
            const double rk = r_entry<k>();
            static_for<0, N>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const double xk = LG::template bcast<k>(c[j] * rdinv);
                c[j] = fma(rk, xk, c[j]);
            });
        });
        static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; t[j] = c[j] * rdinv; });
    }
    // (g: the pre-multiplied right-hand sides of the QR, LinLayout::oGs)
    template <int N>
    __device__ __forceinline__ void schur_solve_n(const double (&u)[N], const double (&g)[N], double (&t)[N], double (&xs)[N]) const {
        const double* tAi = tab + L.oAi; const double* tDy1 = tab + L.oDy1;
        qr_solve_n<N>(g, t);
        double w[N], x[N], ww[N];
        static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; w[j] = x[j] = 0.0; });
        if constexpr (G == 32) {
            static_for<0, N>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                mv32<NY>(w[j], t[j], [&](auto kc) { return tDy1[decltype(kc)::value * G + l]; });
                ww[j] = u[j] + w[j];
                mv32<NX>(x[j], ww[j], [&](auto kc) { return tAi[decltype(kc)::value * G + l]; });
                xs[j] = x[j];
            });
            return;
        }
        stage_n<N>(t);
        static_for<0, NY>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const double d = tDy1[k * G + l];
            static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; w[j] = fma(d, bv[j * G + k], w[j]); });
        });
        static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; ww[j] = u[j] + w[j]; });
        stage_n<N>(ww);
        static_for<0, NX>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            const double a_ = tAi[k * G + l];
            static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; x[j] = fma(a_, bv[j * G + k], x[j]); });
        });
        static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; xs[j] = x[j]; });
    }

    // ---- adjoint form of the sensitivity pass (Model::ADJ; after factorize<true>, the QR of M^T) ---------------------------
    // dx/dtheta = A^-1 rthdyn + (A^-1 B) M^-1 Gs: the same x rows schur_solve! (schur.jl:93-110) gives column by column
    // (linearized_solver.jl:451-479), with A2 = (A^-1 B) M^-1 formed ROW by row - row i is the solution of M^T a = (A^-1 B)[i, :]^T,
    // nx solves instead of nths.  Every solve leaves entry k of its row in lane k; the rows go through the R tile (RROWS rows per
    // pass) to their owner lanes: lane i < nx ends with A2[k] = entry (i, k).
    template <int N>
    __device__ __forceinline__ void adjoint_rows(int i0, double* dst) const {
        const double* tB = ctab + L.oAiB + i0 * NY + (vy ? l : 0);
        double b[N], z[N];
        static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; b[j] = vy ? tB[j * NY] : 0.0; });
        if constexpr (G == 16) {
            if constexpr (N >= 2 && N <= 4) qr_solve_n16<N>(b, z);
            else static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; z[j] = qr_solve(b[j]); });
        } else qr_solve_n<N>(b, z);
        static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; dst[j * M::RST_LD + l] = z[j]; });
    }
    // (RLDS: the tile holds R during the solves - a trip's rows go to their owner lanes through the staging vectors instead)
    template <int N>
    __device__ __forceinline__ void adjoint_trip(int i0, double (&A2)[NY]) const {
        const double* tB = ctab + L.oAiB + i0 * NY + (vy ? l : 0);
        double b[N], z[N];
        static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; b[j] = vy ? tB[j * NY] : 0.0; });
        qr_solve_n<N>(b, z);
        wave_lds_fence();
        static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; bv[j * G + l] = z[j]; });
        wave_lds_fence();
        const int j = l - i0;
        if (j >= 0 && j < N) {
            const double* row = bv + j * G;
            static_for<0, NY>([&](auto kc) { constexpr int k = decltype(kc)::value; A2[k] = row[k]; });
        }
    }
    __device__ __forceinline__ void adjoint_factor(double (&A2)[NY]) const {
        constexpr int AIL = M::ADJ_ILP;      // solves side by side (32-lane groups: one staging vector each)
        if constexpr (M::RLDS) {
            static_for<0, NY>([&](auto kc) { A2[decltype(kc)::value] = 0.0; });
#pragma unroll 1
            for (int i0 = 0; i0 + AIL <= NX; i0 += AIL) adjoint_trip<AIL>(i0, A2);
            if constexpr (NX % AIL != 0) adjoint_trip<NX % AIL>((NX / AIL) * AIL, A2);
            wave_lds_fence();
            return;
        }
        static_for<0, (NX + M::RROWS - 1) / M::RROWS>([&](auto pc) {
            constexpr int r0 = decltype(pc)::value * M::RROWS, n = NX - r0 < M::RROWS ? NX - r0 : M::RROWS;
            wave_lds_fence();
#pragma unroll 1
            for (int i = 0; i + AIL <= n; i += AIL) adjoint_rows<AIL>(r0 + i, Rst + i * M::RST_LD);
            if constexpr (n % AIL != 0) adjoint_rows<n % AIL>(r0 + (n / AIL) * AIL, Rst + (n / AIL) * AIL * M::RST_LD);
            wave_lds_fence();
            const bool own = l >= r0 && l < r0 + n;
            const double* row = Rst + (own ? l - r0 : 0) * M::RST_LD;
            static_for<0, NY>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (r0 == 0) A2[k] = own ? row[k] : 0.0;
                else A2[k] = own ? row[k] : A2[k];
            });
        });
        wave_lds_fence();
    }

    // linear_solve!(Delta, rz, r) (linearized_solver.jl:424-444)
    __device__ __forceinline__ void linear_solve() {
        const double u = rdyn;
        const double v = vy ? (rrst - ry2 * rbil * iy1r) : 0.0;
        const double t = schur_solve(u, v, Dx_);
        Dy1_ = -t;
        Dy2_ = vy ? ((rbil - y2r * Dy1_) * iy1r) : 0.0;
    }
    __device__ __forceinline__ double step_length(double tau) const {
        double a = 1.0;
        if (vy && Dy1_ > 0.0) a = fmin(a, tau * y1 / Dy1_);
        if (vy && Dy2_ > 0.0) a = fmin(a, tau * y2 / Dy2_);
        return LG::all_min(a);
    }

    // ONE interior-point iteration (DESIGN.md "IP iteration spec").  Returns true when the step
    // length stalled (the solve fails).  Convergence / iteration-budget checks are the caller's.
    __device__ __forceinline__ bool iterate(const cimpc_ip_opts& o, double kc_floor, double tau_floor, double& reg, double& r_vio, double& k_vio) {
        reg = (k_vio < o.kappa_reg) ? k_vio * o.gamma_reg : 0.0;
        factorize(reg);
        linear_solve();                                   // predictor
        const double a_aff = step_length(1.0);
        const double mu = LG::all_sum(vy ? y1 * y2 : 0.0) / (double)NY;
        const double mu_aff =
            LG::all_sum(vy ? (y1 - a_aff * Dy1_) * (y2 - a_aff * Dy2_) : 0.0) / (double)NY;
        double sg = fmin(fmax(mu_aff / mu, 0.0), 1.0);
        sg = sg * sg * sg;
        const double kc = fmax(sg * mu, kc_floor);                 // = o.kappa_tol / o.undercut (IpParams::kc_floor)
        // corrector residual: rdyn, rrst unchanged (same z), rbil = y1*y2 - kc + Dy1*Dy2
        rbil = vy ? ((y1 * y2 - kc) + Dy1_ * Dy2_) : 0.0;
        linear_solve();                                   // corrector
        const double vm = fmax(r_vio, k_vio);
        const double tau = fmax(tau_floor, 1.0 - vm * vm);         // tau_floor = 1 - o.eps_min
        const double alpha = step_length(tau);
        if (alpha < o.stall_alpha) return true;           // [spec] stall exit: jammed on the boundary
        x -= alpha * Dx_;
        y1 = vy ? (y1 - alpha * Dy1_) : 1.0;
        y2 = vy ? (y2 - alpha * Dy2_) : 1.0;
        double k_c = 0.0, r_c = 0.0, back = alpha;
        for (int s = 1; s <= o.max_ls; ++s) {
            residual(0.0);
            k_c = k_violation();
            r_c = r_violation();
            if (r_c <= r_vio || k_c <= k_vio) break;
            back *= o.ls_scale;                           // alpha * ls_scale^s
            x += back * Dx_;
            y1 = vy ? (y1 + back * Dy1_) : 1.0;
            y2 = vy ? (y2 + back * Dy2_) : 1.0;
        }
        k_vio = k_c;
        r_vio = r_c;
        return false;
    }
};

// one value of lane 0 of the group to the whole group
template <int G>
__device__ __forceinline__ int group_bcast0(int v) {
    if constexpr (G == 16) return __builtin_amdgcn_update_dpp(0, v, 0x150, 0xF, 0xF, true);
    else if constexpr (G == 64) return __builtin_amdgcn_readfirstlane(v);
    else return __shfl(v, (int)(threadIdx.x & 63) & ~31, 64);
}

template <class M>
__device__ __forceinline__ void stage_table(double* tab, const double* src_tab, int knot, int tid) {
    constexpr LinLayout L(M::NX, M::NY, M::NTH, M::G, M::NTHS, M::ADJ);
    const double2* src = reinterpret_cast<const double2*>(src_tab + (size_t)knot * L.size);
    double2* dst = reinterpret_cast<double2*>(tab);
    if constexpr (L.hot == L.size) {
        for (int k = tid; k < L.size / 2; k += (int)blockDim.x) dst[k] = src[k];   // 16 B per lane, coalesced
    } else {      // 64-lane models: the operators of the iteration and the vectors only (LinLayout::hot)
        static_assert(L.oRthDyn % 2 == 0 && L.oVec % 2 == 0, "16-byte copies");
        for (int k = tid; k < L.oRthDyn / 2; k += (int)blockDim.x) dst[k] = src[k];
        for (int k = tid; k < LinLayout::V_COUNT * L.G / 2; k += (int)blockDim.x) dst[L.oVecL / 2 + k] = src[L.oVec / 2 + k];
    }
}

// ----------------------------------------------------------------------------------------
// Sensitivities of one converged problem: differentiate_solution!, dz = -(rz^-1 rth) at z*,
// reg = max(reg, kappa_tol*gamma_reg); only the consumed block (rows 1:nd, columns q0,q1,u1 -
// implicit_dynamics.jl:84-86) is computed.  Always executed by the lane group that solved the
// problem (z* is re-read by the very lanes that stored it, so no cross-workgroup hand-off).
// ----------------------------------------------------------------------------------------
// One problem of evaluation slot sb is finished (its d / dz / status are written).  Lock-step
// rounds only count; the asynchronous solve hands the rollout to the residual stage when the last
// problem of the last outstanding slot of its line-search batch completes.
template <bool ASYNC>
__device__ __forceinline__ void problem_done(const IpParams& p, int sb, int l, bool leader = true) {
    if constexpr (ASYNC) xfence(p.A.flags);             // release this group's d / dz / status stores
    if (l == 0 && leader) {
        const int old = atomicAdd(&p.Q.done_count[sb], 1);
        if constexpr (ASYNC) {
            if (old == p.H - 1) {
                xfence(p.A.flags);
                const int b = sb / p.slots;
                if (atomicSub(&p.A.evals_left[b], 1) == 1) {
                    xfence(p.A.flags);
                    aq_push(p.A.rq_items, p.A.rq_tail, b);
                    wake_job(p.A, b);
                }
            }
        }
    }
}

// Split form (round 3): when a wave holds fewer pending problems than lane groups, `nparts` groups take the SAME problem -
// each repeats the factorization (the groups of a wave run in lock step, so the copies cost nothing) and solves its own
// share [NTHS*part/nparts, NTHS*(part+1)/nparts) of the independent right-hand sides; part 0 reports the problem.
template <class M, bool ASYNC>
__device__ __forceinline__ void sensitivities(const IpParams& p, IpSolver<M>& S, const double* tab, int prob, int l, int part = 0, int nparts = 1) {
    constexpr int NX = M::NX, NY = M::NY, NTH = M::NTH, NTHS = M::NTHS, ND = M::ND, G = M::G;
    constexpr int NC = M::NC, NB = M::NB;
    constexpr LinLayout L(NX, NY, NTH, G, M::NTHS, M::ADJ);
    constexpr int PS = 2 * NX + 4 * NY + 4;
    using LG = LaneGroup<G>;
    const bool vx = S.vx, vy = S.vy;
    const size_t pi = (size_t)prob;
    const double* ps = p.pstate + pi * PS;
    int lg = l;                          // (lane index of the global-memory addresses: opaque, see serve_knot)
    asm volatile("" : "+v"(lg));
    S.x = vx ? ps[lg] : 0.0;
    S.y1 = vy ? ps[NX + lg] : 1.0;
    S.y2 = vy ? ps[NX + NY + lg] : 1.0;
    const double reg = LG::template bcast<0>((l == 0) ? ps[PS - 2] : 0.0);
    S.template factorize<M::ADJ != 0>(fmax(reg, p.reg_floor));          // = o.kappa_tol * o.gamma_reg (IpParams::reg_floor)
    double* dzo = p.dz + pi * (size_t)(NTHS * ND);
    // delta^T nu of every column (IpParams::dtn).  The columns of a chunk are parked in the R tile (idle once the rows of R sit
    // in registers), then lane j sums column j with the SAME multiply-add chain the decision stage would run on the stored
    // block, s = fma(dz[k, j], nu[k], s), k = 0 .. nd-1 - bit-identical to reading the block back - at nd LDS reads + nd
    // broadcast-FMAs per chunk and lane instead of a 16-lane reduction per column (measured: +4.5 % sweep time for that form).
    constexpr int ILP = M::SENS_ILP;       // independent right-hand sides per trip: their triangular-solve chains interleave
    constexpr int CH0 = M::TILE / ND, CH1 = CH0 < G ? CH0 : G, CH = (CH1 / ILP) * ILP;      // columns per chunk
    static_assert(CH >= ILP, "the R tile holds at least one trip of sensitivity columns");
    const bool want = p.dtn != nullptr;          // (wave-uniform: a kernel parameter)
    double* scr = S.Rst;
    double nux = 0.0;
    [[maybe_unused]] double nuy = 0.0;
    if (want) {
        const double* nv = p.nu + pi * ND;
        nux = vx ? xld<ASYNC>(nv + lg) : 0.0;
        if constexpr (M::MODE == CIMPC_MODE_CONFIGURATIONFORCE) nuy = (l < NC + NB) ? xld<ASYNC>(nv + NX + lg) : 0.0;
    }
    auto column = [&](int c, int cc) {
        const double u = S.ctab[L.oRthDyn + c * G + l];
        const double g = S.ctab[L.oGs + c * G + l];          // CAi * rthdyn[:, c] - rthrst[:, c], a constant of the knot (lin_table.h)
        double xs;
        const double t = S.template schur_solve<true>(u, g, xs);
        if (vx) xst<ASYNC>(dzo + c * ND + lg, -xs);
        if constexpr (M::MODE == CIMPC_MODE_CONFIGURATIONFORCE) {
            if (l < NC + NB) xst<ASYNC>(dzo + c * ND + NX + lg, t);   // -(S.y) = +temp
        }
        if (want) {
            if (vx) scr[cc * ND + l] = -xs;
            if constexpr (M::MODE == CIMPC_MODE_CONFIGURATIONFORCE) { if (l < NC + NB) scr[cc * ND + NX + l] = t; }
        }
    };
    // 32-lane groups: columns c .. c + ILP - 1 in one pass (IpSolver::schur_solve_n)
    [[maybe_unused]] auto column_n = [&](int c, int cc) {
        if constexpr (G != 16) {
            double u[ILP], g[ILP], t[ILP], xs[ILP];
            static_for<0, ILP>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                u[j] = S.ctab[L.oRthDyn + (c + j) * G + l];
                g[j] = S.ctab[L.oGs + (c + j) * G + l];
            });
            S.template schur_solve_n<ILP>(u, g, t, xs);
            static_for<0, ILP>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (vx) xst<ASYNC>(dzo + (c + j) * ND + lg, -xs[j]);
                if constexpr (M::MODE == CIMPC_MODE_CONFIGURATIONFORCE) { if (l < NC + NB) xst<ASYNC>(dzo + (c + j) * ND + NX + lg, t[j]); }
                if (want) {
                    if (vx) scr[(cc + j) * ND + l] = -xs[j];
                    if constexpr (M::MODE == CIMPC_MODE_CONFIGURATIONFORCE) { if (l < NC + NB) scr[(cc + j) * ND + NX + l] = t[j]; }
                }
            });
        }
    };
    const int lo = nparts > 1 ? (NTHS * part) / nparts : 0, hi = nparts > 1 ? (NTHS * (part + 1)) / nparts : NTHS;
    if constexpr (M::ADJ) {
        // Adjoint form: A2 = (A^-1 B) M^-1 once per problem (nx transposed solves), then column c of dx/dtheta is
        // K0[:, c] + A2 Gs[:, c] - a row of the table against a row in registers, Gs read with group-uniform addresses.
        double A2[NY];
        S.adjoint_factor(A2);
        if constexpr (L.gst != 0) {
            // 16-lane groups: one COLUMN c of dx/dtheta per lane.  Entry (i, c) = K0[i, c] + sum_k A2[i, k] Gs[k, c]: row k of Gs sits in
            // this lane's register g[k] (the table stores the block by row for these models), A2[i, k] rides on the DPP operand of the
            // multiply-add from lane i - nx accumulators per lane, no LDS traffic in the product, every lane busy (the row-per-lane
            // form kept 11 of 16).  The signs are folded into the operands (-K0, -A2): the accumulators hold the stored values.
            // delta^T nu of the lane's column: the chain s = fma(dz[k, c], nu[k], s), k = 0 .. nx-1, the decision stage would run on
            // the stored block (IpParams::dtn) - nu[k] on the DPP operand.
            static_for<0, NY>([&](auto kc) { constexpr int k = decltype(kc)::value; A2[k] = -A2[k]; });
            const double* tG = S.ctab + L.oGs;
            const double* tK = S.ctab + L.oK0;
#pragma unroll 1
            for (int cb = lo; cb < hi; cb += G) {
                const int c = cb + lg;
                const bool vc = c < hi;
                const int cq = vc ? c : lo;
                double g[NY], acc[NX];
                static_for<0, NY>([&](auto kc) { constexpr int k = decltype(kc)::value; g[k] = tG[k * NTHS + cq]; });
                static_for<0, NX>([&](auto ic) { constexpr int i = decltype(ic)::value; acc[i] = -tK[cq * NX + i]; });
                static_for<0, NY>([&](auto kc) { constexpr int k = decltype(kc)::value; Dpp16::outer<NX>(acc, A2[k], g[k]); });
                if (vc) static_for<0, NX>([&](auto ic) { constexpr int i = decltype(ic)::value; xst<ASYNC>(dzo + (size_t)c * ND + i, acc[i]); });
                if (want) {
                    double s_ = 0.0;
                    Dpp16::chain<NX>(s_, nux, [&](auto ic) { return acc[decltype(ic)::value]; });
                    if (vc) xst<ASYNC>(p.dtn + pi * (size_t)M::DTN_LD + c, s_);
                }
            }
            problem_done<ASYNC>(p, prob / p.H, l, part == 0);
            return;
        }
        constexpr int CHA = CH1;
        const double* tK0 = S.ctab + L.oK0 + (vx ? l : 0);
#pragma unroll 1
        for (int c0 = lo; c0 < hi; c0 += CHA) {
            const int n = hi - c0 < CHA ? hi - c0 : CHA;
            // CB columns side by side, two partial sums each: 2 CB independent multiply-add chains (a wave of the 32-lane latency
            // build has its SIMD to itself - the chains are the time)
            constexpr int CB = M::WIDE ? 2 : G == 64 ? CIMPC_ADJ_CB64 : CIMPC_ADJ_CB;      // (throughput build of the 32-lane models: 256 registers, two waves hide each other)
            auto columns = [&](auto nb, int cc) {
                constexpr int N = decltype(nb)::value;
                const double* g = S.ctab + L.oGs + (c0 + cc) * G;      // (32-lane groups: the block is stored by column, lin_table.h)
                double a[N][2];
                static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; a[j][0] = tK0[(c0 + cc + j) * NX]; a[j][1] = 0.0; });
                if constexpr (G == 32 && (NY % 2) == 0 && NY > 16) {
                    // round 6: the column of Gs is picked up spread over every 16-lane row (two ds_read_b64 instead of NY / 2 broadcast
                    // ds_read_b128) and reaches the multiply-adds through DPP - same products, same even / odd partial sums
                    static_for<0, N>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        const double g0 = g[j * G + (l & 15)], g1 = g[j * G + 16 + ((l & 15) < NY - 16 ? (l & 15) : 0)];
                        Dpp16::chain2<16, 0, 8>(a[j], g0, [&](auto kc) { return A2[decltype(kc)::value]; });
                        Dpp16::chain2<NY - 16, 0, 8>(a[j], g1, [&](auto kc) { return A2[16 + decltype(kc)::value]; });
                    });
                } else
                static_for<0, NY>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; a[j][k & 1] = fma(A2[k], g[j * G + k], a[j][k & 1]); });
                });
                static_for<0, N>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const double mx = -(a[j][0] + a[j][1]);
                    if (vx) xst<ASYNC>(dzo + (c0 + cc + j) * ND + lg, mx);
                    if (want && vx) scr[(cc + j) * ND + l] = mx;
                });
            };
            int cc = 0;
#pragma unroll 1
            for (; cc + CB <= n; cc += CB) columns(std::integral_constant<int, CB>{}, cc);
#pragma unroll 1
            for (; cc < n; ++cc) columns(std::integral_constant<int, 1>{}, cc);
            if (want) {
                wave_lds_fence();
                const double* col = scr + (l < n ? l : 0) * ND;
                double s_ = 0.0;
                static_for<0, NX>([&](auto kc) { constexpr int k = decltype(kc)::value; s_ = fma(col[k], LG::template bcast<k>(nux), s_); });
                if (l < n) xst<ASYNC>(p.dtn + pi * (size_t)M::DTN_LD + c0 + lg, s_);
                wave_lds_fence();
            }
        }
        problem_done<ASYNC>(p, prob / p.H, l, part == 0);
        return;
    }
    constexpr int ILP2 = ILP > 2 ? ILP - 1 : 1;     // second interleave width: a share of 7 / 8 columns runs as 4 + 3 / 4 + 4
    const bool narrow = nparts > 1 && (hi - lo) % ILP != 0;
#pragma unroll 1
    for (int c0 = lo; c0 < hi; c0 += CH) {
        const int n = hi - c0 < CH ? hi - c0 : CH;
        int cc = 0;
        if (!narrow) {
#pragma unroll 1
            for (; cc + ILP <= n; cc += ILP) {
                if constexpr (G != 16) column_n(c0 + cc, cc);
                else static_for<0, ILP>([&](auto jc) { column(c0 + cc + decltype(jc)::value, cc + decltype(jc)::value); });
            }
        }
        if constexpr (ILP2 > 1) {
#pragma unroll 1
            for (; cc + ILP2 <= n; cc += ILP2) {
                static_for<0, ILP2>([&](auto jc) { column(c0 + cc + decltype(jc)::value, cc + decltype(jc)::value); });
            }
        }
        if constexpr (ILP2 > 2) {
            if (cc + ILP2 - 1 <= n) {
                static_for<0, ILP2 - 1>([&](auto jc) { column(c0 + cc + decltype(jc)::value, cc + decltype(jc)::value); });
                cc += ILP2 - 1;
            }
        }
        for (; cc < n; ++cc) column(c0 + cc, cc);
        if (want) {
            wave_lds_fence();
            const double* col = scr + (l < n ? l : 0) * ND;       // (every lane runs the broadcasts; lanes beyond the chunk discard)
            double s_ = 0.0;
            static_for<0, NX>([&](auto kc) { constexpr int k = decltype(kc)::value; s_ = fma(col[k], LG::template bcast<k>(nux), s_); });
            if constexpr (M::MODE == CIMPC_MODE_CONFIGURATIONFORCE)
                static_for<0, NC + NB>([&](auto kc) { constexpr int k = decltype(kc)::value; s_ = fma(col[NX + k], LG::template bcast<k>(nuy), s_); });
            if (l < n) xst<ASYNC>(p.dtn + pi * (size_t)M::DTN_LD + c0 + lg, s_);
            wave_lds_fence();
        }
    }
    problem_done<ASYNC>(p, prob / p.H, l, part == 0);
}

// ----------------------------------------------------------------------------------------
// Remaining-work proportional knot pick (WG-uniform result in *s_knot, -1 = no work anywhere):
// workgroup b takes the knot holding quantile b/gridDim of the problems not yet pulled, so the
// workgroups spread over the knots like the work does - initially and after every hop.
#define CIMPC_PICK_MAXK 256      // (a constant of the build: the -D override is gone with the experiment it served)
constexpr int PICK_MAXK = CIMPC_PICK_MAXK;
__device__ __forceinline__ int pick_knot(const IpParams& p, int* s_rem, int* s_total, int* s_knot, int tid, int wg, int nwg) {
    const int K = p.Q.K, par = p.Q.par;
    if (tid == 0) *s_total = 0;
    __syncthreads();
    {
        int part = 0;
        for (int k = tid; k < K; k += (int)blockDim.x) {
            int rem = aload(qcount(p.Q, par, k)) - aload(qhead(p.Q, k));
            rem = rem > 0 ? rem : 0;
            if (k < PICK_MAXK) s_rem[k] = rem;
            part += rem;
        }
        if (part > 0) atomicAdd(s_total, part);
    }
    __syncthreads();
    if (tid == 0) {
        const int total = *s_total;
        int pick = -1;
        if (total > 0) {
            const long long target = ((long long)wg * total) / (long long)nwg;
            long long acc = 0;
            for (int k = 0; k < K; ++k) {
                const int rem = k < PICK_MAXK ? s_rem[k] : max(0, aload(qcount(p.Q, par, k)) - aload(qhead(p.Q, k)));
                if (rem > 0) { pick = k; acc += rem; if (acc > target) break; }
            }
        }
        *s_knot = pick;
    }
    __syncthreads();
    return *s_knot;
}

// Serve one knot: stage its table, then every lane group pulls problems of the knot until the queue
// is empty (ASYNC: momentarily empty).  All groups of a wave run the same iteration body, a group
// whose solve ends (converged / failed / parked) finalises it and pulls the next problem while its
// neighbours keep iterating.  Converged problems wait in a per-group backlog for a wave-uniform
// sensitivity trip.  Must be entered by the whole workgroup; ends with every wave drained.
template <class M, bool ASYNC>
__device__ __forceinline__ void serve_knot(const IpParams& p, double* smem, int knot, int tid, [[maybe_unused]] long long* sp = nullptr) {
    constexpr int NX = M::NX, NY = M::NY, NTH = M::NTH, ND = M::ND, G = M::G;
    constexpr int NC = M::NC, NB = M::NB;
    constexpr LinLayout L(NX, NY, NTH, G, M::NTHS, M::ADJ);
    constexpr int PS = 2 * NX + 4 * NY + 4;
    double* tab = smem;
    const int K = p.Q.K, cap = p.Q.cap, par = p.Q.par;
    const int grp = tid / G;
    const int l = tid % G;
    double* Rst = smem + L.hot + (size_t)grp * M::LDS_GROUP;   // [NY][G+1]
    double* dth = Rst;                                           // [NTH]  (aliases the tile, see Model::LDS_GROUP)
    // converged problems whose sensitivities are pending: ONE list per wave (kept behind the tile of the wave's first group), so
    // that a sensitivity trip is full as soon as the wave holds one problem per group, whoever solved them
    constexpr int GPW = 64 / G;                                  // lane groups per wave
    static_assert(M::SENS_MAX >= 2 * GPW, "the wave's list holds a trip's worth of problems plus one trip of new arrivals");
    const int gw = (tid & 63) / G;
    int* backlog = reinterpret_cast<int*>(smem + L.hot + (size_t)(grp - gw) * M::LDS_GROUP + M::TILE);
    int nback = 0;                                               // wave-uniform
    const bool vx = l < NX, vy = l < NY;
    const cimpc_ip_opts o = p.o;
    // per-solve time budget (cimpc_ip_opts::max_time, policy.jl:9,61): `tbase` = clock value at which the solve's running time was
    // zero - kept in LDS, read once per trip, and only when a budget is set (wave-uniform kernel parameter)
    long long* tbase = reinterpret_cast<long long*>(smem + L.hot + (size_t)grp * M::LDS_GROUP + M::TILE + M::SENS_MAX / 2);
    const bool timed = p.budget_ticks > 0;

    [[maybe_unused]] const long long sp_t0 = SPROF_T();
    const int n = ASYNC ? 1 : *qcount(p.Q, par, knot);
    stage_table<M>(tab, p.tab, knot, tid);
    __syncthreads();
#ifdef CIMPC_SWEEP_PROF
    if (sp) sp[2] += SPROF_T() - sp_t0;
#endif
    const int* items = p.Q.items + ((size_t)par * K + knot) * cap;
    int* head = qhead(p.Q, knot);
    [[maybe_unused]] const int* tailp = qcount(p.Q, par, knot);
    IpSolver<M> S;
    const double* ctab = (L.hot == L.size) ? tab : p.tab + (size_t)knot * L.size;      // blocks touched once per solve (64-lane models: global memory)
    S.bind(tab, Rst, l, ctab);        // caches the per-lane constants of this knot's table
    bool have = false, exhausted = false, stalled = false;
    int prob = 0, iters = 0, done_here = 0;
    [[maybe_unused]] int idle_trips = 0, dbg_trips = 0, dbg_act = 0;
    [[maybe_unused]] long long dbg_tpop = 0, dbg_tfence = 0;
    [[maybe_unused]] long long dbg_pop[4] = {0, 0, 0, 0};
    [[maybe_unused]] const bool dbg_on = ASYNC && p.A.dbg != nullptr;      // diagnostics only with CIMPC_ASYNC_DEBUG
    double reg = 0.0, r_vio = 0.0, k_vio = 0.0, qinit = 0.0;
    [[maybe_unused]] int drain_raw = 0;       // (lock-step launches) workgroups that have left: the load is issued at the top of a trip and
                                              // read at the top of the NEXT one (a wait right behind the load cost a memory round trip per trip)

    while (true) {
        [[maybe_unused]] const long long sp_a = SPROF_T();
        [[maybe_unused]] bool draining = false;
        // lane index as the GLOBAL-memory address expressions of this trip see it: opaque per trip, so that `pointer + 8 l` is
        // formed where it is used (end of a solve, pull) instead of being hoisted out of the loop - a dozen loop-invariant 64-bit
        // lane addresses were live across the interior-point iteration, spilled to scratch and re-loaded every trip
        int lg = l;
        asm volatile("" : "+v"(lg));
        if constexpr (!ASYNC) {
            if (p.drain_count != nullptr) {
                draining = __builtin_amdgcn_readfirstlane(drain_raw) >= p.drain_thresh;
                drain_raw = aload(p.drain_count);      // consumed at the top of the next trip
            }
        }
        // ---- 1. end of a solve? ---------------------------------------------------------------
        bool push = false;
        long long t_used = 0;                 // (budgeted solves only) device-clock ticks this solve has spent iterating
        if (timed) t_used = (long long)wall_clock64() - *tbase;
        if (have) {
            int code = -1;
            if (stalled) code = 0;
            else if (r_vio < o.r_tol && k_vio < o.kappa_tol) code = 1;
            else if (iters >= o.max_iter || (timed && t_used >= p.budget_ticks)) code = 0;      // out of iterations / out of time
            else if (done_here >= p.iter_cap || (!ASYNC && draining && done_here >= p.drain_min)) code = 2;
            if (code >= 0) {
                const size_t pi = (size_t)prob;
                const int sb = prob / p.H;
                double* ps = p.pstate + pi * PS;
                if (code == 2) {             // park: exact state, re-queued for the next round
                    if (vx) { ps[lg] = S.x; ps[NX + 2 * NY + lg] = S.rdyn; }
                    if (vy) { ps[NX + lg] = S.y1; ps[NX + NY + lg] = S.y2; ps[2 * NX + 2 * NY + lg] = S.rrst; ps[2 * NX + 3 * NY + lg] = S.rbil; }
                    if (l == 0) {
                        // (iteration count and, for budgeted solves, the time used so far share the last word: iters < 128)
                        ps[PS - 4] = r_vio; ps[PS - 3] = k_vio; ps[PS - 2] = reg; ps[PS - 1] = (double)((long long)iters + (timed ? t_used * 128 : 0));
                        p.pflag[pi] = 1;
                        const int pos = atomicAdd(qcount(p.Q, par ^ 1, knot), 1);
                        p.Q.items[((size_t)(par ^ 1) * K + knot) * cap + pos] = prob;
                        atomicAdd(p.pending_count, 1);
                    }
                } else {
                    if (l == 0) { xst<ASYNC>(p.status + pi, code); xst<ASYNC>(p.iters + pi, iters); }
                    // d = z[1:nd] - [q_{i+2}; gamma_i; b_i]  (implicit_dynamics.jl:180-190)
                    if (vx) xst<ASYNC>(p.d + pi * ND + lg, S.x - qinit);
                    if constexpr (M::MODE == CIMPC_MODE_CONFIGURATIONFORCE) {
                        if (l < NC) xst<ASYNC>(p.d + pi * ND + NX + lg, S.y1 - xld<ASYNC>(p.gam + pi * NC + lg));
                        else if (l < NC + NB) xst<ASYNC>(p.d + pi * ND + NX + lg, S.y1 - xld<ASYNC>(p.bfr + pi * NB + (lg - NC)));
                    }
                    if (p.zout != nullptr) {
                        double* zo = p.zout + pi * M::NZ;
                        if (vx) zo[lg] = S.x;
                        if (vy) { zo[NX + lg] = S.y1; zo[NX + NY + lg] = S.y2; }
                    }
                    if (code == 1) {         // converged: z* parked, sensitivities deferred to an idle moment
                        if (vx) ps[lg] = S.x;
                        if (vy) { ps[NX + lg] = S.y1; ps[NX + NY + lg] = S.y2; }
                        if (l == 0) ps[PS - 2] = reg;
                        push = true;
                    } else {                 // failed: the slot keeps its previous sensitivities
                        {
                            const long long t0 = dbg_on ? wall_clock64() : 0;
                            problem_done<ASYNC>(p, sb, l);
                            if (dbg_on) dbg_tfence += wall_clock64() - t0;
                        }
                    }
                }
                have = false;
            }
        }
        {   // converged this trip -> the wave's list (positions by group order)
            const unsigned long long pm = __ballot((push && l == 0) ? 1 : 0);
            if (pm != 0ull) {
                if (push && l == 0) backlog[nback + __popcll(pm & ((1ull << (tid & 63)) - 1ull))] = prob;
                nback += __popcll(pm);
            }
        }
        [[maybe_unused]] const long long sp_b = SPROF_T();
        // ---- 2. pull the next problem of this knot -----------------------------------------
        // (ASYNC: the queue is live - an idle group looks again every few trips while its wave is busy)
        if (!have && (!exhausted || (ASYNC && (++idle_trips & 7) == 0))) {
            int idx = 0, item = 0;
            if constexpr (ASYNC) {       // live queue: claim only what has been published
                const long long tq0 = dbg_on ? wall_clock64() : 0;
                if (l == 0) item = aq_pop(const_cast<int*>(items), head, tailp, p.A.dbg ? dbg_pop : nullptr, p.A.abort_flag);
                item = group_bcast0<G>(item);
                idx = item < 0 ? n : 0;
                const long long tq1 = dbg_on ? wall_clock64() : 0;
                if (item >= 0) xfence(p.A.flags);      // acquire the candidate trajectory of the producer
                if (dbg_on) { dbg_tpop += tq1 - tq0; dbg_tfence += wall_clock64() - tq1; }
            } else {
                if (l == 0) idx = atomicAdd(head, 1);
                idx = group_bcast0<G>(idx);
            }
            if (idx < n) {
                prob = ASYNC ? item : items[idx];
                const int sb = prob / p.H, i = prob - sb * p.H;
                const size_t pi = (size_t)prob;
                const double* th = p.theta + pi * NTH;
                // every input that depends on the problem id alone is REQUESTED here, in one batch, before the first LDS hand-over:
                // theta (the runtime loop over its two chunks waited for each load in turn), the rollout's altitude, the initial
                // configuration and the parked-solve flag (the fences of the hand-over kept these loads behind it - the pull was
                // five to six dependent round trips instead of three: claim, problem id, inputs)
                constexpr int NTC = (NTH + G - 1) / G;
                double thv[NTC];
                static_for<0, NTC>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const int k = lg + j * G;
                    thv[j] = xld<ASYNC>(th + (k < NTH ? k : 0));
                });
                const double alt_in = (p.alt != nullptr && l < NC) ? p.alt[(size_t)(sb / p.slots) * NC + lg] : 0.0;
                const double* qrow = p.q + ((size_t)sb * (p.H + 2) + (i + 2)) * M::NQ;
                const double q_in = vx ? xld<ASYNC>(qrow + lg) : 0.0;
                const int parked_in = p.pflag[pi];
                static_for<0, NTC>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const int k = l + j * G;
                    if (k < NTH) dth[k] = thv[j] - ctab[L.oTh0 + k];
                });
                wave_lds_fence();
                {   // rthdyn*(th-th0), rthrst*(th-th0): constant per solve.  Even / odd partial sums (the association of rounds 1-5);
                    // 16-lane groups: in chunks of eight columns whose 24 operands are requested together and waited for once - the
                    // rolled loop waited for its own three reads in each of its NTH / 2 trips
                    double a0 = 0.0, a1 = 0.0, c0 = 0.0, c1 = 0.0;
                    if constexpr (G == 16) {
                        constexpr int CHK = 8;
                        static_for<0, (NTH + CHK - 1) / CHK>([&](auto cc) {
                            constexpr int k0 = decltype(cc)::value * CHK, n = NTH - k0 < CHK ? NTH - k0 : CHK;
                            double dv[n], od[n], orr[n];
                            static_for<0, n>([&](auto jc) {
                                constexpr int j = decltype(jc)::value;
                                dv[j] = dth[k0 + j]; od[j] = ctab[L.oRthDyn + (k0 + j) * G + l]; orr[j] = ctab[L.oRthRst + (k0 + j) * G + l];
                            });
                            pin_values(dv); pin_values(od); pin_values(orr);
                            static_for<0, n>([&](auto jc) {
                                constexpr int j = decltype(jc)::value;
                                if constexpr (((k0 + j) & 1) == 0) { a0 = fma(od[j], dv[j], a0); c0 = fma(orr[j], dv[j], c0); }
                                else { a1 = fma(od[j], dv[j], a1); c1 = fma(orr[j], dv[j], c1); }
                            });
                        });
                    } else {
                        int k = 0;
                        for (; k + 1 < NTH; k += 2) {
                            const double d0 = dth[k], d1 = dth[k + 1];
                            a0 = fma(ctab[L.oRthDyn + k * G + l], d0, a0);
                            c0 = fma(ctab[L.oRthRst + k * G + l], d0, c0);
                            a1 = fma(ctab[L.oRthDyn + (k + 1) * G + l], d1, a1);
                            c1 = fma(ctab[L.oRthRst + (k + 1) * G + l], d1, c1);
                        }
                        for (; k < NTH; ++k) {
                            const double d0 = dth[k];
                            a0 = fma(ctab[L.oRthDyn + k * G + l], d0, a0);
                            c0 = fma(ctab[L.oRthRst + k * G + l], d0, c0);
                        }
                    }
                    S.tthdyn = a0 + a1;
                    S.tthrst = c0 + c1;
                }
                S.altl = alt_in;
                qinit = q_in;
                const double* ps = p.pstate + pi * PS;
                if (parked_in == 1) {        // resume a parked solve
                    S.x = vx ? ps[lg] : 0.0;
                    S.y1 = vy ? ps[NX + lg] : 1.0;
                    S.y2 = vy ? ps[NX + NY + lg] : 1.0;
                    S.rdyn = vx ? ps[NX + 2 * NY + lg] : 0.0;
                    S.rrst = vy ? ps[2 * NX + 2 * NY + lg] : 0.0;
                    S.rbil = vy ? ps[2 * NX + 3 * NY + lg] : 0.0;
                    r_vio = ps[PS - 4]; k_vio = ps[PS - 3]; reg = ps[PS - 2];
                    const long long it_word = (long long)ps[PS - 1];
                    // the word packs (ticks used << 7 | iterations) only when a budget is set (then max_iter < 128,
                    // cimpc_create); without one it is the plain count, whatever max_iter is
                    iters = timed ? (int)(it_word & 127) : (int)it_word;
                    if (timed && l == 0) *tbase = (long long)wall_clock64() - (it_word >> 7);      // the time already used counts
                    wave_lds_fence();
                    if (l == 0) p.pflag[pi] = 0;
                } else {
                    // z_initialize!: z .= 1, z[iq2] = q   (simulation.jl:59-63)
                    S.x = qinit; S.y1 = 1.0; S.y2 = 1.0;
                    S.residual(0.0);
                    r_vio = S.r_violation();
                    k_vio = S.k_violation();
                    iters = 0;
                    reg = 0.0;
                    if (timed && l == 0) *tbase = (long long)wall_clock64();
                }
                if (timed) wave_lds_fence();
                done_here = 0;
                stalled = false;
                have = true;
#ifdef CIMPC_SWEEP_PROF
                if (sp) sp[8] += 1;
#endif
            } else {
                exhausted = true;
            }
        }
        // ---- 3. wave-uniform trip: EITHER one interior-point iteration for every group that holds a
        //         problem, OR one deferred sensitivity pass.  The sensitivity trip is taken when
        //         every group of the wave has something in its backlog (all four groups then run
        //         the same code on their own problem - no SIMT divergence), when the wave has no
        //         interior-point work left, or when a backlog is full.
        [[maybe_unused]] const long long sp_c = SPROF_T();
#ifdef CIMPC_SWEEP_PROF
        if (sp) { sp[3] += sp_b - sp_a; sp[4] += sp_c - sp_b; }
#endif
        if (dbg_on) { dbg_trips += 1; dbg_act += have ? 1 : 0; }
        const bool any_ip = __any(have ? 1 : 0);
        if (!any_ip && nback == 0) break;
        // (ASYNC: a finished solve must not wait for its neighbours' work to dry up - the rollout's next
        //  stage hangs on it; a group that has a backlog and nothing else to do gets its trip at once)
        const bool sens_trip = !any_ip || nback >= GPW || (ASYNC && nback > 0 && __any(have ? 0 : 1));
        if (!sens_trip) {
            if (have && !(r_vio < o.r_tol && k_vio < o.kappa_tol) && iters < o.max_iter && done_here < p.iter_cap) {
                ++done_here;
                ++iters;
                stalled = S.iterate(o, p.kc_floor, p.tau_floor, reg, r_vio, k_vio);
            }
        } else {
            // one problem per group while the list holds that many; a shorter list (only when the wave has nothing else to do, or
            // a rollout is waiting) is shared out: 1 problem -> every group takes a share of its columns, 2 of 4 groups per problem
            int take, part = 0, nparts = 1;
            if (nback >= GPW) { take = nback - 1 - gw; nback -= GPW; }
            else {
                if (nback == 1) { take = 0; part = gw; nparts = GPW; }
                else if (GPW == 4 && nback == 2) { take = gw >> 1; part = gw & 1; nparts = 2; }
                else take = gw < nback ? gw : -1;
                nback = 0;
            }
            wave_lds_fence();
            if (take >= 0) {
                // the live iterate (if any) survives in registers; factorization registers are scratch
                const double sx = S.x, sy1 = S.y1, sy2 = S.y2, sd = S.rdyn, sr = S.rrst, sb_ = S.rbil, st = S.tthdyn, su = S.tthrst, sa = S.altl;
                const int pr = backlog[take];
                sensitivities<M, ASYNC>(p, S, tab, pr, l, part, nparts);
                S.x = sx; S.y1 = sy1; S.y2 = sy2; S.rdyn = sd; S.rrst = sr; S.rbil = sb_; S.tthdyn = st; S.tthrst = su; S.altl = sa;
            }
        }
#ifdef CIMPC_SWEEP_PROF
        if (sp) { const long long sp_d = SPROF_T(); if (sens_trip) { sp[6] += sp_d - sp_c; sp[10] += 1; } else { sp[5] += sp_d - sp_c; sp[9] += 1; } }
#endif
    }
    if constexpr (ASYNC) {      // diagnostics: group-trips with / without a problem
        if (p.A.dbg != nullptr && l == 0) {
            atomicAdd((unsigned long long*)p.A.dbg + 4, (unsigned long long)dbg_act);
            atomicAdd((unsigned long long*)p.A.dbg + 5, (unsigned long long)dbg_trips);
            atomicAdd((unsigned long long*)p.A.dbg + 6, (unsigned long long)dbg_tpop);
            atomicAdd((unsigned long long*)p.A.dbg + 7, (unsigned long long)dbg_tfence);
            for (int k = 0; k < 4; ++k) atomicAdd((unsigned long long*)p.A.dbg + 12 + k, (unsigned long long)dbg_pop[k]);
        }
    }
}

// ----------------------------------------------------------------------------------------
// Queue kernel of the lock-step rounds: persistent workgroups; each serves one knot at a time and
// hops to another knot that still has work when its queue is empty.
// ----------------------------------------------------------------------------------------
// (32-lane models: the table + group scratch fill most of a CU's LDS, one workgroup per CU is resident anyway -
//  the kernel may then use the full 512-VGPR budget instead of spilling: centroidal 341 spilled VGPRs -> 0)
// Occupancy of the 16-lane sweep: CIMPC_SWEEP_THREADS (largest workgroup launched) x CIMPC_SWEEP_OCC workgroups per CU.
// Default 256 x 2 = 8 waves per CU (2 per SIMD, 256 VGPRs).
#define CIMPC_SWEEP_OCC 2      // (a constant of the build: the -D override is gone with the experiment it served)
#define CIMPC_SWEEP_THREADS 256      // (a constant of the build: the -D override is gone with the experiment it served)
// 32-lane models come in two builds of this kernel (round 4).  Model::WIDE = false: 256 threads, one wave per SIMD with the whole register
// file (no scratch) - the latency form, right while a launch holds about one problem per lane group (configs[4] at 64 rollouts:
// 7.9 ms of sweeps per step, 11.6 ms with the other build).  Model::WIDE = true: 512 threads = two waves per SIMD at 256 registers (the R rows
// spill to scratch between the factorization and the back-substitutions, 324 bytes per lane) next to the ONE table a CU has room for -
// the throughput form (Model::Wide: its own LDS layout, a windowed R transposition), 13-14 % faster from 128 rollouts on
// (profiles/r04/cent_w8b.log, cent_two_builds.log).  The host picks by problems per sweep.
template <class M>
__global__ __launch_bounds__(M::G == 16 ? CIMPC_SWEEP_THREADS : (M::WIDE ? 512 : 256), M::G == 16 ? CIMPC_SWEEP_OCC : 1) void ip_queue_kernel(IpParams p) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int s_knot, s_total, s_rem[PICK_MAXK];
    const int tid = (int)threadIdx.x;
    __builtin_amdgcn_s_setprio(CIMPC_SWEEP_PRIO);
    [[maybe_unused]] long long sp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    [[maybe_unused]] const long long sp_life = SPROF_T();
    if (p.direct) {      // single rollouts: workgroup k = knot k, no remaining-work scan before or after (IpParams::direct)
        const int knot = (int)blockIdx.x;
        if (knot < p.Q.K && aload(qcount(p.Q, p.Q.par, knot)) > 0) serve_knot<M, false>(p, smem, knot, tid, sp);
        return;
    }
    while (true) {
        [[maybe_unused]] const long long sp_t = SPROF_T();
        __syncthreads();            // every wave is done with the staged table
        const int knot = pick_knot(p, s_rem, &s_total, &s_knot, tid, (int)blockIdx.x, (int)gridDim.x);
#ifdef CIMPC_SWEEP_PROF
        sp[1] += SPROF_T() - sp_t;
#endif
        if (knot < 0) {
            if (tid == 0 && p.drain_count != nullptr) atomicAdd(p.drain_count, 1);
            break;
        }
        serve_knot<M, false>(p, smem, knot, tid, sp);
    }
#ifdef CIMPC_SWEEP_PROF
    if ((tid & 63) == 0) {
        sp[0] = SPROF_T() - sp_life; sp[7] = 1;
        for (int k = 0; k < 11; ++k) atomicAdd(&g_sweep_prof[k], (unsigned long long)sp[k]);
    }
#endif
}

// ----------------------------------------------------------------------------------------
// B2 seam: the reference-side callbacks RoboDojo's interior point calls on a knot's linearized problem, exposed one by
// one (cimpc_ip_residual / cimpc_ip_linear_solve).  The SAME IpSolver members the sweep runs - rlin!
// (linearized_solver.jl:364-373), rzlin! + schur_factorize! (:378-399, schur.jl:80-88), linear_solve!(D, rz, r)
// (:424-444) - on caller-supplied points, so that the reference's own known-answer tests of these operators
// (test/controller/linearized_solver.jl:55-57, test/solver/schur.jl:19-62) run against the device arithmetic.
// One wavefront per workgroup, 64 / G problems per wavefront.
// ----------------------------------------------------------------------------------------
template <class M>
__global__ __launch_bounds__(64) void ip_callback_kernel(IpCallbackArgs a) {
    constexpr int NX = M::NX, NY = M::NY, NTH = M::NTH, G = M::G;
    constexpr LinLayout L(NX, NY, NTH, G, M::NTHS, M::ADJ);
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = (int)threadIdx.x, grp = tid / G, l = tid % G;
    double* tab = smem;
    double* Rst = smem + L.size + (size_t)grp * M::LDS_GROUP;
    stage_table<M>(tab, a.tab, a.knot, tid);
    __syncthreads();
    const int prob = (int)blockIdx.x * (64 / G) + grp;
    const bool live = prob < a.n;
    const size_t pi = (size_t)(live ? prob : 0);
    IpSolver<M> S;
    S.bind(tab, Rst, l);
    const bool vx = S.vx, vy = S.vy;
    const double* z = a.z + pi * M::NZ;
    S.x = vx ? z[l] : 0.0;
    S.y1 = vy ? z[NX + l] : 1.0;
    S.y2 = vy ? z[NX + NY + l] : 1.0;
    double* out = a.out + pi * M::NZ;
    if (a.op == 0) {         // rlin!(r, z, theta, kappa)
        const double* th = a.theta + pi * NTH;
        double td[2] = {0.0, 0.0}, tr[2] = {0.0, 0.0};      // (even / odd partial sums: the association serve_knot uses)
        for (int k = 0; k < NTH; ++k) {
            const double dk = th[k] - tab[L.oTh0 + k];
            td[k & 1] = fma(tab[L.oRthDyn + k * G + l], dk, td[k & 1]);
            tr[k & 1] = fma(tab[L.oRthRst + k * G + l], dk, tr[k & 1]);
        }
        S.tthdyn = td[0] + td[1]; S.tthrst = tr[0] + tr[1];
        S.altl = (a.alt != nullptr && l < M::NC) ? a.alt[pi * M::NC + l] : 0.0;
        S.residual(a.kappa);
        if (live && vx) out[l] = S.rdyn;
        if (live && vy) { out[NX + l] = S.rrst; out[NX + NY + l] = S.rbil; }
    } else {                 // rzlin!(rz, z; reg) ; linear_solve!(Delta, rz, r; reg)
        const double* r = a.r + pi * M::NZ;
        S.rdyn = vx ? r[l] : 0.0;
        S.rrst = vy ? r[NX + l] : 0.0;
        S.rbil = vy ? r[NX + NY + l] : 0.0;
        S.factorize(a.reg);
        S.linear_solve();
        if (live && vx) out[l] = S.Dx_;
        if (live && vy) { out[NX + l] = S.Dy1_; out[NX + NY + l] = S.Dy2_; }
    }
}

template <class M>
int launch_callback(const IpCallbackArgs& a, hipStream_t s) {
    constexpr LinLayout L(M::NX, M::NY, M::NTH, M::G, M::NTHS, M::ADJ);
    const int ppw = 64 / M::G;
    const size_t lds = (size_t)(L.size + ppw * M::LDS_GROUP) * sizeof(double);
    static LdsOptIn optin;
    if (lds_opt_in(optin, (const void*)ip_callback_kernel<M>, lds) != CIMPC_OK) return CIMPC_ERR_HIP;
    hipLaunchKernelGGL((ip_callback_kernel<M>), dim3((a.n + ppw - 1) / ppw), dim3(64), lds, s, a);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}

// ----------------------------------------------------------------------------------------
// per-model launch / info (instantiated in ip_model_*.hip, one translation unit per model
// so that the models build in parallel)
// ----------------------------------------------------------------------------------------
template <class M>
int launch_model(const IpParams& p, int waves, hipStream_t s) {
    constexpr LinLayout L(M::NX, M::NY, M::NTH, M::G, M::NTHS, M::ADJ);
    if (waves < 1 || waves > (M::G == 16 ? CIMPC_SWEEP_THREADS : 512) / 64 || waves == 3) return CIMPC_ERR_INVALID;
    const int ppw = 64 / M::G;
    const size_t lds = (size_t)(L.hot + waves * ppw * M::LDS_GROUP) * sizeof(double);
    const int grid = p.wpk;      // persistent workgroups of the launch
    if constexpr (M::G == 64) { if (waves > 4) return CIMPC_ERR_INVALID; }      // (one build: four waves of one problem each)
    if constexpr (M::G == 32) {
        if (waves > 4) {         // throughput build: two waves per SIMD, its own per-problem LDS layout
            using MW = typename M::Wide;
            constexpr LinLayout LW(MW::NX, MW::NY, MW::NTH, MW::G, MW::NTHS, MW::ADJ);
            const size_t lds_w = (size_t)(LW.size + waves * ppw * MW::LDS_GROUP) * sizeof(double);
            static LdsOptIn optin_w;
            if (lds_opt_in(optin_w, (const void*)ip_queue_kernel<MW>, lds_w) != CIMPC_OK) return CIMPC_ERR_HIP;
            hipLaunchKernelGGL((ip_queue_kernel<MW>), dim3(grid), dim3(64 * waves), lds_w, s, p);
            return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
        }
    }
    static LdsOptIn optin;
    if (lds_opt_in(optin, (const void*)ip_queue_kernel<M>, lds) != CIMPC_OK) return CIMPC_ERR_HIP;
    hipLaunchKernelGGL((ip_queue_kernel<M>), dim3(grid), dim3(64 * waves), lds, s, p);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}

template <class M>
void info_model(KernelInfo* info) {
    constexpr LinLayout L(M::NX, M::NY, M::NTH, M::G, M::NTHS, M::ADJ);
    info->G = M::G;
    info->lds_table = L.hot;
    info->lds_group = M::LDS_GROUP;
    info->tab_size = L.size;
    info->dtn_ld = M::DTN_LD;
    info->adj = M::ADJ;
}

#define CIMPC_DEFINE_MODEL(name, q, u, w, c, b)                                              \
    int ip_launch_##name(int mode, const IpParams& p, int waves, hipStream_t s) {            \
        if (mode == 0) return launch_model<Model<q, u, w, c, b, 0>>(p, waves, s);            \
        return launch_model<Model<q, u, w, c, b, 1>>(p, waves, s);                           \
    }                                                                                        \
    int ip_callback_##name(int mode, const IpCallbackArgs& a, hipStream_t s) {               \
        if (mode == 0) return launch_callback<Model<q, u, w, c, b, 0>>(a, s);                \
        return launch_callback<Model<q, u, w, c, b, 1>>(a, s);                               \
    }                                                                                        \
    void ip_info_##name(int mode, KernelInfo* info) {                                        \
        if (mode == 0) info_model<Model<q, u, w, c, b, 0>>(info);                            \
        else info_model<Model<q, u, w, c, b, 1>>(info);                                      \
    }

// 64-lane models (ny > 32: one problem per wavefront): :configuration mode only - :configurationforce and the B2 callbacks stay
// with the runtime-dimension kernel / are not offered (ip_dispatch.hip)
#define CIMPC_DEFINE_MODEL64(name, q, u, w, c, b)                                            \
    int ip_launch_##name(int mode, const IpParams& p, int waves, hipStream_t s) {            \
        if (mode != 0) return CIMPC_ERR_INVALID;                                             \
        return launch_model<Model<q, u, w, c, b, 0>>(p, waves, s);                           \
    }                                                                                        \
    void ip_info_##name(int mode, KernelInfo* info) { (void)mode; info_model<Model<q, u, w, c, b, 0>>(info); }

}  // namespace cimpc

#ifdef CIMPC_UBENCH
// diagnostic builds only (-DCIMPC_UBENCH, scripts/dbg/ubench_ip.py): shader-clock cost of the interior-point primitives on `waves`
// wavefronts of ONE workgroup (4 problems per wave, the same problem in every group) - what a lone wave pays per call, i.e. the
// latency chain the small batches, the launch tails and the asynchronous solve run on.
namespace cimpc {
template <class M>
__global__ __launch_bounds__(M::G == 16 ? 256 : 512, M::G == 16 ? 2 : 1) void ip_ubench_kernel(const double* tabs, cimpc_ip_opts o, int reps, long long* out) {
    constexpr LinLayout L(M::NX, M::NY, M::NTH, M::G, M::NTHS, M::ADJ);
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = (int)threadIdx.x, grp = tid / M::G, l = tid % M::G;
    stage_table<M>(smem, tabs, 0, tid);
    __syncthreads();
    IpSolver<M> S;
    S.bind(smem, smem + L.hot + (size_t)grp * M::LDS_GROUP, l);
    S.x = S.vx ? S.x0 : 0.0; S.y1 = 1.0; S.y2 = 1.0;
    long long t[8];
    double sink = 0.0;
    auto opaque = [&] { asm volatile("" : "+v"(S.x), "+v"(S.y1), "+v"(S.y2)); };
    t[0] = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) { S.residual(1e-3); sink += S.rdyn + S.rrst; opaque(); }
    t[1] = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) { sink += S.r_violation() + S.k_violation(); S.rdyn += 1e-30; opaque(); }
    t[2] = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) { S.factorize(1e-9); sink += S.rdinv; S.y1 += 1e-30 * S.rdinv; opaque(); }
    t[3] = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) { S.linear_solve(); sink += S.Dx_ + S.Dy1_; S.rdyn += 1e-30 * S.Dx_; opaque(); }
    t[4] = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int r = 0; r < reps; ++r) { sink += S.step_length(0.99); S.Dy1_ += 1e-30; opaque(); }
    t[5] = (long long)__builtin_amdgcn_s_memtime();
    double reg = 0.0, r_vio = 1.0, k_vio = 1.0;
    S.x = S.vx ? S.x0 : 0.0; S.y1 = 1.0; S.y2 = 1.0;
    S.residual(0.0); r_vio = S.r_violation(); k_vio = S.k_violation();
    int its = 0;
#pragma unroll 1
    for (int r = 0; r < reps; ++r) {
        if (r_vio < o.r_tol && k_vio < o.kappa_tol) { S.x = S.vx ? S.x0 : 0.0; S.y1 = 1.0; S.y2 = 1.0; S.residual(0.0); r_vio = S.r_violation(); k_vio = S.k_violation(); }
        (void)S.iterate(o, o.kappa_tol / o.undercut, 1.0 - o.eps_min, reg, r_vio, k_vio);
        ++its;
    }
    t[6] = (long long)__builtin_amdgcn_s_memtime();
    if ((tid & 63) == 0) {
        long long* w = out + 8 * (tid >> 6);
        for (int k = 0; k < 6; ++k) w[k] = t[k + 1] - t[k];
        w[6] = its; w[7] = (long long)(sink * 1e-300);
    }
}
}  // namespace cimpc
#endif
