// Device bodies of the horizon-level Newton stages (shared by the lock-step kernels of
// newton_kernels.hip and by the single-launch asynchronous solve of newton_async_impl.h):
//   resid_decide_body    residual! + gradient!     /root/reference/src/controller/newton_residual.jl:113-138, 178-219, 256-281
//                        line search / accept / beta schedule   newton.jl:223-280
//                        update_traj! + update_theta!            newton_residual.jl:140-176
//   kkt_body             jacobian! + linear_solve!(solver, D, R, r)  newton_jacobian.jl:148-198,
//                        newton.jl:210-218, solved through the condensing of
//                        newton_structure_solver/methods.jl:386-557 (dual Schur complement
//                        Y = C P^-1 C^T + rho I, block-pentadiagonal, block Cholesky).
// `Sync` is the barrier of the executing unit: a whole workgroup (BlockSync) or one wavefront of a
// larger workgroup (WaveSync).
#pragma once
#include "lane_group.h"
#include "newton_state.h"

namespace cimpc {

struct BlockSync { static __device__ __forceinline__ void sync() { __syncthreads(); } };
struct WaveSync {     // one wavefront: memory operations of the wave are complete and visible to its lanes
    static __device__ __forceinline__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
};

__device__ __forceinline__ void theta_update(const NewtonDev& S, const TrajDev& T, int b, int i,
                                             int tid, int nthreads) {
    // update_theta!(traj, i) (trajectory.jl:67-82): th_i = [q_i; q_{i+1}; u_i; w_i; mu; h]
    const cimpc_dims& m = S.dm;
    double* th = T.th + ((size_t)b * m.H + i) * S.nth;
    const double* q = T.q + ((size_t)b * (m.H + 2) + i) * m.nq;
    const double* u = T.u + ((size_t)b * m.H + i) * m.nu;
    const double* w = T.w + ((size_t)b * m.H + i) * m.nw;
    for (int k = tid; k < 2 * m.nq; k += nthreads) th[k] = q[k];   // q_i, q_{i+1} contiguous
    for (int k = tid; k < m.nu; k += nthreads) th[2 * m.nq + k] = u[k];
    for (int k = tid; k < m.nw; k += nthreads) th[2 * m.nq + m.nu + k] = w[k];
}

__device__ __forceinline__ void copy_n(double* dst, const double* src, int n, int tid, int nt) {
    for (int k = tid; k < n; k += nt) dst[k] = src[k];
}

// Candidate slots.  The reference's backtracking line search (newton.jl:223-269) evaluates
// alpha = 1, 1/2, ..., 1/64 one after the other, each evaluation being a full implicit_dynamics!
// sweep.  Here a rollout owns CS = 4 evaluation slots and the SAME sequence is evaluated
// speculatively in at most three lock-step rounds:
//     round A: alpha = 1                      (slot 0)
//     round B: alpha = 1/2, 1/4               (slots 0-1)
//     round C: alpha = 1/8, 1/16, 1/32, 1/64  (slots 0-3)
// or, for a rollout whose PREVIOUS line search needed round C (typically one sitting at its
// residual floor, which exhausts the search in every iteration), all seven step lengths in one
// round (STAGE_LS7, slots 0-6):
// and the first alpha (in the reference's order) that passes the Armijo-type test is taken, so the
// accepted step, residual and implicit-dynamics data are exactly those the sequential loop would
// have produced.  Every "evaluation" array (candidate trajectory, nu_cand, d, dz, status,
// res_cand, ...) is indexed by slot sb = b*CS + it where `it` is the line-search iterate (alpha = 2^-it): a
// step length always lives in the same slot, so the whole search history of the running Newton iteration
// stays available (needed by the stale-sensitivity rule, dz_eff below).
__device__ __forceinline__ double ls_alpha(int iter) { return ldexp(1.0, -iter); }

// slot sb will have been evaluated when the round that consumes queue `par` reaches its residual launch
__device__ __forceinline__ void list_slot(const NewtonDev& S, size_t sb, int par, int pos = -1) {
    if (S.slot_list == nullptr) return;
    if (pos < 0) pos = atomicAdd(&S.counters[4 * CPAD], 1);
    if (pos >= S.dm.B * CS) __builtin_trap();      // (a slot is listed at most once per round: cannot happen)
    S.slot_list[(size_t)par * S.dm.B * CS + pos] = (int)sb;
}

// Request the evaluation (implicit_dynamics! sweep) of slot sb: one queue entry per horizon step,
// bucketed by the reference knot of that step.
// list_pos: position of the slot in the round's slot list (reset / B3 seam: slot 0 of rollout b sits at b), -1 = append
__device__ __forceinline__ void enqueue_eval(const NewtonDev& S, size_t sb, int b, int par, int tid, int nt, int list_pos = -1) {
    const int H = S.dm.H, K = S.WQ.K;
    for (int k = tid; k < H; k += nt) {
        const int t = S.WQ.window[(size_t)b * (H + 2) + k];
        const int pos = atomicAdd(qcount(S.WQ, par, t), 1);
        S.WQ.items[((size_t)par * K + t) * S.WQ.cap + pos] = (int)(sb * H + k);
    }
    if (tid == 0) {
        S.WQ.done_count[sb] = 0;
        S.need_sweep[sb] = 1;
        list_slot(S, sb, par, list_pos);
    }
}

// The same for the consecutive slots sb .. sb+n-1 of one rollout (the candidates of a line-search batch): they share the window,
// so one atomic per horizon step reserves the entries of all of them, and one reserves their places in the round's slot list
// (a round of 512 rollouts x 3 candidates x 40 steps put 61 k atomics on the 60 queue counters and 1.5 k on the list counter:
// same-address atomics serialise, the last decision blocks of a round queued behind them).
// mark: value written to need_sweep - 1 = "evaluated by the sweep this round's decision stage follows", 2 = "requested by the KKT
// kernel that runs NEXT TO the running round for the round after it" (see the decision stage's entry checks)
__device__ __forceinline__ void enqueue_evals(const NewtonDev& S, size_t sb, int n, int b, int par, int tid, int nt, int mark = 1) {
    const int H = S.dm.H, K = S.WQ.K;
    for (int k = tid; k < H; k += nt) {
        const int t = S.WQ.window[(size_t)b * (H + 2) + k];
        const int pos = atomicAdd(qcount(S.WQ, par, t), n);
        int* it = S.WQ.items + ((size_t)par * K + t) * S.WQ.cap + pos;
        for (int c = 0; c < n; ++c) it[c] = (int)((sb + c) * H + k);
    }
    if (tid == 0) {
        int pos = 0;
        if (S.slot_list != nullptr) pos = atomicAdd(&S.counters[4 * CPAD], n);
        atomicAdd(&S.counters[5 * CPAD], n);      // NEW evaluation requests of the round (a deterministic count: slots re-listed while a solve is parked are not in it)
        if (pos + n > S.dm.B * CS) __builtin_trap();      // (a slot is listed at most once per round: cannot happen)
        for (int c = 0; c < n; ++c) {
            S.WQ.done_count[sb + c] = 0;
            S.need_sweep[sb + c] = mark;
            if (S.slot_list != nullptr) S.slot_list[(size_t)par * S.dm.B * CS + pos + c] = (int)(sb + c);
        }
    }
}

// Asynchronous solve: request the evaluation of slots sb0 .. sb0+nslots-1 of rollout b.  The slots'
// candidate trajectories must have been written by the calling unit; they are released before the
// first queue entry is published (consumers acquire after claiming an entry).
template <class Sync>
__device__ __forceinline__ void enqueue_eval_async(const NewtonDev& S, size_t sb0, int first, int nslots, int b, int tid, int nt) {
    const int H = S.dm.H;
    if (tid == 0) {
        for (int c = 0; c < CS; ++c) {
            const bool on = c >= first && c < first + nslots;
            if (on) S.WQ.done_count[sb0 + c] = 0;
            S.need_sweep[sb0 + c] = on;
        }
        astore(S.A.evals_left + b, nslots);
    }
    xfence(S.A.flags);
    Sync::sync();
    for (int e = tid; e < nslots * H; e += nt) {
        const int c = first + e / H, k = e % H;
        const int t = S.WQ.window[(size_t)b * (H + 2) + k];
        aq_push(S.WQ.items + (size_t)t * S.WQ.cap, qcount(S.WQ, 0, t), (int)((sb0 + c) * H + k));
    }
    Sync::sync();
    if (tid == 0) wake_ip(S.A, b);      // wake (a sixteenth of) the idle workgroups
}

// x_dst = traj - alpha*Delta for q_{t+2}, u_t, nu_t (+ gamma, b in cf mode), then update_theta!.
// dst arrays are indexed by `di` (a rollout index for traj, a slot index for candidates).
template <class Sync>
__device__ __forceinline__ void apply_step(const NewtonDev& S, const TrajDev& dst, double* nu_dst_base, size_t di, int b,
                           double alpha, int tid, int nt) {
    const cimpc_dims& m = S.dm;
    const int H = m.H, nq = m.nq, nu = m.nu, nr = S.nr, nd = S.nd;
    const double* D = S.delta + (size_t)b * S.N;
    const bool cf = m.mode == CIMPC_MODE_CONFIGURATIONFORCE;
    const int oq_in_block = cf ? nu + m.nc + m.nb : nu;
    for (int k = tid; k < H * nq; k += nt) {
        const int t = k / nq, c = k - t * nq;
        dst.q[(di * (H + 2) + t + 2) * nq + c] =
            S.traj.q[((size_t)b * (H + 2) + t + 2) * nq + c] - alpha * D[t * nr + oq_in_block + c];
    }
    for (int k = tid; k < 2 * nq; k += nt)       // q_1, q_2 are fixed by (q0, q1)
        dst.q[di * (H + 2) * nq + k] = S.traj.q[(size_t)b * (H + 2) * nq + k];
    for (int k = tid; k < H * nu; k += nt) {
        const int t = k / nu, c = k - t * nu;
        dst.u[(di * H + t) * nu + c] = S.traj.u[((size_t)b * H + t) * nu + c] - alpha * D[t * nr + c];
    }
    if (cf) {
        for (int k = tid; k < H * m.nc; k += nt) {
            const int t = k / m.nc, c = k - t * m.nc;
            dst.g[(di * H + t) * m.nc + c] = S.traj.g[((size_t)b * H + t) * m.nc + c] - alpha * D[t * nr + nu + c];
        }
        for (int k = tid; k < H * m.nb; k += nt) {
            const int t = k / m.nb, c = k - t * m.nb;
            dst.b[(di * H + t) * m.nb + c] = S.traj.b[((size_t)b * H + t) * m.nb + c] - alpha * D[t * nr + nu + m.nc + c];
        }
    }
    for (int k = tid; k < H * nd; k += nt)
        nu_dst_base[di * H * nd + k] = S.nu[(size_t)b * H * nd + k] - alpha * D[H * nr + k];
    Sync::sync();
    {   // update_theta!(dst): th_t = [q_t; q_{t+1}; u_t; w_t; mu; h]; mu, h come from traj
        const int nth = S.nth, nw = m.nw;
        for (int k = tid; k < H * nth; k += nt) {
            const int t = k / nth, c = k - t * nth;
            double v;
            if (c < 2 * nq) v = dst.q[(di * (H + 2) + t) * nq + c];
            else if (c < 2 * nq + nu) v = dst.u[(di * H + t) * nu + (c - 2 * nq)];
            else if (c < 2 * nq + nu + nw) v = S.traj.w[((size_t)b * H + t) * nw + (c - 2 * nq - nu)];
            else v = S.traj.th[((size_t)b * H + t) * nth + c];
            dst.th[(di * H + t) * nth + c] = v;
        }
    }
    Sync::sync();
}

// Accepting an evaluated candidate (newton.jl:273, traj <- traj - alpha*Delta): the slot already holds exactly that trajectory -
// apply_steps formed it with the same multiply-adds - so the accept is a copy of the slot's arrays (q, u, theta, nu, and gamma, b
// in cf mode), every load of a thread issued before its first store, instead of a second pass over traj and Delta with the theta
// update behind a barrier (7 us in the blocks that set the length of a round's decision launch).
template <int UN>
__device__ __forceinline__ void copy_rows(double* __restrict__ dst, const double* __restrict__ src, int n, int tid, int nt) {
    for (int e0 = tid; e0 < n; e0 += UN * nt) {
        double v[UN];
        static_for<0, UN>([&](auto jc) { constexpr int j = decltype(jc)::value; const int e = e0 + j * nt; v[j] = src[e < n ? e : 0]; });
        static_for<0, UN>([&](auto jc) { constexpr int j = decltype(jc)::value; const int e = e0 + j * nt; if (e < n) dst[e] = v[j]; });
    }
}
template <class Sync>
__device__ __forceinline__ void accept_candidate(const NewtonDev& S, size_t sb, int b, int tid, int nt) {
    const cimpc_dims& m = S.dm;
    const int H = m.H, nq = m.nq, nu = m.nu, nd = S.nd, nth = S.nth;
    copy_rows<2>(S.traj.q + (size_t)b * (H + 2) * nq, S.cand.q + sb * (H + 2) * nq, (H + 2) * nq, tid, nt);
    copy_rows<2>(S.traj.u + (size_t)b * H * nu, S.cand.u + sb * H * nu, H * nu, tid, nt);
    copy_rows<5>(S.traj.th + (size_t)b * H * nth, S.cand.th + sb * H * nth, H * nth, tid, nt);
    copy_rows<2>(S.nu + (size_t)b * H * nd, S.nu_cand + sb * H * nd, H * nd, tid, nt);
    if (m.mode == CIMPC_MODE_CONFIGURATIONFORCE) {
        copy_rows<2>(S.traj.g + (size_t)b * H * m.nc, S.cand.g + sb * H * m.nc, H * m.nc, tid, nt);
        copy_rows<2>(S.traj.b + (size_t)b * H * m.nb, S.cand.b + sb * H * m.nb, H * m.nb, tid, nt);
    }
    Sync::sync();
}

// Candidates x = traj - alpha*Delta for q_{t+2}, u_t, nu_t (+ gamma, b in cf mode) and update_theta! (th_t = [q_t; q_{t+1}; u_t; w_t;
// mu; h], mu and h from traj) for `count` line-search candidates at once (evaluation slots sb_first .., step lengths
// 2^-(it_first + c)): one pass over the accepted trajectory and the step writes every candidate, and theta is formed directly from
// the sources (no second pass over the freshly written arrays, no barrier in between).  Every entry is fma(-alpha, Delta, x), which
// is what `x - alpha * Delta` contracts to.  (Measured in the decision kernel: the four candidates of a
// deep line search cost 4 x 4.5 us one after the other and set the length of the launch.)
template <class Sync>
__device__ __forceinline__ void apply_steps(const NewtonDev& S, size_t sb_first, int it_first, int count, int b, int tid, int nt) {
    const cimpc_dims& m = S.dm;
    const int H = m.H, nq = m.nq, nu = m.nu, nw = m.nw, nr = S.nr, nd = S.nd, nth = S.nth;
    const bool cf = m.mode == CIMPC_MODE_CONFIGURATIONFORCE;
    const int oq = cf ? nu + m.nc + m.nb : nu;
    const double* __restrict__ D = S.delta + (size_t)b * S.N;
    const double* __restrict__ tq = S.traj.q + (size_t)b * (H + 2) * nq;
    const double* __restrict__ tu = S.traj.u + (size_t)b * H * nu;
    const double* __restrict__ tw = S.traj.w + (size_t)b * H * nw;
    const double* __restrict__ tth = S.traj.th + (size_t)b * H * nth;
    const double* __restrict__ tnu = S.nu + (size_t)b * H * nd;
    double* __restrict__ cq = S.cand.q + sb_first * (H + 2) * nq;
    double* __restrict__ cu = S.cand.u + sb_first * H * nu;
    double* __restrict__ cth = S.cand.th + sb_first * H * nth;
    double* __restrict__ cnu = S.nu_cand + sb_first * H * nd;
    auto put = [&](double* dst, size_t stride, int k, double base, double dl, bool moves) {
        for (int c = 0; c < count; ++c) dst[(size_t)c * stride + k] = moves ? fma(-ls_alpha(it_first + c), dl, base) : base;
    };
    // UN entries per trip (sized so that 192-256 threads cover a quadruped H = 40 array in ONE trip): the sources of all of them are
    // requested before the first store - one array after the other: with all four arrays gathered in one pass the KKT chains this
    // function is inlined into lost 6 us to register pressure (scripts/dbg/twisted_prof.py) - (round 6: one entry per trip made the
    // 7-candidate start of a deep line search a chain of ~12 dependent load round trips - 22 us in the ten workgroups that set the
    // length of every decision launch, scripts/resid_prof.py)
    // (round 6, end: theta is no longer gathered entry by entry - the choice between four source arrays per entry was compiled as a
    //  pointer table in scratch memory with a full wait behind every lookup, eight dependent round trips per trip, 15-23 us for the four
    //  candidates of a deep search in the ~14 workgroups that set the length of every decision launch (scripts/resid_prof.py).  Every
    //  q / u entry now writes its own candidate array AND the one or two places of theta that hold it; w and the constant tail of
    //  theta are plain copies: five independent gathers, no select on a pointer, no load under a lane condition.)
    auto rows = [&](auto unc, int n, auto&& src, auto&& sink) {
        constexpr int UN = decltype(unc)::value;
        for (int e0 = tid; e0 < n; e0 += UN * nt) {
            double base[UN], dl[UN]; bool mv[UN];
            static_for<0, UN>([&](auto jc) { constexpr int j = decltype(jc)::value; const int e = e0 + j * nt; src(e < n ? e : 0, base[j], dl[j], mv[j]); });
            static_for<0, UN>([&](auto jc) { constexpr int j = decltype(jc)::value; const int e = e0 + j * nt; if (e < n) sink(e, base[j], dl[j], mv[j]); });
        }
    };
    const size_t sth = (size_t)H * nth;
    rows(std::integral_constant<int, 3>{}, (H + 2) * nq, [&](int k, double& base, double& dl, bool& mv) {          // q_1, q_2 are fixed by (q0, q1); q_{t+2} moves
        const int j = k / nq, c = k - j * nq;
        base = tq[k]; mv = j >= 2; dl = D[mv ? (j - 2) * nr + oq + c : 0];      // (a fixed entry reads D[0] and ignores it: no load under a lane condition)
    }, [&](int k, double base, double dl, bool mv) {                                  // update_theta!: th_t = [q_t; q_{t+1}; u_t; w_t; mu; h]
        const int j = k / nq, c = k - j * nq;
        put(cq, (size_t)(H + 2) * nq, k, base, dl, mv);
        if (j < H) put(cth, sth, j * nth + c, base, dl, mv);                      // q_t of step j
        if (j >= 1 && j <= H) put(cth, sth, (j - 1) * nth + nq + c, base, dl, mv);      // q_{t+1} of step j - 1
    });
    rows(std::integral_constant<int, 2>{}, H * nu, [&](int k, double& base, double& dl, bool& mv) {
        const int t = k / nu, c = k - t * nu;
        base = tu[k]; dl = D[t * nr + c]; mv = true;
    }, [&](int k, double base, double dl, bool mv) {
        const int t = k / nu, c = k - t * nu;
        put(cu, (size_t)H * nu, k, base, dl, mv);
        put(cth, sth, t * nth + 2 * nq + c, base, dl, mv);
    });
    if (cf) {
        const double* __restrict__ tg = S.traj.g + (size_t)b * H * m.nc;
        const double* __restrict__ tb = S.traj.b + (size_t)b * H * m.nb;
        for (int k = tid; k < H * m.nc; k += nt) { const int t = k / m.nc, c = k - t * m.nc; put(S.cand.g + sb_first * H * m.nc, (size_t)H * m.nc, k, tg[k], D[t * nr + nu + c], true); }
        for (int k = tid; k < H * m.nb; k += nt) { const int t = k / m.nb, c = k - t * m.nb; put(S.cand.b + sb_first * H * m.nb, (size_t)H * m.nb, k, tb[k], D[t * nr + nu + m.nc + c], true); }
    }
    rows(std::integral_constant<int, 3>{}, H * nd, [&](int k, double& base, double& dl, bool& mv) { base = tnu[k]; dl = D[H * nr + k]; mv = true; },
         [&](int k, double base, double dl, bool mv) { put(cnu, (size_t)H * nd, k, base, dl, mv); });
    const int o_w = 2 * nq + nu, o_c = o_w + nw, n_c = nth - o_c;      // theta's w_t, then its constant tail (mu, h)
    rows(std::integral_constant<int, 1>{}, H * nw, [&](int k, double& base, double& dl, bool& mv) { base = tw[k]; dl = 0.0; mv = false; },
         [&](int k, double base, double dl, bool mv) { const int t = k / nw; put(cth, sth, t * nth + o_w + (k - t * nw), base, dl, mv); });
    if (n_c > 0)
        rows(std::integral_constant<int, 1>{}, H * n_c, [&](int k, double& base, double& dl, bool& mv) { const int t = k / n_c; base = tth[t * nth + o_c + (k - t * n_c)]; dl = 0.0; mv = false; },
             [&](int k, double base, double dl, bool mv) { const int t = k / n_c; put(cth, sth, t * nth + o_c + (k - t * n_c), base, dl, mv); });
    Sync::sync();
}

// Line-search start after the KKT solve (newton.jl:223-228): candidate(s) = traj - alpha*Delta.  One
// candidate (alpha = 1), or all seven when the rollout's previous search went deep (see above).
// mode 1: lock-step (queue `par`), mode 2: asynchronous solve (live queues).
template <class Sync>
__device__ __forceinline__ void start_line_search(const NewtonDev& S, int b, int mode, int lane, int nt) {
    const size_t sb0 = (size_t)b * CS;
    // How many step lengths the first round evaluates: the reference accepts the FIRST alpha = 2^-it that passes, so a
    // batch always starts at it = 0; a rollout whose previous search went to it >= spec_mid / spec_all starts with three /
    // all seven candidates (one round instead of two / three), the first iteration of a solve with spec_first.
    int n = 1;
    {
        const int prev = S.ls_iter[b];
        if (S.newton_l[b] == 0) n = S.spec_first;
        else if (prev >= S.spec_all) n = 7;
        else if (prev >= S.spec_mid) n = 3;
        n = (n >= 7) ? 7 : (n >= 3) ? 3 : 1;
    }
    Sync::sync();
    apply_steps<Sync>(S, sb0, 0, n, b, lane, nt);
    if (lane == 0) { S.alpha[b] = 1.0; S.ls_iter[b] = 0; S.stage[b] = (n == 7) ? STAGE_LS7 : (n == 3) ? STAGE_LS3 : STAGE_LS0; }
    if (mode == 2) {
        enqueue_eval_async<Sync>(S, sb0, 0, n, b, lane, nt);
    } else {
        // the KKT kernel runs on its own stream NEXT TO the sweep of the running round: its
        // candidates join the queue of the next round
        // (chained round, kkt_same_round = 2: the round's second sweep consumes par ^ 1 - nothing is requested of the next round)
        const int par = (S.kkt_same_round == 1) ? S.WQ.par : (S.WQ.par ^ 1);
        // overlapped KKT kernel (kkt_same_round = 0): this round's decision kernel runs AFTER this kernel and would take the
        // freshly requested candidates (stage LS*, need_sweep set, done_count 0) for an evaluation of ITS round with parked
        // solves - it re-listed every one of them (round 3: each such slot sat twice on the next round's list, two residual
        // blocks per slot, and a batch of 128 rollouts with seven-candidate searches ran over the end of the list).  Mark 2
        // tells the decision stage "not yours yet".
        enqueue_evals(S, sb0, n, b, par, lane, nt, S.kkt_same_round == 0 ? 2 : 1);
        if (lane == 0) {
            for (int c = n; c < CS; ++c) S.need_sweep[sb0 + c] = 0;
            // (counter 0 = rollouts that need a sweep in the NEXT round: only the overlapped KKT kernel requests one there.  With the KKT
            //  stage in the same round as the evaluation of its candidates - small batches - the count made the host launch one more,
            //  empty round after the decision that ended a solve: three launches, ~40 us of a 0.55 ms single-rollout solve)
            if (S.kkt_same_round == 0) atomicAdd(&S.counters[0 * CPAD], 1);
        }
    }
}


// Sensitivities in effect for step i of line-search iterate `it` of rollout b.  The reference keeps ONE
// sensitivity block per knot (ip[t].dz) that every successful solve overwrites and a failed solve leaves
// untouched (implicit_dynamics.jl:169-176): after a failure the block still holds the most recent successful
// solve of that knot in evaluation order - an earlier step length of the same search, else the accepted
// evaluation of the previous Newton iteration.  With the step lengths evaluated speculatively in slots the
// same value is found by walking back through the slots of the search, then to dz_good.  (The result does
// not depend on how the evaluations were scheduled.)
__device__ __forceinline__ const double* dz_eff(const NewtonDev& S, int b, int it, int i) {
    const int H = S.dm.H;
    const size_t sb0 = (size_t)b * CS, blk = (size_t)S.nths * S.nd;
    int s = it;
    while (s >= 0 && S.ip_status[(sb0 + s) * H + i] == 0) --s;
    return s >= 0 ? S.dz + ((sb0 + s) * H + i) * blk : S.dz_good + ((size_t)b * H + i) * blk;
}

// residual! on evaluation slot sb (candidate trajectory, nu_cand, d, dz of that slot): writes res_cand of the
// slot and returns THIS THREAD's share of |r|_1 (the caller reduces all candidate slots in one pass)
// `absr` (optional, LDS, [N]): |r_e| of every entry instead of the per-thread partial sums - the caller then forms the partial
// sums of the canonical 256-thread pass itself, so the norm does not depend on how many threads computed the entries.
template <int NQ, int NU, bool CF>
__device__ double slot_residual(const NewtonDev& S, size_t sb, int b, const int* eff, int tid, int nt, double* absr = nullptr) {
    const cimpc_dims& m = S.dm;
    const int nq = NQ > 0 ? NQ : m.nq, nu = NQ > 0 ? NU : m.nu;      // NQ = 0: runtime dimensions (models without a compiled set)
    constexpr bool cf = CF;
    const int H = m.H, nc = m.nc, nb = m.nb, nr = S.nr;
    const int nd = cf ? nq + nc + nb : nq;       // compile-time in :configuration mode
    const int oq_in_block = cf ? nu + nc + nb : nu;
    double* r = S.res_cand + sb * S.N;
    const double* nuc = S.nu_cand + sb * H * nd;
    const int it = (int)(sb - (size_t)b * CS);      // line-search iterate held by this slot
    const size_t blk = (size_t)(2 * nq + nu) * nd, sb0 = (size_t)b * CS;
    // sensitivities in effect for step i: from the table the caller resolved once (LDS), or on the fly
    auto dzp = [&](int i) -> const double* {
        if (eff == nullptr) return dz_eff(S, b, it, i);
        const int s_ = eff[i];
        return s_ >= 0 ? S.dz + ((sb0 + s_) * H + i) * blk : S.dz_good + ((size_t)b * H + i) * blk;
    };
    if constexpr (NQ > 0 && !CF) {
        if (absr != nullptr && S.V == nullptr) {
            // :configuration + TrackingObjective (the hot configuration).  The pass is a chain of memory round trips, not arithmetic
            // (measured: 20 us per slot, 11 us for a lone workgroup, for 13 k multiply-adds), so it is written as ONE straight-line
            // body per thread - a u row, a q row and a d row, indices clamped instead of branched on - and every operand is requested
            // before the first multiply-add: one round trip for all three rows.  Same products, same summation order per row as the
            // general loop below.
            constexpr int nrc = NQ + NU, JM = NQ > NU ? NQ : NU;
            const double* dtn = S.dtn; const int dld = S.dtn_ld;
            const double* cu = S.cand.u + sb * H * NU; const double* ru = S.ref.u + (size_t)b * H * NU;
            const double* cqp = S.cand.q + sb * (H + 2) * NQ; const double* rqp = S.ref.q + (size_t)b * (H + 2) * NQ;
            const double* dd = S.d + sb * H * NQ;
            for (int j = tid; j < H * JM; j += nt) {
                // ---- u1[i]: R_i (u - u_ref) + du1_i^T nu_i
                const bool on_u = j < H * NU;
                const int ju = on_u ? j : 0, iu = ju / NU, cu_ = ju - iu * NU;
                const double* Rm = S.R + (size_t)iu * NU * NU + cu_; const double* uu = cu + iu * NU; const double* ur = ru + iu * NU;
                // (delta^T nu: the sweep's product where this slot's own solve of the step succeeded - the usual case -, else the
                //  dot product with the block dz_eff resolves; `own*` selects after the loads, which stay valid either way)
                const bool ownu = dtn != nullptr && eff != nullptr && eff[iu] == it;
                const double* A0 = dzp(iu) + (size_t)(2 * NQ + cu_) * NQ; const double* nv = nuc + iu * NQ;
                // ---- q2[i]: Q_i (q - q_ref) - nu_i + dq1_{i+1}^T nu_{i+1} + dq0_{i+2}^T nu_{i+2}
                const bool on_q = j < H * NQ;
                const int jq = on_q ? j : 0, i = jq / NQ, cq = jq - i * NQ;
                const int i1 = min(i + 1, H - 1), i2 = min(i + 2, H - 1);          // (clamped: the loads stay valid, the terms are dropped below)
                const double* Qm = S.Q + (size_t)i * NQ * NQ + cq; const double* qq = cqp + (i + 2) * NQ; const double* qr = rqp + (i + 2) * NQ;
                const bool own1 = dtn != nullptr && eff != nullptr && eff[i1] == it, own2 = dtn != nullptr && eff != nullptr && eff[i2] == it;
                const double* A1 = dzp(i1) + (size_t)(NQ + cq) * NQ; const double* n1 = nuc + i1 * NQ;
                const double* A2 = dzp(i2) + (size_t)cq * NQ; const double* n2 = nuc + i2 * NQ;
                double rv[NU], du_[NU], qv[NQ], dq_[NQ];
#pragma unroll
                for (int k = 0; k < NU; ++k) { rv[k] = Rm[k * NU]; du_[k] = uu[k] - ur[k]; }
#pragma unroll
                for (int k = 0; k < NQ; ++k) { qv[k] = Qm[k * NQ]; dq_[k] = qq[k] - qr[k]; }
                const double nui = nuc[i * NQ + cq];
                const double dv = dd[jq];                              // ---- rd[i] = d_i
                double su, s1, s2;
                if (dtn != nullptr) {
                    su = dtn[((size_t)sb * H + iu) * dld + 2 * NQ + cu_]; s1 = dtn[((size_t)sb * H + i1) * dld + NQ + cq]; s2 = dtn[((size_t)sb * H + i2) * dld + cq];
                }
                auto dot = [&](const double* A, const double* n_) { double a_ = 0.0;
#pragma unroll
                    for (int k = 0; k < NQ; ++k) a_ = fma(A[k], n_[k], a_);
                    return a_; };
                if (!ownu) su = dot(A0, nv);           // (rare: a failed solve's fallback block, or no products at all)
                if (!own1) s1 = dot(A1, n1);
                if (!own2) s2 = dot(A2, n2);
                double vu = 0.0, vq = 0.0;
#pragma unroll
                for (int k = 0; k < NU; ++k) vu = fma(rv[k], du_[k], vu);
                vu += su;
#pragma unroll
                for (int k = 0; k < NQ; ++k) vq = fma(qv[k], dq_[k], vq);
                vq -= nui;
                if (i + 1 < H) vq += s1;
                if (i + 2 < H) vq += s2;
                if (on_u) { r[iu * nrc + cu_] = vu; absr[iu * nrc + cu_] = fabs(vu); }
                if (on_q) { r[i * nrc + NU + cq] = vq; absr[i * nrc + NU + cq] = fabs(vq); r[H * nrc + jq] = dv; absr[H * nrc + jq] = fabs(dv); }
            }
            return 0.0;
        }
    }
    double part = 0.0;
    for (int e = tid; e < S.N; e += nt) {
        double v = 0.0;
        if (e >= H * nr) {                       // rd[i] = d_i
            v = S.d[sb * H * nd + (e - H * nr)];
        } else {
            const int i = e / nr, c = e - i * nr;
            if (c < nu) {                         // u1[i]: R_i (u - u_ref) + du1_i^T nu_i
                const double* Rm = S.R + (size_t)i * nu * nu;
                const double* uu = S.cand.u + (sb * H + i) * nu;
                const double* ur = S.ref.u + ((size_t)b * H + i) * nu;
#pragma unroll
                for (int k = 0; k < nu; ++k) v = fma(Rm[c + k * nu], uu[k] - ur[k], v);
                const double* A0 = dzp(i) + (size_t)(2 * nq + c) * nd;   // column c of du1_i
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < nq; ++k) s = fma(A0[k], nuc[i * nd + k], s);
                for (int k = nq; k < nd; ++k) s = fma(A0[k], nuc[i * nd + k], s);
                v += s;
            } else if (c >= oq_in_block) {        // q2[i]
                const int cq = c - oq_in_block;
                const double* Qm = S.Q + (size_t)i * nq * nq;
                const double* qq = S.cand.q + (sb * (H + 2) + i + 2) * nq;
                const double* qr = S.ref.q + ((size_t)b * (H + 2) + i + 2) * nq;
                if (S.V == nullptr) {
#pragma unroll
                    for (int k = 0; k < nq; ++k) v = fma(Qm[cq + k * nq], qq[k] - qr[k], v);
                } else {   // TrackingVelocityObjective (newton_residual.jl:221-281): q tracks q_ref + q_target,
                           // V_i penalises w_i = q_{i+2} - q_{i+1} (- v_target_i in :configuration mode)
                    const double* qt = S.q_target + (size_t)i * nq;
                    for (int k = 0; k < nq; ++k) v = fma(Qm[cq + k * nq], qq[k] - (qr[k] + qt[k]), v);
                    const double* Vi = S.V + (size_t)i * nq * nq;
                    const double* vt = S.v_target + (size_t)i * nq;
                    for (int k = 0; k < nq; ++k)
                        v = fma(Vi[cq + k * nq], qq[k] - qq[k - nq] - (cf ? 0.0 : vt[k]), v);
                    if (i + 1 < H) {
                        const double* Vn = Vi + nq * nq;
                        for (int k = 0; k < nq; ++k)
                            v = fma(-Vn[cq + k * nq], qq[k + nq] - qq[k] - (cf ? 0.0 : vt[k + nq]), v);
                    }
                }
                v -= nuc[i * nd + cq];                                      // rI[i] -= nu_i
                if (i + 1 < H) {                                            // dq1_{i+1}^T nu_{i+1}
                    const double* A1 = dzp(i + 1) + (size_t)(nq + cq) * nd;
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < nq; ++k) s = fma(A1[k], nuc[(i + 1) * nd + k], s);
                    for (int k = nq; k < nd; ++k) s = fma(A1[k], nuc[(i + 1) * nd + k], s);
                    v += s;
                }
                if (i + 2 < H) {                                            // dq0_{i+2}^T nu_{i+2}
                    const double* A2 = dzp(i + 2) + (size_t)cq * nd;
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < nq; ++k) s = fma(A2[k], nuc[(i + 2) * nd + k], s);
                    for (int k = nq; k < nd; ++k) s = fma(A2[k], nuc[(i + 2) * nd + k], s);
                    v += s;
                }
            } else if (c < nu + nc) {             // gamma1[i] (cf)
                const int cg = c - nu;
                const double* Cm = S.Cg + (size_t)i * nc * nc;
                const double* gg = S.cand.g + (sb * H + i) * nc;
                const double* gr = S.ref.g + ((size_t)b * H + i) * nc;
                for (int k = 0; k < nc; ++k) v = fma(Cm[cg + k * nc], gg[k] - gr[k], v);
                v -= nuc[i * nd + nq + cg];
            } else {                              // b1[i] (cf)
                const int cb = c - nu - nc;
                const double* Cm = S.Cb + (size_t)i * nb * nb;
                const double* bb = S.cand.b + (sb * H + i) * nb;
                const double* br = S.ref.b + ((size_t)b * H + i) * nb;
                for (int k = 0; k < nb; ++k) v = fma(Cm[cb + k * nb], bb[k] - br[k], v);
                v -= nuc[i * nd + nq + nc + cb];
            }
        }
        r[e] = v;
        if (absr != nullptr) absr[e] = fabs(v);
        else part += fabs(v);
    }
    return part;
}

// SPLIT (lock-step rounds): the stage runs as TWO launches.  SPLIT = 1 (resid_slot_kernel): one workgroup per (rollout,
// evaluation slot `my`) forms that slot's residual and its 1-norm - a rollout that evaluated seven step lengths used to serialise
// seven residual passes in one workgroup and set the length of the whole launch (measured: mean workgroup 42 us, longest 157 us =
// the launch).  SPLIT = 2 (resid_decide_kernel): one workgroup per rollout reads the norms and takes the decision.  The kernel
// boundary is the hand-over (an in-kernel hand-over needs agent-scope release fences = L2 write-backs on this multi-XCD part:
// measured 104 -> 210 us per launch).  The arithmetic per slot is unchanged (same thread -> entry map, same reduction tree), so
// results are bit-identical to the one-workgroup form (SPLIT = 0: the asynchronous kernel's residual job).
// -DCIMPC_RESID_PROF (diagnostic builds, scripts/resid_prof.py): clock accounting of the two launches of the decision stage,
// 100 MHz ticks summed over workgroups.  g_resid_prof[k][0] = workgroups, [k][1..12] = phases, [k][13] = lifetime sum, [k][14] = max
// lifetime; k = 0 slot kernel, 1 decision kernel.
#ifdef CIMPC_RESID_PROF
static __device__ unsigned long long g_resid_prof[4][16];      // rows 2, 3: the same sums over the SLOW workgroups only (lifetime > 30 us)
struct ResidProf {
    long long t0, t, acc[13]; int k;
    __device__ ResidProf(int k_) : k(k_) { t0 = t = wall_clock64(); for (int j = 0; j < 13; ++j) acc[j] = 0; }
    __device__ void mark(int j) { const long long n = wall_clock64(); acc[j] += n - t; t = n; }
    __device__ ~ResidProf() {
        if (threadIdx.x != 0) return;
        const long long te = wall_clock64();
        atomicAdd(&g_resid_prof[k][0], 1ull);
        for (int j = 1; j < 13; ++j) if (acc[j]) atomicAdd(&g_resid_prof[k][j], (unsigned long long)acc[j]);
        atomicAdd(&g_resid_prof[k][13], (unsigned long long)(te - t0));
        atomicMax(&g_resid_prof[k][14], (unsigned long long)(te - t0));
        if (te - t0 > 3000) {
            atomicAdd(&g_resid_prof[2 + k][0], 1ull);
            for (int j = 1; j < 13; ++j) if (acc[j]) atomicAdd(&g_resid_prof[2 + k][j], (unsigned long long)acc[j]);
            atomicAdd(&g_resid_prof[2 + k][13], (unsigned long long)(te - t0));
        }
    }
};
#define RPROF_BEGIN(k) ResidProf rprof_(k);
#define RPROF(j) rprof_.mark(j);
#else
#define RPROF_BEGIN(k)
#define RPROF(j)
#endif
constexpr int SLOT_ABS_MAX = 6144;      // longest Newton vector whose |r_e| fit the slot kernel's LDS scratch (48 KB); longer ones: 256-thread pass
template <int NQ, int NU, bool CF, bool ASYNC, int SPLIT = 0>
__device__ __forceinline__ void resid_decide_body(const NewtonDev& S, int b, double* red, double* rc, int* sh) {
    const cimpc_dims& m = S.dm;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int H = m.H;
    const size_t sb0 = (size_t)b * CS;
    if (!ASYNC && blockIdx.x == 0) {   // the queue of this round has been consumed: recycle it
        const int K = S.WQ.K, par = S.WQ.par;
        for (int k = tid; k < K; k += nt) { *qcount(S.WQ, par, k) = 0; *qhead(S.WQ, k) = 0; }
        if (tid < 8) S.counters_next[tid * CPAD] = 0;      // counter block of the next round
    }
    RPROF_BEGIN(1)
    // (SPLIT = 2: the per-slot scalars of the rollout are requested together with its stage, one round trip for the entry checks)
    int pre_dc = H, pre_ns = 0;
    double pre_rc = 0.0;
    if constexpr (SPLIT == 2) {
        if (tid < CS) { pre_dc = S.WQ.done_count[sb0 + tid]; pre_ns = S.need_sweep[sb0 + tid]; pre_rc = S.r_cand[sb0 + tid]; }
    }
    const int stage = S.stage[b];
    if (stage == STAGE_DONE || stage == STAGE_KKT) return;
    const int ncand = (stage == STAGE_LS1) ? 2 : (stage == STAGE_LS2) ? 4 : (stage == STAGE_LS7) ? 7 : (stage == STAGE_LS3) ? 3 : 1;
    const int it0 = (stage == STAGE_LS1) ? 1 : (stage == STAGE_LS2) ? 3 : 0;      // first iterate (= slot) of the batch
    if constexpr (SPLIT == 2) {
        const bool mine = tid >= it0 && tid < it0 + ncand;
        if (!__syncthreads_or(tid == it0 && pre_ns != 0)) return;    // nothing was evaluated for this rollout
        if (__syncthreads_or(tid == it0 && pre_ns == 2)) {           // requested by this round's KKT kernel for the NEXT round: already
            if (mine) S.need_sweep[sb0 + tid] = 1;                   // counted and listed there - from the next round on it is ours
            return;
        }
        if (__syncthreads_or(mine && pre_dc < H)) {                  // an interior-point solve of this evaluation is still parked: wait for the next round
            if (tid == 0) atomicAdd(&S.counters[0 * CPAD], 1);
            if (mine) list_slot(S, sb0 + tid, S.WQ.par ^ 1);         // its slots stay on the list
            return;
        }
        if (mine) rc[tid - it0] = pre_rc;
    } else {
        if (S.need_sweep[sb0 + it0] == 0) return;    // nothing was evaluated for this rollout
        if constexpr (!ASYNC) {
            if (S.need_sweep[sb0 + it0] == 2) {      // requested by this round's overlapped KKT kernel for the next round (see above)
                __syncthreads();
                if (tid < ncand) S.need_sweep[sb0 + it0 + tid] = 1;
                return;
            }
        }
        if constexpr (!ASYNC) {   // an interior-point solve of this evaluation is still parked: wait for the next round
            int pend = 0;
            if (tid < ncand) pend = (S.WQ.done_count[sb0 + it0 + tid] < H);
            if (__syncthreads_or(pend)) {
                if (tid == 0) atomicAdd(&S.counters[0 * CPAD], 1);
                return;
            }
        }
    }
    int& s_act = sh[0]; int& s_slot = sh[1]; int& s_iter = sh[2];
    constexpr int EFF_H = 128;                       // horizon steps the LDS table holds (longer horizons resolve on the fly)
    __shared__ int eff[(SPLIT ? 1 : CS) * EFF_H];
    const bool use_eff = H <= EFF_H;
    // effective sensitivity source of every step of candidate c (dz_eff), resolved ONCE into LDS: slot index or -1
    auto resolve_eff = [&](int c, int* row) {
        for (int i = tid; i < H && use_eff; i += nt) {
            int s_ = it0 + c;
            while (s_ >= 0 && S.ip_status[(sb0 + s_) * H + i] == 0) --s_;
            row[i] = s_;
        }
    };
    if constexpr (SPLIT == 2) {
        __syncthreads();      // rc[] (the slot kernel's norms) is in place
    } else {   // all candidate slots in this workgroup (the asynchronous kernel's residual job): slot after slot, each norm summed
               // as the canonical 256-thread pass sums it (see SPLIT = 1), ONE reduction tree for all of them
        constexpr int CT = 256;
        for (int c = 0; c < ncand; ++c) resolve_eff(c, eff + c * EFF_H);
        __syncthreads();
        double* absr = red + CS * CT;               // [N] scratch behind the partial sums (the caller's buffer holds CS * 256 + N doubles)
        for (int c = 0; c < ncand; ++c) {
            slot_residual<NQ, NU, CF>(S, sb0 + it0 + c, b, use_eff ? eff + c * EFF_H : nullptr, tid, nt, absr);
            __syncthreads();
            for (int j = tid; j < CT; j += nt) {
                double part = 0.0;
                for (int e = j; e < S.N; e += CT) part += absr[e];
                red[c * CT + j] = part;
            }
            __syncthreads();
        }
        for (int st = CT / 2; st > 0; st >>= 1) {
            for (int j = tid; j < st; j += nt)
                for (int c = 0; c < ncand; ++c) red[c * CT + j] += red[c * CT + j + st];
            __syncthreads();
        }
        if (tid < ncand) { rc[tid] = red[tid * CT]; S.r_cand[sb0 + it0 + tid] = red[tid * CT]; }
        __syncthreads();
    }
    RPROF(1)
    // ---- decision (newton.jl:198-280) ------------------------------------------------------
    if (tid == 0) {
        int act = 2, slot = 0, iter = 0;   // act: 0 accept initial evaluation, 1 accept step, 2 more candidates
        if (stage == STAGE_INIT) {
            act = 0;
        } else {
            const double rn2 = S.r_norm[b] * S.r_norm[b];
            for (int c = 0; c < ncand; ++c) {
                const double a = ls_alpha(it0 + c);
                if (!(rc[c] * rc[c] >= (1.0 - 0.001 * a) * rn2)) { act = 1; slot = c; iter = it0 + c; break; }
            }
            if (act == 2 && (stage == STAGE_LS2 || stage == STAGE_LS7)) {   // iter = 7 > 6: break out, halved alpha, last evaluation
                act = 1; slot = ncand - 1; iter = 7;
            }
        }
        s_act = act; s_slot = slot; s_iter = iter;
    }
    __syncthreads();
    RPROF(2)
    {   // statistics (parallel reduction).  Global counters: every evaluation that ran (speculative
        // ones included).  Per-rollout counters: only the evaluations the reference's sequential line
        // search would have performed (candidates up to and including the accepted one).
        const int nref = (s_act == 1 && s_iter <= 6) ? s_slot + 1 : ncand;
        int its = 0, fails = 0, its_ref = 0, fails_ref = 0;
        for (int k = tid; k < ncand * H; k += nt) {
            const int it = S.ip_iters[(sb0 + it0) * H + k], fl = (S.ip_status[(sb0 + it0) * H + k] == 0);
            its += it; fails += fl;
            if (k < nref * H) { its_ref += it; fails_ref += fl; }
        }
        // integer sums: inside a wavefront by lane exchange, then one partial per wavefront through the reduction buffer
        // (one barrier instead of the nine of a 256-wide tree)
        for (int o = 32; o > 0; o >>= 1) {
            its += __shfl_xor(its, o, 64); fails += __shfl_xor(fails, o, 64);
            its_ref += __shfl_xor(its_ref, o, 64); fails_ref += __shfl_xor(fails_ref, o, 64);
        }
        if ((tid & 63) == 0) {
            double* w = red + 4 * (tid >> 6);
            w[0] = (double)its; w[1] = (double)fails; w[2] = (double)its_ref; w[3] = (double)fails_ref;
        }
        __syncthreads();
        if (tid == 0) {
            long long t_its = 0, t_fails = 0, t_its_ref = 0, t_fails_ref = 0;
            for (int w = 0; w < (nt + 63) / 64; ++w) {
                t_its += (long long)red[4 * w]; t_fails += (long long)red[4 * w + 1];
                t_its_ref += (long long)red[4 * w + 2]; t_fails_ref += (long long)red[4 * w + 3];
            }
            S.ro_sweeps[b] += nref;
            S.ro_ip_iters[b] += (int)t_its_ref;
            S.ro_ip_fail[b] += (int)t_fails_ref;
            long long* st = S.stats + (size_t)b * 4;      // this rollout's line: one decision at a time per rollout
            st[0] += ncand; st[1] += (long long)ncand * H; st[2] += (long long)t_its; st[3] += t_fails;
        }
    }
    __syncthreads();
    RPROF(3)
    const int act = s_act, slot = s_slot, iter = s_iter;
    if (act == 2) {                       // next batch of candidates: (1/2, 1/4) or (1/8 .. 1/64)
        const int nstage = (stage == STAGE_LS0) ? STAGE_LS1 : STAGE_LS2;
        const int it1 = (nstage == STAGE_LS1) ? 1 : 3, nn = (nstage == STAGE_LS1) ? 2 : 4;      // iterates (= slots) it1 .. it1+nn-1
        if constexpr (ASYNC) {
            apply_steps<BlockSync>(S, sb0 + it1, it1, nn, b, tid, nt);
            if (tid == 0) S.stage[b] = nstage;
            enqueue_eval_async<BlockSync>(S, sb0, it1, nn, b, tid, nt);
            return;
        }
        if (tid == 0) for (int c = 0; c < CS; ++c) S.need_sweep[sb0 + c] = 0;
        __syncthreads();
        apply_steps<BlockSync>(S, sb0 + it1, it1, nn, b, tid, nt);
        RPROF(4)
        enqueue_evals(S, sb0 + it1, nn, b, S.WQ.par ^ 1, tid, nt);      // evaluated in the next round
        RPROF(5)
        if (tid == 0) {
            S.stage[b] = nstage;
            atomicAdd(&S.counters[0 * CPAD], 1);
        }
        return;
    }
    const double alpha = ls_alpha(iter);
    if (act == 1) {                       // accept: traj <- traj - alpha*Delta (newton.jl:273)
        // (iter = 7: the search was exhausted - the reference steps with the HALVED alpha = 2^-7, which no slot holds, and keeps the
        //  implicit-dynamics data of the last evaluation: newton.jl:255-262)
        if (iter <= 6) accept_candidate<BlockSync>(S, sb0 + it0 + slot, b, tid, nt);
        else apply_step<BlockSync>(S, S.traj, S.nu, (size_t)b, b, alpha, tid, nt);
    }
    RPROF(6)
    {   // res <- res_cand ; r_norm <- r_cand
        double* rr = S.res + (size_t)b * S.N;
        const double* r = S.res_cand + (sb0 + it0 + slot) * S.N;
        for (int e = tid; e < S.N; e += nt) rr[e] = r[e];
    }
    RPROF(7)
    {   // im_traj of the accepted evaluation: the sensitivities in effect (a failed step keeps what dz_eff resolves;
        // steps that resolve to dz_good itself stay as they are).  Source slots come from the table: plain
        // independent loads, two doubles per lane.
        const int blk = S.nths * S.nd, ita = it0 + slot;
        double* good = S.dz_good + (size_t)b * H * blk;
        {
        if constexpr (SPLIT == 2) {      // the table row of the ACCEPTED slot
            resolve_eff(slot, eff);
            __syncthreads();
        }
        bool lazy_commit = false;
        if constexpr (!ASYNC) lazy_commit = S.good_src != nullptr && use_eff;
        if (lazy_commit) {
            // Round 6: no copy - the index of the slot that holds each step's block (every entry is -1 = "in dz_good" at this point,
            // see NewtonDev::good_src); the rollout's next KKT stage, or the flush at the end of the solve, moves the blocks
            const int* ef = eff + (SPLIT ? 0 : slot * EFF_H);
            for (int i = tid; i < H; i += nt) S.good_src[(size_t)b * H + i] = ef[i];
        } else if (use_eff && (blk & 1) == 0) {
            const int* ef = eff + (SPLIT ? 0 : slot * EFF_H);
            const int hb = blk / 2;
            // thirteen independent 16-byte loads in flight per thread before the first store: the quadruped's 6600 pairs move in
            // two round trips (source and destination are distinct buffers, which the compiler cannot see: it kept one load -
            // store pair in flight, 10 us per accepted evaluation; with four in flight the blocks that accept - the slowest
            // of a round's decision launch, 40 us against 16 on average - still spent 21 us here)
            constexpr int UN = 13;
            // (static_for: with a `#pragma unroll` loop the staging array stayed in scratch memory - 80 bytes per lane)
            for (int e0 = tid; e0 < H * hb; e0 += UN * nt) {
                typedef double v2d __attribute__((ext_vector_type(2)));      // (HIP's double2 is a class: arrays of it are not promoted to registers)
                v2d v[UN]; int dst[UN];
                static_for<0, UN>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const int e = e0 + j * nt, ec = e < H * hb ? e : 0, i = ec / hb, s_ = ef[i];
                    dst[j] = (e < H * hb && s_ >= 0) ? i * blk + 2 * (ec - i * hb) : -1;
                    v[j] = reinterpret_cast<const v2d*>(S.dz + ((sb0 + (s_ >= 0 ? s_ : 0)) * H + i) * (size_t)blk)[ec - i * hb];
                });
                static_for<0, UN>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    if (dst[j] >= 0) *reinterpret_cast<v2d*>(good + dst[j]) = v[j];
                });
            }
        } else {
            for (int e = tid; e < H * blk; e += nt) {
                const int i = e / blk;
                good[e] = dz_eff(S, b, ita, i)[e - i * blk];
            }
        }
        }
    }
    RPROF(8)
    if (tid == 0) {
        const double rn = rc[slot];
        if (act == 1 && S.nlog != nullptr && S.newton_l[b] < NLOG) {      // status line of this iteration (print_status, newton.jl:290-301)
            double* e = S.nlog + ((size_t)b * NLOG + S.newton_l[b]) * 4;
            e[0] = alpha; e[1] = S.r_norm[b] / (double)S.N; e[2] = rn / (double)S.N; e[3] = (double)iter;
        }
        S.r_norm[b] = rn;
        S.cur_slot[b] = it0 + slot;
        S.alpha[b] = alpha;
        S.ls_iter[b] = iter;
        int l = S.newton_l[b];
        if (act == 1) {
            l += 1;
            S.newton_l[b] = l;
            const double be = S.beta[b];
            S.beta[b] = (iter > 6) ? fmin(be * 1.3, 1.0e2) : fmax(1.0e1, be / 1.3);
        }
        const bool done = (l >= S.max_iter) || (rn / (double)S.N < S.r_tol);
        for (int c = 0; c < CS; ++c) S.need_sweep[sb0 + c] = 0;
        if (done) {
            S.stage[b] = STAGE_DONE;
            if (!ASYNC && S.A.n_done != nullptr) atomicAdd(S.A.n_done, 1);    // finished rollouts of this solve (hybrid hand-off)
        } else {
            S.stage[b] = STAGE_KKT;
            if constexpr (!ASYNC) {
                const int pos = atomicAdd(&S.counters[1 * CPAD], 1);
                if (S.kkt_list != nullptr) S.kkt_list[(size_t)S.WQ.par * m.B + pos] = b;
            }
        }
        sh[3] = done;
    }
    if constexpr (ASYNC) {   // hand over: res / traj / scalars of this rollout are released to the KKT stage
        xfence(S.A.flags);
        __syncthreads();
        if (tid == 0) {
            if (sh[3]) {
                if (atomicAdd(S.A.n_done, 1) == S.A.B - 1) wake_all(S.A);     // the solve is over: everybody leaves
            } else {
                push_kkt_job(S.A, b);
            }
        }
    }
}

#define CIMPC_KKT_PRIO 2      // (a constant of the build: the -D override is gone with the experiment it served)
#define CIMPC_RESID_THREADS 256      // (a constant of the build: the -D override is gone with the experiment it served)
#define CIMPC_SLOT_THREADS 256      // (a constant of the build: the -D override is gone with the experiment it served)
// first launch of the stage: one workgroup per evaluated slot - its residual and 1-norm.  The slots come from the compact list the
// requesters of the round built (NewtonDev::slot_list).  Every scalar the block needs is requested in ONE batch of loads before the
// first branch.
template <int NQ, int NU, bool CF>
__global__ __launch_bounds__(CIMPC_SLOT_THREADS) void resid_slot_kernel(NewtonDev S, const int* list) {
    constexpr int CT = 256, EFF_H = 128;
    extern __shared__ __attribute__((aligned(16))) double red[];      // [CT + min(N, SLOT_ABS_MAX)]: a small footprint keeps every block of a
                                                                      // round resident at once (808 blocks per round on average at B = 512)
    __shared__ int eff[EFF_H];
    const int tid = threadIdx.x, nt = blockDim.x, H = S.dm.H;
    RPROF_BEGIN(0)
    int my, b;
    if (list != nullptr) {      // one block per requested slot
        const int e = list[blockIdx.x];
        b = e / CS; my = e - b * CS;
    } else {                    // one block per (rollout, slot) pair, slot-major
        my = (int)blockIdx.x / S.nb_launch; b = (int)blockIdx.x - my * S.nb_launch + S.b0;
    }
    const size_t sb0 = (size_t)b * CS, sb = sb0 + my;
    const bool use_eff = H <= EFF_H;
    const int ns_my = S.need_sweep[sb];
    const int stage = S.stage[b];
    const int dc = tid < CS ? S.WQ.done_count[sb0 + tid] : H;
    const int nsl = tid < CS ? S.need_sweep[sb0 + tid] : 0;
    int eff_i = -1;      // step i = tid: the last slot <= my whose solve of this step succeeded (dz_eff), -1 = dz_good
    if (use_eff && tid < H) {
#pragma unroll
        for (int s_ = 0; s_ < CS; ++s_) {
            const int ok = (s_ <= my) ? S.ip_status[(sb0 + s_) * H + tid] : 0;
            if (ok != 0) eff_i = s_;
        }
    }
    if (ns_my != 1 || stage == STAGE_DONE || stage == STAGE_KKT) return;      // (2 = requested for the next round, not evaluated yet)
    const int ncand = (stage == STAGE_LS1) ? 2 : (stage == STAGE_LS2) ? 4 : (stage == STAGE_LS7) ? 7 : (stage == STAGE_LS3) ? 3 : 1;
    const int it0 = (stage == STAGE_LS1) ? 1 : (stage == STAGE_LS2) ? 3 : 0;
    if (my < it0 || my >= it0 + ncand) return;
    const bool mine = tid >= it0 && tid < it0 + ncand;
    // nothing evaluated for this rollout, or one of its evaluations still has a parked solve (the decision waits a round)
    if (__syncthreads_or((tid == it0 && nsl != 1) || (mine && dc < H))) return;
    if (use_eff && tid < H) eff[tid] = eff_i;
    __syncthreads();
    RPROF(1)
    // |r_e| goes to LDS and the 1-norm is summed exactly as the canonical 256-thread pass sums it: partial j = entries j, j + 256,
    // ... in order, then the pairwise tree - the norm does not depend on the number of threads that formed the entries
    if (S.N <= SLOT_ABS_MAX) {
        double* absr = red + CT;
        slot_residual<NQ, NU, CF>(S, sb, b, use_eff ? eff : nullptr, tid, nt, absr);
        __syncthreads();
        RPROF(2)
        if (tid < CT) {
            double part = 0.0;
            for (int e = tid; e < S.N; e += CT) part += absr[e];
            red[tid] = part;
        }
    } else {
        const double part = tid < CT ? slot_residual<NQ, NU, CF>(S, sb, b, use_eff ? eff : nullptr, tid, CT) : 0.0;
        if (tid < CT) red[tid] = part;
    }
    __syncthreads();
    for (int st = CT / 2; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) S.r_cand[sb] = red[0];
    RPROF(3)
}
// second launch: one workgroup per rollout - decision, trajectory update, next requests.
// SINGLE = true: the whole stage in this one launch (residuals of the rollout's slots, then the decision - resid_decide_body with
// SPLIT = 0): small batches, where a second launch costs more than the serialised residual passes of a rollout (B = 1: one slot).
template <int NQ, int NU, bool CF, bool SINGLE = false>
__global__ __launch_bounds__(CIMPC_RESID_THREADS) void resid_decide_kernel(NewtonDev S) {
    extern __shared__ __attribute__((aligned(16))) double red[];      // SINGLE: [CS * 256 + N], else [2 * threads]
    __shared__ double rc[CS];
    __shared__ int sh[4];
    resid_decide_body<NQ, NU, CF, false, SINGLE ? 0 : 2>(S, (int)blockIdx.x + S.b0, red, rc, sh);
    // ---- epilogue: the LAST block to finish publishes the round's counters to host-mapped pinned
    //      memory (the host polls the stamp; no memcpy / event on the critical path)
    // (The release fence per block lets the host start the next round's KKT kernel on the second stream as soon as it sees the
    //  stamp.  Measured alternative: no fence, KKT stream ordered behind the END of this kernel by an event - this launch 53 -> 42 us,
    //  but the KKT kernel then starts later: 9.72 -> 9.82 ms per batch step.)
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const int ticket = atomicAdd(&S.counters[7 * CPAD], 1);
        if (ticket == (int)gridDim.x - 1) {
            const int n_sweep = atomicAdd(&S.counters[0 * CPAD], 0), n_kkt = atomicAdd(&S.counters[1 * CPAD], 0);
            volatile int* hm = S.host_flag;
            hm[0] = n_sweep;
            hm[1] = n_kkt;
            hm[4] = atomicAdd(&S.counters[2 * CPAD], 0);                                   // solves parked by this round
            hm[6] = atomicAdd(&S.counters[4 * CPAD], 0);                                   // evaluation slots of the next round
            hm[7] = atomicAdd(&S.counters[5 * CPAD], 0);                                   // ... of which newly requested (not waiting for a parked solve)
            hm[5] = S.A.n_done != nullptr ? atomicAdd(S.A.n_done, 0) : 0;            // rollouts finished so far
            __threadfence_system();
            hm[2] = S.round_stamp;
            __threadfence_system();
        }
    }
}

// -------------------------------------------------------------------------------------------
// KKT: condensed solve, one wavefront per rollout.  Every operand of the block recursion is
// staged in LDS tiles with leading dimension LD (>= nd, nq, nu; multiple of 4); global memory
// is touched only to stream the step's sensitivities / objective inverses in (coalesced) and
// to spill the factors L1_i, L2_i, L0_i^-1, y_i for the backward pass.
//
//   Y_ii     = du1 R^-1 du1^T + Q_i^-1 + dq1 Q_{i-1}^-1 dq1^T + dq0 Q_{i-2}^-1 dq0^T + rho I
//   Y_i,i-1  = -dq1_i Q_{i-1}^-1 + dq0_i Q_{i-2}^-1 dq1_{i-1}^T ,   Y_i,i-2 = -dq0_i Q_{i-2}^-1
//   beta_i   = (C P^-1 r_p - r_d)_i                      (methods.jl:386-446, 487-504)
//   L2_i = Y_i,i-2 L0_{i-2}^-T ; L1_i = (Y_i,i-1 - L2_i L1_{i-1}^T) L0_{i-1}^-T ;
//   L0_i L0_i^T = Y_ii - L1_i L1_i^T - L2_i L2_i^T        (compute_L!, methods.jl:466-485)
//   forward / backward block substitution, Delta_x = P^-1 (r_p - C^T dnu)   (:506-557)
// -------------------------------------------------------------------------------------------
struct KktArgs {
    const double* r;     // [B][N] right-hand side
    double* delta;       // [B][N]
    const double* beta;  // [B] or null (then beta_scalar)
    double beta_scalar;
    const int* stage;    // only rollouts with stage == STAGE_KKT (null = all)
    int finish;          // 1: set alpha/ls_iter/cand/stage after the solve (newton loop)
    const double* dz_override;   // [B][H][nths][nd] sensitivities to use instead of S.dz_good / S.dz (cf-mode reduction)
    const int* only_flag;        // [B] or null: run only the rollouts whose flag is non-zero (mixed-precision refinement / fallback)
};
// sensitivities a KKT solve reads: the accepted evaluation's (Newton loop), slot 0 of the last implicit_dynamics! (B1
// seam, no stage array), or an explicit buffer
__device__ __forceinline__ const double* kkt_dz(const NewtonDev& S, const KktArgs& K, int b, int H, int nths, int nd) {
    if (K.dz_override) return K.dz_override + (size_t)b * H * nths * nd;
    return K.stage ? S.dz_good + (size_t)b * H * nths * nd : S.dz + (size_t)b * CS * H * nths * nd;
}

__device__ __forceinline__ void lds_sync() { __syncthreads(); }

// number of LDS tiles used by kkt_kernel
constexpr int KKT_TILES = 21;

template <int NQ, int NU>
__global__ __launch_bounds__(64) void kkt_kernel_scalar(NewtonDev S, KktArgs K) {
    constexpr int LD = ((NQ > NU ? NQ : NU) + 3) & ~3;
    const cimpc_dims& m = S.dm;
    const int b = blockIdx.x + S.b0, lane = threadIdx.x;
    if (K.stage != nullptr && K.stage[b] != STAGE_KKT) return;
    constexpr int nq = NQ, nu = NU, nd = NQ, nr = NQ + NU, nths = 2 * NQ + NU;
    const int H = m.H;
    constexpr int T2D = LD * LD;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* A0 = sm;                  // du1_i      nd x nu
    double* A1 = A0 + T2D;            // dq1_i      nd x nq
    double* A2 = A1 + T2D;            // dq0_i      nd x nq
    double* A1p = A2 + T2D;           // dq1_{i-1}
    double* T0 = A1p + T2D;           // du1 Rinv_i
    double* T1 = T0 + T2D;            // dq1 Qinv_{i-1}
    double* T2 = T1 + T2D;            // dq0 Qinv_{i-2}
    double* Y0 = T2 + T2D;
    double* Y1 = Y0 + T2D;
    double* Y2 = Y1 + T2D;
    double* Lc = Y2 + T2D;            // chol factor of the step
    double* Lbuf[3] = {Lc + T2D, Lc + 2 * T2D, Lc + 3 * T2D};      // L0inv ring (i, i-1, i-2)
    double* L1buf[2] = {Lc + 4 * T2D, Lc + 5 * T2D};               // L1 ring (i, i-1)
    double* L2c = Lc + 6 * T2D;
    double* Qbuf[3] = {Lc + 7 * T2D, Lc + 8 * T2D, Lc + 9 * T2D};  // Qinv ring (i, i-1, i-2)
    double* Ri = Lc + 10 * T2D;       // Rinv_i   (tile 21)
    double* vec = sm + KKT_TILES * T2D;
    double* bet = vec;                // LD each
    double* ybuf[3] = {vec + LD, vec + 2 * LD, vec + 3 * LD};      // y ring
    double* rpu = vec + 4 * LD;       // r_p(u) of step i
    double* rpq[3] = {vec + 5 * LD, vec + 6 * LD, vec + 7 * LD};   // r_p(q) ring (i, i-1, i-2)
    double* tv = vec + 8 * LD;
    const double* rb = K.r + (size_t)b * S.N;
    const double beta = K.beta ? K.beta[b] : K.beta_scalar;
    const double rho = (double)H * beta * S.kappa;     // newton_jacobian.jl:169-186 quirk
    // Newton loop: the accepted evaluation's sensitivities; B1 seam (no stage): slot 0 of the last implicit_dynamics!
    const double* dzb = kkt_dz(S, K, b, H, nths, nd);
    constexpr int n2 = nd * nd;
    double* ws = S.kkt_ws + (size_t)b * H * (3 * n2 + nd);
    const int oq = nu;   // offset of q2 inside a primal block (:configuration)

    // software prefetch of the next step's operands (global -> registers), consumed at the top
    // of the following iteration: hides the ~1 us HBM/L2 latency behind the block factorization
    constexpr int PF_DZ = (nd * nths + 63) / 64, PF_Q = (nq * nq + 63) / 64, PF_R = (nu * nu + 63) / 64;
    double pf_dz[PF_DZ], pf_q[PF_Q], pf_r[PF_R], pf_rp = 0.0, pf_rd = 0.0;
    auto prefetch = [&](int i) {
        if (i >= H) return;
        const double* dzi = dzb + (size_t)i * nths * nd;
#pragma unroll
        for (int j = 0; j < PF_DZ; ++j) { const int k = lane + 64 * j; pf_dz[j] = (k < nd * nths) ? dzi[k] : 0.0; }
#pragma unroll
        for (int j = 0; j < PF_Q; ++j) { const int k = lane + 64 * j; pf_q[j] = (k < nq * nq) ? S.Qinv[(size_t)i * nq * nq + k] : 0.0; }
#pragma unroll
        for (int j = 0; j < PF_R; ++j) { const int k = lane + 64 * j; pf_r[j] = (k < nu * nu) ? S.Rinv[(size_t)i * nu * nu + k] : 0.0; }
        pf_rp = (lane < nr) ? rb[i * nr + lane] : 0.0;                 // [u | q2] block of r_p
        pf_rd = (lane < nd) ? rb[H * nr + i * nd + lane] : 0.0;
    };
#ifdef CIMPC_KKT_PROF
    long long pt[16] = {0}; long long tp = clock64();
#define KPROF(j) { const long long tn = clock64(); pt[j] += tn - tp; tp = tn; }
#else
#define KPROF(j)
#endif
    prefetch(0);
    for (int i = 0; i < H; ++i) {
        double* Li = Lbuf[i % 3];      double* Li1 = Lbuf[(i + 2) % 3];  double* Li2 = Lbuf[(i + 1) % 3];
        double* L1c = L1buf[i % 2];    double* L1p = L1buf[(i + 1) % 2];
        double* Qi0 = Qbuf[i % 3];     double* Qi1 = Qbuf[(i + 2) % 3];  double* Qi2 = Qbuf[(i + 1) % 3];
        double* yc = ybuf[i % 3];      double* y1 = ybuf[(i + 2) % 3];   double* y2 = ybuf[(i + 1) % 3];
        double* q0r = rpq[i % 3];      double* q1r = rpq[(i + 2) % 3];   double* q2r = rpq[(i + 1) % 3];
        // ---- phase 1: stream step i in (A1p keeps dq1_{i-1}: swap roles of A1/A1p) ----------
        {
            double* t = A1; A1 = A1p; A1p = t;
        }
        // registers -> LDS (data of step i was fetched while step i-1 was being factored)
#pragma unroll
        for (int j = 0; j < PF_DZ; ++j) {
            const int k = lane + 64 * j;
            if (k < nd * nths) {
                const int r = k % nd, c = k / nd;
                double* dst = (c < nq) ? (A2 + c * LD) : (c < 2 * nq) ? (A1 + (c - nq) * LD) : (A0 + (c - 2 * nq) * LD);
                dst[r] = pf_dz[j];
            }
        }
#pragma unroll
        for (int j = 0; j < PF_Q; ++j) {
            const int k = lane + 64 * j;
            if (k < nq * nq) Qi0[(k % nq) + (k / nq) * LD] = pf_q[j];
        }
#pragma unroll
        for (int j = 0; j < PF_R; ++j) {
            const int k = lane + 64 * j;
            if (k < nu * nu) Ri[(k % nu) + (k / nu) * LD] = pf_r[j];
        }
        if (lane < nu) rpu[lane] = pf_rp;
        else if (lane < nu + nq) q0r[lane - nu] = pf_rp;
        const double rd_i = pf_rd;
        lds_sync();
        prefetch(i + 1);
        KPROF(1)
        // ---- phase 2: T0 = A0 Ri, T1 = A1 Qi1, T2 = A2 Qi2 -----------------------------------
        for (int idx = lane; idx < nd * (nu + 2 * nq); idx += 64) {
            const int r = idx % nd;
            int c = idx / nd;
            double s = 0.0;
            if (c < nu) {
                for (int k = 0; k < nu; ++k) s = fma(A0[r + k * LD], Ri[k + c * LD], s);
                T0[r + c * LD] = s;
            } else if (c < nu + nq) {
                c -= nu;
                if (i >= 1) for (int k = 0; k < nq; ++k) s = fma(A1[r + k * LD], Qi1[k + c * LD], s);
                T1[r + c * LD] = s;
            } else {
                c -= nu + nq;
                if (i >= 2) for (int k = 0; k < nq; ++k) s = fma(A2[r + k * LD], Qi2[k + c * LD], s);
                T2[r + c * LD] = s;
            }
        }
        lds_sync();
        KPROF(2)
        // ---- phase 3: Y0, Y1, Y2, beta_i -----------------------------------------------------
        for (int idx = lane; idx < 3 * n2 + nd; idx += 64) {
            if (idx < n2) {
                const int r = idx % nd, c = idx / nd;
                double s = Qi0[r + c * LD] + ((r == c) ? rho : 0.0);
                for (int k = 0; k < nu; ++k) s = fma(T0[r + k * LD], A0[c + k * LD], s);
                if (i >= 1) for (int k = 0; k < nq; ++k) s = fma(T1[r + k * LD], A1[c + k * LD], s);
                if (i >= 2) for (int k = 0; k < nq; ++k) s = fma(T2[r + k * LD], A2[c + k * LD], s);
                Y0[r + c * LD] = s;
            } else if (idx < 2 * n2) {
                const int e = idx - n2, r = e % nd, c = e / nd;
                double s = -T1[r + c * LD];
                if (i >= 2) for (int k = 0; k < nq; ++k) s = fma(T2[r + k * LD], A1p[c + k * LD], s);
                Y1[r + c * LD] = s;
            } else if (idx < 3 * n2) {
                const int e = idx - 2 * n2, r = e % nd, c = e / nd;
                Y2[r + c * LD] = -T2[r + c * LD];
            } else {
                const int r = idx - 3 * n2;
                double s = 0.0;
                for (int k = 0; k < nu; ++k) s = fma(T0[r + k * LD], rpu[k], s);
                double t = 0.0;
                for (int k = 0; k < nq; ++k) t = fma(Qi0[r + k * LD], q0r[k], t);
                s -= t;
                if (i >= 1) { t = 0.0; for (int k = 0; k < nq; ++k) t = fma(T1[r + k * LD], q1r[k], t); s += t; }
                if (i >= 2) { t = 0.0; for (int k = 0; k < nq; ++k) t = fma(T2[r + k * LD], q2r[k], t); s += t; }
                bet[r] = s;            // - r_d[i] is applied below by the lanes that prefetched it
            }
        }
        lds_sync();
        KPROF(3)
        if (lane < nd) bet[lane] -= rd_i;
        // ---- phase 4: L2c = Y2 Li2^T (Li2 lower triangular: k <= c) --------------------------
        if (i >= 2) {
            for (int idx = lane; idx < n2; idx += 64) {
                const int r = idx % nd, c = idx / nd;
                double s = 0.0;
                for (int k = 0; k <= c; ++k) s = fma(Y2[r + k * LD], Li2[c + k * LD], s);
                L2c[r + c * LD] = s;
            }
            lds_sync();
            // ---- phase 5: Y1 -= L2c L1p^T ----------------------------------------------------
            for (int idx = lane; idx < n2; idx += 64) {
                const int r = idx % nd, c = idx / nd;
                double s = Y1[r + c * LD];
                for (int k = 0; k < nd; ++k) s = fma(-L2c[r + k * LD], L1p[c + k * LD], s);
                Y1[r + c * LD] = s;
            }
            lds_sync();
        }
        KPROF(4)
        // ---- phase 6: L1c = Y1 Li1^T ----------------------------------------------------------
        if (i >= 1) {
            for (int idx = lane; idx < n2; idx += 64) {
                const int r = idx % nd, c = idx / nd;
                double s = 0.0;
                for (int k = 0; k <= c; ++k) s = fma(Y1[r + k * LD], Li1[c + k * LD], s);
                L1c[r + c * LD] = s;
            }
            lds_sync();
        }
        KPROF(5)
        // ---- phase 7: Lc = Y0 - L1c L1c^T - L2c L2c^T (lower triangle) ------------------------
        for (int idx = lane; idx < n2; idx += 64) {
            const int r = idx % nd, c = idx / nd;
            if (r < c) continue;
            double s = Y0[r + c * LD];
            if (i >= 1) for (int k = 0; k < nd; ++k) s = fma(-L1c[r + k * LD], L1c[c + k * LD], s);
            if (i >= 2) for (int k = 0; k < nd; ++k) s = fma(-L2c[r + k * LD], L2c[c + k * LD], s);
            Lc[r + c * LD] = s;
        }
        lds_sync();
        KPROF(6)
        // ---- phase 8/9: Cholesky of Lc and its inverse --------------------------------------
        if constexpr (nd <= 16) {
            // register version: lane rl of each DPP row holds ROW rl of the matrix; the pivot and
            // the multipliers travel by row_newbcast, no LDS round trip inside the recursion.
            using LG = LaneGroup<16>;
            const int rl = lane & 15;
            double a[nd], invd[nd];
            static_for<0, nd>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                a[c] = (rl < nd) ? Lc[rl + c * LD] : ((c == rl) ? 1.0 : 0.0);
            });
            static_for<0, nd>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const double dk = LG::template bcast<k>(a[k]);
                const double inv = fast_rsqrt(dk);
                invd[k] = inv;                       // 1 / L[k,k], identical in every lane
                a[k] *= inv;                         // lane k: sqrt(dk); lanes r > k: L[r,k]
                static_for<k + 1, nd>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    a[c] = fma(-a[k], LG::template bcast<c>(a[k]), a[c]);
                });
            });
            // column rl of Li = L^-1 by forward substitution; L[r,k] broadcast from lane r
            double xc[nd];
            static_for<0, nd>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                double acc = 0.0;
                static_for<0, r>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    acc = fma(LG::template bcast<r>(a[k]), xc[k], acc);
                });
                xc[r] = (((r == rl) ? 1.0 : 0.0) - acc) * invd[r];
            });
            if (lane < nd) {
                static_for<0, nd>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    Li[r + lane * LD] = xc[r];
                });
            }
        } else {
            for (int k = 0; k < nd; ++k) {
                const double dkk = sqrt(Lc[k + k * LD]);
                const double inv = 1.0 / dkk;
                lds_sync();
                for (int r = k + lane; r < nd; r += 64) Lc[r + k * LD] = (r == k) ? dkk : Lc[r + k * LD] * inv;
                lds_sync();
                const int rem = nd - k - 1;
                for (int idx = lane; idx < rem * rem; idx += 64) {
                    const int r = k + 1 + idx % rem, c = k + 1 + idx / rem;
                    if (r >= c) Lc[r + c * LD] = fma(-Lc[r + k * LD], Lc[c + k * LD], Lc[r + c * LD]);
                }
                lds_sync();
            }
            for (int c = lane; c < nd; c += 64) {
                double xi[nd];
                for (int r = 0; r < nd; ++r) {
                    double s = (r == c) ? 1.0 : 0.0;
                    for (int k = 0; k < r; ++k) s = fma(-Lc[r + k * LD], (k >= c) ? xi[k] : 0.0, s);
                    xi[r] = (r < c) ? 0.0 : s / Lc[r + r * LD];
                }
                for (int r = 0; r < nd; ++r) Li[r + c * LD] = xi[r];
            }
        }
        KPROF(7)
        for (int r = lane - 32; r >= 0 && r < nd; r += 64) {     // lanes 32.. do the rhs update meanwhile
            double s = bet[r];
            if (i >= 1) for (int k = 0; k < nd; ++k) s = fma(-L1c[r + k * LD], y1[k], s);
            if (i >= 2) for (int k = 0; k < nd; ++k) s = fma(-L2c[r + k * LD], y2[k], s);
            tv[r] = s;
        }
        lds_sync();
        KPROF(8)
        // ---- phase 10: y_i = Li * tv ; spill factors ------------------------------------------
        for (int r = lane; r < nd; r += 64) {
            double s = 0.0;
            for (int k = 0; k <= r; ++k) s = fma(Li[r + k * LD], tv[k], s);
            yc[r] = s;
        }
        double* wsi = ws + (size_t)i * (3 * n2 + nd);
        for (int k = lane; k < n2; k += 64) {
            const int r = k % nd, c = k / nd;
            wsi[k] = (i >= 1) ? L1c[r + c * LD] : 0.0;
            wsi[n2 + k] = (i >= 2) ? L2c[r + c * LD] : 0.0;
            wsi[2 * n2 + k] = Li[r + c * LD];
        }
        lds_sync();
        for (int k = lane; k < nd; k += 64) wsi[3 * n2 + k] = yc[k];
        KPROF(9)
    }
    __threadfence_block();
    lds_sync();
    // ---- backward substitution: dnu_i = Li_i^T (y_i - L1_{i+1}^T dnu_{i+1} - L2_{i+2}^T dnu_{i+2})
    double* D = K.delta + (size_t)b * S.N;
    double* Wt[3] = {A0, A1, A2};            // LDS staging of (L1_{i+1}, L2_{i+2}, Li_i)
    double* dnr[3] = {ybuf[0], ybuf[1], ybuf[2]};
    for (int i = H - 1; i >= 0; --i) {
        const double* wsi = ws + (size_t)i * (3 * n2 + nd);
        double* dnc = dnr[i % 3]; double* dn1 = dnr[(i + 1) % 3]; double* dn2 = dnr[(i + 2) % 3];
        for (int k = lane; k < n2; k += 64) {
            const int r = k % nd, c = k / nd;
            Wt[2][r + c * LD] = wsi[2 * n2 + k];
            if (i + 1 < H) Wt[0][r + c * LD] = ws[(size_t)(i + 1) * (3 * n2 + nd) + k];
            if (i + 2 < H) Wt[1][r + c * LD] = ws[(size_t)(i + 2) * (3 * n2 + nd) + n2 + k];
        }
        for (int k = lane; k < nd; k += 64) bet[k] = wsi[3 * n2 + k];
        lds_sync();
        for (int r = lane; r < nd; r += 64) {
            double s = bet[r];
            if (i + 1 < H) for (int k = 0; k < nd; ++k) s = fma(-Wt[0][k + r * LD], dn1[k], s);
            if (i + 2 < H) for (int k = 0; k < nd; ++k) s = fma(-Wt[1][k + r * LD], dn2[k], s);
            tv[r] = s;
        }
        lds_sync();
        for (int r = lane; r < nd; r += 64) {
            double s = 0.0;
            for (int k = r; k < nd; ++k) s = fma(Wt[2][k + r * LD], tv[k], s);
            dnc[r] = s;
            D[H * nr + i * nd + r] = s;
        }
        lds_sync();
    }
    __threadfence_block();
    lds_sync();
    KPROF(10)
    // ---- primal recovery: Delta_x = P^-1 (r_p - C^T dnu) ----------------------------------
    const double* dn = D + H * nr;
    for (int i = 0; i < H; ++i) {
        const double* dzi = dzb + (size_t)i * nths * nd;
        for (int c = lane; c < nu + nq; c += 64) {
            if (c < nu) {
                double s = 0.0;
                const double* a0 = dzi + (size_t)(2 * nq + c) * nd;
                for (int k = 0; k < nd; ++k) s = fma(a0[k], dn[i * nd + k], s);
                tv[c] = rb[i * nr + c] - s;
            } else {
                const int cq = c - nu;
                double s = -dn[i * nd + cq];
                if (i + 1 < H) {
                    const double* a1 = dzb + ((size_t)(i + 1) * nths + nq + cq) * nd;
                    double t = 0.0;
                    for (int k = 0; k < nd; ++k) t = fma(a1[k], dn[(i + 1) * nd + k], t);
                    s += t;
                }
                if (i + 2 < H) {
                    const double* a2 = dzb + ((size_t)(i + 2) * nths + cq) * nd;
                    double t = 0.0;
                    for (int k = 0; k < nd; ++k) t = fma(a2[k], dn[(i + 2) * nd + k], t);
                    s += t;
                }
                tv[c] = rb[i * nr + oq + cq] - s;
            }
        }
        lds_sync();
        for (int c = lane; c < nu + nq; c += 64) {
            double s = 0.0;
            if (c < nu) {
                const double* Rm = S.Rinv + (size_t)i * nu * nu;
                for (int k = 0; k < nu; ++k) s = fma(Rm[c + k * nu], tv[k], s);
                D[i * nr + c] = s;
            } else {
                const int cq = c - nu;
                const double* Qm = S.Qinv + (size_t)i * nq * nq;
                for (int k = 0; k < nq; ++k) s = fma(Qm[cq + k * nq], tv[nu + k], s);
                D[i * nr + oq + cq] = s;
            }
        }
        lds_sync();
    }
    KPROF(11)
#ifdef CIMPC_KKT_PROF
    if (lane == 0 && b == 0) for (int j = 0; j < 16; ++j) ((long long*)S.stats)[8 + j] = pt[j];      // (diagnostic builds: overwrites the statistics of rollouts 2..5)
#endif
    if (K.finish) {
        __threadfence_block();
        start_line_search<BlockSync>(S, b, 1, lane, 64);
    }
}


// -------------------------------------------------------------------------------------------
// MFMA version (nq, nu <= 16): every block product of the recursion is C (+)= X * Y^T on
// 16x16 LDS tiles, executed by v_mfma_f64_16x16x4_f64 (one wavefront = one rollout).  The
// operands are fed as A := Y, B := X so that the accumulator holds C^T-by-MFMA-layout = C with
// lane -> row: lane l owns C[l&15][(l>>4) + 4*reg], which stores to the column-major tile with
// consecutive lanes on consecutive addresses (no LDS bank conflicts).  Padded rows / columns
// of every tile are kept at exactly zero, so K can always be rounded up to a multiple of 4.
// -------------------------------------------------------------------------------------------
using d4 = __attribute__((ext_vector_type(4))) double;
constexpr int TL = 16;            // tile leading dimension
constexpr int TSZ = TL * TL;      // doubles per tile
constexpr int KKT_MFMA_TILES = 22;

// Tiles are TLD x TLD column-major (TLD = 16, or 24 for the larger models); a product C (+)= X * Y^T runs as
// NB x NB sub-blocks of the 16x16x4 MFMA (NB = ceil(TLD / 16)), rows / columns beyond TLD masked to zero.
// F32 = true: the SAME block products on v_mfma_f32_16x16x4_f32 (BASELINE configs[4]: "fp32 mixed-precision Schur GEMM on
// MFMA"): operands are rounded to fp32 when they leave the fp64 LDS tiles, the accumulator is fp32, results are widened
// when stored back; Cholesky, triangular inverses, substitutions and the right-hand side stay fp64.  The fp32 MFMA keeps
// four CONSECUTIVE rows of D per lane (D[4 (l>>4) + reg][l & 15]) where the fp64 one interleaves them
// (D[(l>>4) + 4 reg][l & 15]): only the column map of ld / st / add_diag differs.  The caller refines in fp64
// (newton_kernels.hip: launch_kkt_mixed).
using f4 = __attribute__((ext_vector_type(4))) float;
template <int NB, bool F32 = false>
struct TAcc {
    std::conditional_t<F32, f4, d4> v[NB][NB];
};
template <bool F32>
__device__ __forceinline__ constexpr int tile_col(int J, int lk, int r) { return 16 * J + (F32 ? 4 * lk + r : lk + 4 * r); }
template <int NB, bool F32 = false>
__device__ __forceinline__ TAcc<NB, F32> tile_zero() {
    TAcc<NB, F32> z;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) z.v[I][J][r] = 0;
    return z;
}
template <int KB, bool NEG, int TLD, bool F32 = false>
__device__ __forceinline__ TAcc<(TLD + 15) / 16, F32> tile_mma(const double* X, const double* Y, TAcc<(TLD + 15) / 16, F32> acc, int li, int lk) {
    constexpr int NB = (TLD + 15) / 16;
    constexpr bool MASK = (TLD % 16) != 0;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        double xa[NB], ya[NB];
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            const int r = 16 * I + li;
            const bool ok = !MASK || r < TLD;
            xa[I] = ok ? X[r + (4 * kb + lk) * TLD] : 0.0;
            ya[I] = ok ? Y[r + (4 * kb + lk) * TLD] : 0.0;
            if (NEG) ya[I] = -ya[I];
        }
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = 0; J < NB; ++J) {
                if constexpr (F32) acc.v[I][J] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)ya[J], (float)xa[I], acc.v[I][J], 0, 0, 0);
                else acc.v[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[J], xa[I], acc.v[I][J], 0, 0, 0);
            }
    }
    return acc;
}
template <int TLD, bool F32 = false>
__device__ __forceinline__ TAcc<(TLD + 15) / 16, F32> tile_ld(const double* C, int li, int lk) {
    constexpr int NB = (TLD + 15) / 16;
    constexpr bool MASK = (TLD % 16) != 0;
    TAcc<NB, F32> a;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * I + li, col = tile_col<F32>(J, lk, r);
                a.v[I][J][r] = (!MASK || (row < TLD && col < TLD)) ? C[row + col * TLD] : 0.0;
            }
    return a;
}
template <int TLD, bool F32 = false>
__device__ __forceinline__ void tile_st(double* C, const TAcc<(TLD + 15) / 16, F32>& a, int li, int lk) {
    constexpr int NB = (TLD + 15) / 16;
    constexpr bool MASK = (TLD % 16) != 0;
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * I + li, col = tile_col<F32>(J, lk, r);
                if (!MASK || (row < TLD && col < TLD)) C[row + col * TLD] = (double)a.v[I][J][r];
            }
}
template <int NB, bool F32 = false>
__device__ __forceinline__ TAcc<NB, F32> tile_neg(TAcc<NB, F32> a) {
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int J = 0; J < NB; ++J) a.v[I][J] = -a.v[I][J];
    return a;
}
// adds rho to the diagonal entries (row == col < n) this lane holds
template <int NB, bool F32 = false>
__device__ __forceinline__ void tile_add_diag(TAcc<NB, F32>& a, double rho, int n, int li, int lk) {
#pragma unroll
    for (int I = 0; I < NB; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * I + li, col = tile_col<F32>(I, lk, r);
            if (row == col && row < n) a.v[I][I][r] += rho;
        }
}
// y[r] = sum_k M[r + k*TLD] * x[k]  (or M^T), one output per lane, operands in LDS
template <int KN, bool TRANS, int TLD>
__device__ __forceinline__ double tile_mv(const double* Mt, const double* x, int r) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k + 1 < KN; k += 2) {
        s0 = fma(TRANS ? Mt[k + r * TLD] : Mt[r + k * TLD], x[k], s0);
        s1 = fma(TRANS ? Mt[k + 1 + r * TLD] : Mt[r + (k + 1) * TLD], x[k + 1], s1);
    }
    if (KN & 1) s0 = fma(TRANS ? Mt[KN - 1 + r * TLD] : Mt[r + (KN - 1) * TLD], x[KN - 1], s0);
    return s0 + s1;
}

// Broadcast of lane K's value for the in-register Cholesky.  TL = 16: four copies of the factorisation on the four
// DPP rows; wider tiles: lane = row over the whole wavefront, broadcasts through v_readlane (ONE matrix per
// wave: the source lane is uniform).
template <int TLD>
struct KktBcast {
    template <int K2>
    static __device__ __forceinline__ double bcast(double v) {
        if constexpr (TLD <= 16) return LaneGroup<16>::template bcast<K2>(v);
        else {
            const long long u = __double_as_longlong(v);
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, K2);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), K2);
            return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        }
    }
};

// PIPE = 1: the whole recursion on one wavefront.  PIPE = 2: the forward pass is software-pipelined over TWO
// wavefronts of the workgroup - wave 0 runs stage A of step t (operands -> LDS, the products with the inverse
// weights, Y_ii / Y_i,i-1 / L2_i / beta_i: everything that needs factors of step <= t-2 only) while wave 1 runs
// stage B of step t-1 (L1, the Cholesky factor and its inverse, y, spill); one workgroup barrier per step.  The
// stages hand Y_ii, Y_i,i-1, L2_i, beta_i over in LDS tiles double-buffered by step parity.  A step then costs
// max(A, B) instead of A + B; the backward pass and the recovery stay on wave 0.
constexpr int KKT_PIPE_TILES = 27;       // 22 + second L2 tile + two Y_ii and two Y_i,i-1 hand-over tiles
// PIPE = 3 (round 3): a third wavefront takes everything that is NOT on the recursion's dependency chain off the other two - stage
// C of step t-2 (the products W1 = L1 L0^-1, W2 = L2 L0^-1 and yhat = L0^-T y the backward pass runs on, and their spill) next to
// stage B of step t-1 (the chain: L1, the Cholesky factor, its inverse, y) and stage A of step t.
constexpr int KKT_PIPE3_TILES = 29;      // 27 + third L2 slot + fourth slot of the L0^-T ring
// TW = 1 / 2 (round 5): the TWISTED (two-ended) factorisation - two workgroups per rollout, each a three-stage pipeline on
// its own end of the block penta-diagonal matrix (newton_structure_solver/methods.jl:466-557 run from both ends).  TW = 1 is the
// TOP chain: the one-ended recursion over rows 0 .. m+1, whose last two rows first receive what the rows eliminated from the
// bottom contribute.  TW = 2 is the BOTTOM chain: chain step i = row H-1-i of the reversed matrix, nb = H-m-2 full steps and two
// trace steps (rows m+1, m: L2', L1' only) whose products S00, S11, S10^T, c0, c1 go to the top chain through global memory;
// the substitutions run outwards from the middle (the bottom chain receives dnu_{m+1}, dnu_m), each chain recovers its share of
// the primal rows, the chain that finishes last starts the line search.  CPU statement with the same rings and index algebra:
// the test oracle (newton.py: kkt_solve_condensed_twisted_device).
constexpr int KKT_TW_TILES = 32;         // 29 + two more slots of the dq0 ring + the tile of Qinv_j dq0_{j+2}^T (bottom chain)
// rows the top chain eliminates before the two middle rows: the bottom chain (two more products per stage A) gets the shorter half,
// so that its traces arrive when the top chain reaches row m.  Measured stage times (profiles/r05/twisted_prof_d.log): 16-wide
// tiles 1.95 us per tick of the top chain against 2.2 of the bottom chain -> nb = (H - 8) / 2; 24-wide tiles (2 x 2 MFMA blocks
// per product) 7.8 against 10.3 us -> nb = (H - 10) / 2
// duo (round 6, kkt_kernel_duo): both chains are ONE-wave bodies (stage A + stage B in sequence, ~6.5 us per step) - the bottom chain's
// two extra products weigh less against a whole step than against a pipelined stage A: nb = (H - 4) / 2
__host__ __device__ inline int kkt_tw_split(int H, int nb_override = 0, bool wide = false, bool duo = false) {
    int nb = nb_override > 0 ? nb_override : (H - (duo ? 4 : wide ? 10 : 8)) / 2;
    if (nb < 2) nb = 2;
    if (nb > H - 4) nb = H - 4;
    return H - 2 - nb;
}
// shorter horizons keep the one-ended kernels: at H = 20 (hopper, BASELINE configs[1]) the two forms take the same time - 22 us
// against 24 us per launch, profiles/r05/loop_trace_hopper_* - and the hand-overs are all that the second workgroup adds
constexpr int KKT_TW_MIN_H = 24;
// per-rollout exchange block of the two chains: [S00 | S11 | S10^T] (nd x nd each), c0, c1, dnu_{m+1}, dnu_m
__host__ __device__ constexpr int kkt_tw_xch_doubles(int nd) { return 3 * nd * nd + 4 * nd; }
constexpr int KKT_TW_FLAGS = 32;         // ints per rollout (one 128-byte line): [0] traces ready, [1] middle dnu ready, [2] chains finished, [3] a hand-over of
                                         // this launch timed out; [4..7] the same four words of the banded twisted kernel (kkt_dense.hip)
constexpr int KKT_TW_SPINS = 1 << 21;    // default bound of a wait (NewtonDev::kkt_tw_spins; cimpc_debug_set_tw_spins for the tests)
// Wait for the partner chain's flag to show THIS launch's stamp (agent scope), then acquire.  Bounded: a partner that never becomes
// resident must not hang the device - the caller poisons its result with NaN AND marks the rollout (kkt_tw_give_up), so that the
// solve is repeated on the one-ended kernel instead of being used.
__device__ __forceinline__ bool kkt_tw_wait(const int* flag, int epoch, int spins_max) {
    bool ok = false;
    for (int spins = 0; spins < spins_max; ++spins) {
        if (__builtin_amdgcn_readfirstlane(aload(flag)) == epoch) { ok = true; break; }
        __builtin_amdgcn_s_sleep(2);
    }
    __threadfence();
    return ok;
}
// a chain gave up waiting: word 3 of the rollout's flag line carries the launch's stamp (read by whichever chain finishes last)
__device__ __forceinline__ void kkt_tw_give_up(int* xfl, int epoch, bool one_lane) {
    if (one_lane) astore(xfl + 3, epoch);
}
// The last chain of a rollout whose hand-over timed out: the KKT stage of the rollout is queued AGAIN (it stays in STAGE_KKT; the
// entry joins the list the decision kernel of this round builds for the next one) and the failure is counted where the host sees it.
__device__ __forceinline__ void kkt_tw_requeue(const NewtonDev& S, int b, int finish) {
    if (finish == 1) {
        const int pos = atomicAdd(&S.counters[1 * CPAD], 1);
        if (S.kkt_list != nullptr && pos < S.dm.B) S.kkt_list[(size_t)S.WQ.par * S.dm.B + pos] = b;
    }
    if (S.kkt_tw_fail != nullptr) { __hip_atomic_fetch_add(S.kkt_tw_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __threadfence_system(); }
}
// tile leading dimension for a model: 16 (one MFMA block per tile) or 24 (2 x 2 blocks, masked)
template <int NQ, int NU>
constexpr int kkt_tld() { return (NQ <= 16 && NU <= 16) ? 16 : 24; }
constexpr int KKT_SRC_DOUBLES = 64;      // behind the vectors: the rollout's good_src row as 128 ints (H <= kkt_max_h <= 108), see NewtonDev::good_src
template <int NQ, int NU>
constexpr int kkt_tw_lds_doubles() {
    constexpr int T = kkt_tld<NQ, NU>();
    return KKT_TW_TILES * T * T + 13 * (T <= 16 ? 16 : 32) + KKT_SRC_DOUBLES;
}
// duo chains (kkt_body<..., PIPE = 1, TW>): physical tiles of a chain's slice - the 20 tiles a one-wave body needs once Y1 / Lc share
// the tiles of T0 / T1, plus (bottom chain) two more slots of the dq0 ring and Tq
__host__ __device__ constexpr int kkt_duo_tiles(int tw) { return tw == 2 ? 23 : 20; }
__host__ __device__ constexpr int kkt_duo_slice_doubles(int tw) { return kkt_duo_tiles(tw) * 256 + 13 * 16 + KKT_SRC_DOUBLES; }
// doubles of LDS one recursion needs
template <int NQ, int NU, int PIPE>
constexpr int kkt_lds_doubles() {
    constexpr int T = kkt_tld<NQ, NU>();
    return (PIPE == 3 ? KKT_PIPE3_TILES : PIPE == 2 ? KKT_PIPE_TILES : KKT_MFMA_TILES) * T * T + 13 * (T <= 16 ? 16 : 32) + KKT_SRC_DOUBLES;
}
// longest horizon the backward pass can stage: all dnu in tiles 16..21, the recovery's first level in tiles 7..15
template <int NQ, int NU>
constexpr int kkt_max_h() {
    constexpr int T = kkt_tld<NQ, NU>(), a = 6 * T * T / (T <= 16 ? 16 : 32), b = 9 * T * T / (NQ + NU);
    return a < b ? a : b;
}
// rollouts per workgroup of the packed launch: as many as fit 160 KB of LDS, at most KKT_PACK
template <int NQ, int NU>
constexpr int kkt_pack();

template <int NQ, int NU, class Sync, int PIPE = 1, bool F32 = false, int TW = 0>
__device__ __forceinline__ void kkt_body(const NewtonDev& S, const KktArgs& K, int b, double* sm, int lane, int wave = 0) {
    static_assert(NQ <= 24 && NU <= 24, "MFMA KKT kernel: tiles of at most 24 x 24");
    static_assert(TW == 0 || ((PIPE == 3 || PIPE == 1) && !F32), "the twisted chains are three-stage pipelines or one-wave bodies, in fp64");
    // DUO (round 6): the two-ended solve on TWO ONE-WAVE chains that are the two waves of one workgroup (kkt_kernel_duo) - the kernel
    // that runs next to the sweep, where three-wave chains would take the sweep's CUs.  Same recurrences, same hand-over through the
    // exchange block; the chains meet at a workgroup barrier at the end and start the line search together.
    constexpr bool DUO = TW != 0 && PIPE == 1;
    static_assert(!DUO || kkt_tld<NQ, NU>() == 16, "duo chains: 16-wide tiles (two slices must fit next to a sweep workgroup)");
    constexpr bool REV = TW == 2;                     // bottom chain: chain step i = row H-1-i
    constexpr int TL = kkt_tld<NQ, NU>();            // (shadow the 16-wide defaults of the file scope)
    constexpr int TSZ = TL * TL;
    constexpr int NB = (TL + 15) / 16;
    constexpr int VS = TL <= 16 ? 16 : 32;           // stride of the small vectors behind the tiles
    using Acc = TAcc<NB, F32>;
    constexpr int NTILES = DUO ? kkt_duo_tiles(TW) : TW != 0 ? KKT_TW_TILES : PIPE == 3 ? KKT_PIPE3_TILES : PIPE == 2 ? KKT_PIPE_TILES : KKT_MFMA_TILES;
    // The body runs on ONE wavefront; its phases hand data over through LDS only.  The hand-off needs the
    // wave's LDS operations complete (lgkmcnt(0)) - NOT its global ones: a full barrier (vmcnt(0)) would
    // expose the latency of the prefetch loads and of the factor spill stores at every phase boundary.
    auto lds_sync = [] {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };
    constexpr int nq = NQ, nu = NU, nd = NQ, nr = NQ + NU, nths = 2 * NQ + NU, n2 = nd * nd;
    constexpr int KBU = (NU + 3) / 4, KBQ = (NQ + 3) / 4, KBD = (nd + 3) / 4;
    const cimpc_dims& m = S.dm;
    const int H = m.H;
    // chain geometry: NS stage-A steps, of which the first NF are full steps (factor, y); row(i) = the matrix row of chain step i
    const int msp = TW != 0 ? kkt_tw_split(H, S.kkt_tw_nb, TL > 16, DUO) : 0;       // twisted: middle rows msp, msp + 1
    const int nbot = H - msp - 2;                                      // rows the bottom chain eliminates
    const int NS = TW == 1 ? msp + 2 : TW == 2 ? nbot + 2 : H;
    [[maybe_unused]] const int NF = TW == 2 ? nbot : NS;
    auto row = [&](int i) { return REV ? H - 1 - i : i; };
    const int li = lane & 15, lk = lane >> 4;
    // physical tile of logical tile t.  DUO packs a chain into 20 (top) / 23 (bottom) tiles so that both slices fit the LDS one
    // packed workgroup takes today: a one-wave body runs stage A and stage B in sequence, so Y1 and Lc (stage B only) share the
    // tiles of T0 and T1 (stage A only); the pipeline's hand-over tiles 22 .. 28 do not exist; the bottom chain's two extra dq0
    // ring slots and Tq follow the 20 common tiles
    auto tix = [](int t) { return !DUO ? t : (t == 7 || t == 8) ? t - 3 : (t >= 9 && t <= 21) ? t - 2 : t >= 29 ? t - 9 : t; };
    auto tile = [&](int t) { return sm + tix(t) * TSZ; };
    double* A0 = tile(0);                                  // du1_i            nd x nu
    double* A2 = tile(1);                                  // dq0_i            nd x nq   (bottom chain: ring of three, tiles 1, 29, 30)
    // tiles 2,3: dq1 ring (i, i-1)
    double* T0 = tile(4); double* T1 = tile(5); double* T2 = tile(6);
    double* Y1 = tile(7); double* Lc = tile(8); double* L2c = tile(9);
    // tiles 10..12: L0^-1 ring, 13..14: L1 ring, 15..17: Qinv ring
    double* Ri = tile(18);
    // tiles 19..21: extra ring slots of the backward pass
    double* vec = sm + NTILES * TSZ;
    [[maybe_unused]] double* bet = vec;                    // 16 each (PIPE = 2: second parity at vec + 144)
    double* tv = vec + VS;
    // vec + (2,3,4) VS: y / dnu ring
    double* rpu = vec + 5 * VS;
    // vec + (6,7,8) VS: r_p(q) ring; vec + 9 VS: second beta (PIPE = 2); vec + 12 VS: scratch word
#ifdef CIMPC_KKT_TWPROF
    // (diagnostic builds) constant-rate clock stamps of the twisted chains -> statistics words 8 + 8 (TW - 1) + j (wave 0), 24 .. 27 (the waits)
#define TWSTAMP(j) { if (TW != 0 && lane == 0 && b == 0) ((long long*)S.stats)[8 + 8 * (TW - 1) + (j)] = (long long)wall_clock64(); }
#define TWSTAMPW(j) { if (TW != 0 && lane == 0 && b == 0) ((long long*)S.stats)[24 + (j)] = (long long)wall_clock64(); }
#else
#define TWSTAMP(j) {}
#define TWSTAMPW(j) {}
#endif
    if (wave == 0) { TWSTAMP(0) }
#ifdef CIMPC_KKT_TWPROF
    if (TW != 0 && wave == 0 && lane == 0 && b == 0) {      // where the chain runs: HW_ID (wave / simd / cu / sh / se), XCC_ID
        ((long long*)S.stats)[28 + 2 * (TW - 1)] = (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        ((long long*)S.stats)[29 + 2 * (TW - 1)] = (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
#endif
    for (int k = lane + 64 * wave; k < NTILES * TSZ + 13 * VS; k += 64 * PIPE) sm[k] = 0.0;
    // Lazily committed sensitivities (NewtonDev::good_src): row j of the accepted evaluation sits in slot srcs[j] of the rollout's
    // evaluation slots (-1: in dz_good).  Every row passes through this kernel's registers once - it goes on to dz_good from there.
    const bool lazy = K.stage != nullptr && K.dz_override == nullptr && S.good_src != nullptr;
    int* const srcs = reinterpret_cast<int*>(sm + NTILES * TSZ + 13 * VS);
    if (lazy) for (int k = lane + 64 * wave; k < H; k += 64 * PIPE) srcs[k] = S.good_src[(size_t)b * H + k];
    if constexpr (PIPE >= 2) __syncthreads(); else lds_sync();
    if (wave == 0) { TWSTAMP(1) }
    const double* rb = K.r + (size_t)b * S.N;
    const double beta = K.beta ? K.beta[b] : K.beta_scalar;
    const double rho = (double)H * beta * S.kappa;         // newton_jacobian.jl:169-186 quirk
    // Newton loop: the accepted evaluation's sensitivities; B1 seam (no stage): slot 0 of the last implicit_dynamics!
    const double* dzb = kkt_dz(S, K, b, H, nths, nd);
    const double* const dz_slots = S.dz + (size_t)b * CS * H * nths * nd;
    auto dzrow = [&](int j) -> const double* {      // sensitivities of row j: the slot the accept named, else the resident block
        const int sl = lazy ? srcs[j] : -1;
        const size_t off_slot = ((size_t)(sl >= 0 ? sl : 0) * H + j) * (size_t)(nths * nd), off_good = (size_t)j * (nths * nd);
        return sl >= 0 ? dz_slots + off_slot : dzb + off_good;
    };
    double* ws = S.kkt_ws + (size_t)b * H * (3 * n2 + nd);
    constexpr int WSR = 3 * n2 + nd;
    auto rec = [&](int i) { return ws + (size_t)row(i) * WSR; };      // record of chain step i (the chains' rows are disjoint)
    // exchange block and flags of the two chains
    [[maybe_unused]] double* const xS = TW != 0 ? S.kkt_tw_xch + (size_t)b * kkt_tw_xch_doubles(nd) : nullptr;
    [[maybe_unused]] double* const xc = xS + 3 * n2;                  // c0, c1
    [[maybe_unused]] double* const xdn = xS + 3 * n2 + 2 * nd;        // dnu_{m+1}, dnu_m
    [[maybe_unused]] int* const xfl = TW != 0 ? S.kkt_tw_flags + (size_t)b * KKT_TW_FLAGS : nullptr;
    [[maybe_unused]] bool tw_ok = true;

    constexpr int PF_DZ = (nd * nths + 63) / 64, PF_Q = (nq * nq + 63) / 64, PF_R = (nu * nu + 63) / 64;
    double pf_dz[PF_DZ], pf_q[PF_Q], pf_r[PF_R], pf_rp = 0.0, pf_rd = 0.0;
    int pf_row = -1;                                      // >= 0: the prefetched block came from a slot and goes on to dz_good[row] at commit
    // Per-lane marshalling plan, computed ONCE and branch-free in the loop: which element of a step's
    // operand block this lane moves (k = lane + 64 j, clamped: surplus lanes re-load a valid element) and
    // where it lands in LDS (surplus lanes write a scratch word).  The divisions by nd / nq / nu, the tile
    // selection and ~35 exec-mask branches per step would otherwise be redone for every element of every
    // step (measured: 2 k of the 14 k cycles of a step).
    const int TRASH = NTILES * TSZ + 12 * VS;                // scratch double behind the vectors
    int dz_src[PF_DZ], dz_dst[PF_DZ], dz_rot[PF_DZ], q_src[PF_Q], q_dst[PF_Q], r_src[PF_R], r_dst[PF_R];
    [[maybe_unused]] int dz_rot2[PF_DZ];                     // bottom chain: 1 for the dq0 entries (ring of three tiles)
#pragma unroll
    for (int j = 0; j < PF_DZ; ++j) {
        const int k = lane + 64 * j, ok = k < nd * nths, kk = ok ? k : 0, r = kk % nd, c = kk / nd;
        // dq0 -> tile 1 (A2), dq1 -> tiles 2/3 (ring by step parity), du1 -> tile 0 (A0)
        const int base = (c < nq) ? 1 * TSZ + c * TL : (c < 2 * nq) ? 2 * TSZ + (c - nq) * TL : (c - 2 * nq) * TL;
        dz_src[j] = kk;
        dz_dst[j] = ok ? base + r : TRASH;
        dz_rot[j] = (ok && c >= nq && c < 2 * nq) ? TSZ : 0;
        dz_rot2[j] = (ok && c < nq) ? 1 : 0;
    }
#pragma unroll
    for (int j = 0; j < PF_Q; ++j) {
        const int k = lane + 64 * j, ok = k < nq * nq, kk = ok ? k : 0;
        q_src[j] = kk; q_dst[j] = ok ? (kk % nq) + (kk / nq) * TL : -1;      // (within the ring slot: tile tix(15) + slot)
    }
#pragma unroll
    for (int j = 0; j < PF_R; ++j) {
        const int k = lane + 64 * j, ok = k < nu * nu, kk = ok ? k : 0;
        r_src[j] = kk; r_dst[j] = ok ? tix(18) * TSZ + (kk % nu) + (kk / nu) * TL : TRASH;
    }
    const int rp_src = lane < nr ? lane : 0, rd_src = lane < nd ? lane : 0;
    // (bottom chain: the sensitivities, Rinv, r_p(u), r_d of row j = H-1-i, the weights Qinv and r_p(q) of row j-2 - they enter two
    //  rows ahead of the operands, the rows H-1, H-2 in the preamble)
    auto prefetch = [&](int i) {
        if (i < 0 || i >= NS) return;
        const int j = row(i), jq = REV ? max(j - 2, 0) : j;
        const double* dzi = dzrow(j);
        pf_row = (lazy && srcs[j] >= 0) ? j : -1;
#pragma unroll
        for (int j2 = 0; j2 < PF_DZ; ++j2) pf_dz[j2] = dzi[dz_src[j2]];
#pragma unroll
        for (int j2 = 0; j2 < PF_Q; ++j2) pf_q[j2] = S.Qinv[(size_t)jq * nq * nq + q_src[j2]];
#pragma unroll
        for (int j2 = 0; j2 < PF_R; ++j2) pf_r[j2] = S.Rinv[(size_t)j * nu * nu + r_src[j2]];
        pf_rp = rb[(REV && lane >= nu ? jq : j) * nr + rp_src];
        pf_rd = rb[H * nr + j * nd + rd_src];
    };
    // registers -> LDS tiles of step i (p0 = i & 1 selects the dq1 ring slot, m0 = i % 3 the dq0 slot of the bottom chain,
    // mq the Qinv / r_p(q) slots: i % 3, bottom chain (i + 2) % 3 = the slot of chain step i + 2)
    auto commit = [&](int p0, int m0, int mq) {
        const int a2off = REV ? (m0 == 0 ? 0 : (tix(28 + m0) - 1) * TSZ) : 0;      // tiles 1, 29, 30
#pragma unroll
        for (int j = 0; j < PF_DZ; ++j) sm[dz_dst[j] + p0 * dz_rot[j] + (REV ? dz_rot2[j] * a2off : 0)] = pf_dz[j];
        if (pf_row >= 0) {      // lazy commit: the block of this step, still in registers, goes on to dz_good (surplus lanes rewrite element 0)
            double* g = S.dz_good + ((size_t)b * H + pf_row) * (size_t)(nths * nd);
#pragma unroll
            for (int j = 0; j < PF_DZ; ++j) g[dz_src[j]] = pf_dz[j];
        }
#pragma unroll
        for (int j = 0; j < PF_Q; ++j) sm[q_dst[j] >= 0 ? (tix(15) + mq) * TSZ + q_dst[j] : TRASH] = pf_q[j];
#pragma unroll
        for (int j = 0; j < PF_R; ++j) sm[r_dst[j]] = pf_r[j];
        // r_p: [u | q2] -> rpu (vec + 5 VS), q ring (vec + (6 + mq) VS)
        const int vb = NTILES * TSZ;
        sm[lane < nu ? vb + 5 * VS + lane : lane < nr ? vb + (6 + mq) * VS + (lane - nu) : TRASH] = pf_rp;
    };
#ifdef CIMPC_KKT_PROF
    long long pt[16] = {0}; long long tp = clock64();
#define KPROF(j) { const long long tn = clock64(); pt[j] += tn - tp; tp = tn; }
#else
#define KPROF(j)
#endif
    // ring / parity slots of step i
    struct Slots { double *Li, *Li1, *Li2, *L1c, *L1p, *A1, *A1p, *Qi0, *Qi1, *Qi2, *yc, *y1, *y2, *q0r, *q1r, *q2r, *L2c, *Y0h, *Y1h, *bet, *LiT, *LiT1, *LiT2, *A2c, *A2p1, *A2p2; int p0, m0; };
    auto slots = [&](int i) {
        const int m0 = i % 3, m1 = (i + 2) % 3, m2 = (i + 1) % 3, p0 = i & 1, p1 = (i + 1) & 1;
        Slots t;
        t.Li = tile(10 + m0); t.Li1 = tile(10 + m1); t.Li2 = tile(10 + m2);
        t.L1c = tile(13 + p0); t.L1p = tile(13 + p1);
        t.A1 = tile(2 + p0); t.A1p = tile(2 + p1);
        t.Qi0 = tile(15 + m0); t.Qi1 = tile(15 + m1); t.Qi2 = tile(15 + m2);
        t.yc = vec + (2 + m0) * VS; t.y1 = vec + (2 + m1) * VS; t.y2 = vec + (2 + m2) * VS;
        t.q0r = vec + (6 + m0) * VS; t.q1r = vec + (6 + m1) * VS; t.q2r = vec + (6 + m2) * VS;
        // hand-over buffers of the pipelined variant (by step parity); the one-wave variant keeps one set
        // L2_i: one tile (PIPE 1), by step parity (PIPE 2), ring of three (PIPE 3: stage C reads it two ticks after stage A wrote it)
        t.L2c = tile(PIPE == 3 ? (m0 == 0 ? 9 : m0 == 1 ? 22 : 27) : (PIPE == 2 && p0) ? 22 : 9);      // (tile 9 = L2c; a select between tile INDICES, not pointers)
        t.Y0h = tile(23 + (PIPE >= 2 ? p0 : 0));
        t.Y1h = tile(25 + (PIPE >= 2 ? p0 : 0));
        // beta_i: by step parity (PIPE 2), ring of three (PIPE 3: read by stage C two ticks after stage A wrote it)
        // (slot chosen as an OFFSET from one base: a select between pointers trips the compiler bug noted at the primal recovery)
        t.bet = vec + (PIPE == 3 ? (m0 == 0 ? 0 : m0 == 1 ? 9 : 10) : (PIPE == 2 && p0) ? 9 : 0) * VS;
        // L0^-T ring (the transposed inverse factor: operand of W1 / W2): tiles 19..21 are free during the forward pass (PIPE 3:
        // four slots - stage C reads step i-2's while stage B writes step i+1's)
        auto lit = [&](int j) { constexpr int R = PIPE == 3 ? 4 : 3; const int q = ((j % R) + R) % R; return tile(q < 3 ? 19 + q : 28); };
        t.LiT = lit(i); t.LiT1 = lit(i - 1); t.LiT2 = lit(i - 2);
        // dq0 ring of the bottom chain (chain steps i, i-1, i-2 = rows j, j+1, j+2)
        auto a2t = [&](int q) { return tile(q == 0 ? 1 : 28 + q); };
        t.A2c = a2t(REV ? m0 : 0); t.A2p1 = a2t(m1); t.A2p2 = a2t(m2);
        t.p0 = p0; t.m0 = m0;
        return t;
    };
    const Acc z4 = tile_zero<NB, F32>();
    Acc y0 = z4, y1a = z4;                 // PIPE = 1: Y_ii / Y_i,i-1 accumulators stay in registers between the stages
    // PIPE 3: the forward substitution y_i does not feed the next step's factor - only the backward pass wants it - and leaves
    // the chain of stage B for stage C.  (Also tried: the down-date by L2_i moved into stage A, which knows L2_i - stage A then
    // becomes the longer stage (quadruped 139 -> 134 us only, hopper H = 20 loop 0.82 -> 0.94 ms) and the sum Y - L1 L1^T - L2 L2^T
    // is taken in another order than in the one- and two-wave variants, whose results must stay identical.)
    constexpr bool OFFCHAIN = PIPE == 3;
    auto put = [&](double* g, const Acc& a) {         // accumulator -> global, compact nd x nd column-major
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = 0; J < NB; ++J)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * I + li, col = tile_col<F32>(J, lk, r);
                    if (row < nd && col < nd) g[row + col * nd] = (double)a.v[I][J][r];
                }
    };
    [[maybe_unused]] auto sub_g = [&](Acc& a, const double* g, bool ok) {      // accumulator -= compact block in global memory (NaN: the partner never came)
#pragma unroll
        for (int I = 0; I < NB; ++I)
#pragma unroll
            for (int J = 0; J < NB; ++J)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * I + li, col = tile_col<F32>(J, lk, r);
                    if (row < nd && col < nd) a.v[I][J][r] -= ok ? g[row + col * nd] : __builtin_nan("");
                }
    };
    // ---- stage A of step i: needs factors of steps <= i-2 only --------------------------------------------
    auto stageA = [&](int i) {
        const Slots t = slots(i);
        double* const Li2 = t.Li2; double* const A1 = t.A1; double* const A1p = t.A1p;
        double* const Qi0 = t.Qi0; double* const Qi1 = t.Qi1; double* const Qi2 = t.Qi2;
        double* const q0r = t.q0r; double* const q1r = t.q1r; double* const q2r = t.q2r;
        double* const L2c = t.L2c; double* const bet = t.bet;
        const int p0 = t.p0, m0 = t.m0;
        // ---- P1: step i operands -> LDS, fetch step i+1 ------------------------------------
        commit(p0, m0, REV ? (i + 2) % 3 : m0);
        const double rd_i = pf_rd;
        lds_sync();
        prefetch(i + 1);
        KPROF(1)
        if constexpr (REV) {
            // Bottom chain, row j = H-1-i of the matrix = step i of the REVERSED matrix Y'_{a,b} = Y_{H-1-a,H-1-b}:
            //   Y'_ii     = Y_jj          = Qinv_j + rho I + T0 du1_j^T + T1 dq1_j^T + T2 dq0_j^T     (T0, T1, T2 as in the top chain)
            //   Y'_i,i-1  = Y_{j+1,j}^T   = -Qinv_j dq1_{j+1}^T + T1 dq0_{j+1}^T
            //   Y'_i,i-2  = Y_{j+2,j}^T   = -Qinv_j dq0_{j+2}^T  =: -Tq ,   L2'_i = -Tq L0'_{i-2}^-T
            // Ring slots follow the chain step: row j-1 = step i+1 -> slot (i+1) % 3 (Qi2 / q2r), row j-2 -> slot (i+2) % 3 (Qi1 / q1r);
            // dq1_{j+1} is the previous step's dq1 (A1p), dq0_{j+1}, dq0_{j+2} the dq0 ring's older slots.
            const int j = H - 1 - i;
            double* const Qj = t.Qi0; double* const Qj1 = t.Qi2; double* const Qj2 = t.Qi1;
            double* const rq0 = t.q0r; double* const rq1 = t.q2r; double* const rq2 = t.q1r;
            double* const A2j = t.A2c; double* const Tq = tile(31);
            tile_st<TL, F32>(T0, tile_mma<KBU, false, TL, F32>(A0, Ri, z4, li, lk), li, lk);
            if (j >= 1) tile_st<TL, F32>(T1, tile_mma<KBQ, false, TL, F32>(A1, Qj1, z4, li, lk), li, lk);
            if (j >= 2) tile_st<TL, F32>(T2, tile_mma<KBQ, false, TL, F32>(A2j, Qj2, z4, li, lk), li, lk);
            if (i >= 2) tile_st<TL, F32>(Tq, tile_mma<KBQ, false, TL, F32>(Qj, t.A2p2, z4, li, lk), li, lk);
            lds_sync();
            KPROF(2)
            y0 = tile_ld<TL, F32>(Qj, li, lk);
            tile_add_diag<NB, F32>(y0, rho, nd, li, lk);
            y0 = tile_mma<KBU, false, TL, F32>(T0, A0, y0, li, lk);
            if (j >= 1) y0 = tile_mma<KBQ, false, TL, F32>(T1, A1, y0, li, lk);
            if (j >= 2) y0 = tile_mma<KBQ, false, TL, F32>(T2, A2j, y0, li, lk);
            y1a = z4;
            if (i >= 1) {
                y1a = tile_mma<KBQ, true, TL, F32>(Qj, A1p, y1a, li, lk);
                if (j >= 1) y1a = tile_mma<KBQ, false, TL, F32>(T1, t.A2p1, y1a, li, lk);
            }
            if (i >= 2) tile_st<TL, F32>(L2c, tile_mma<KBD, true, TL, F32>(Tq, Li2, z4, li, lk), li, lk);
            if (lane < nd) {   // beta_j = T0 rpu_j - Qinv_j rq_j + T1 rq_{j-1} + T2 rq_{j-2} - rd_j
                double s = tile_mv<nu, false, TL>(T0, rpu, lane) - tile_mv<nq, false, TL>(Qj, rq0, lane);
                if (j >= 1) s += tile_mv<nq, false, TL>(T1, rq1, lane);
                if (j >= 2) s += tile_mv<nq, false, TL>(T2, rq2, lane);
                bet[lane] = s - rd_i;
            }
            if constexpr (PIPE >= 2) { tile_st<TL, F32>(t.Y0h, y0, li, lk); tile_st<TL, F32>(t.Y1h, y1a, li, lk); }
            KPROF(3)
            return;
        }
        // ---- P2: T0 = du1 Rinv, T1 = dq1 Qinv_{i-1}, T2 = dq0 Qinv_{i-2}  (Qinv, Rinv symmetric)
        tile_st<TL, F32>(T0, tile_mma<KBU, false, TL, F32>(A0, Ri, z4, li, lk), li, lk);
        if (i >= 1) tile_st<TL, F32>(T1, tile_mma<KBQ, false, TL, F32>(A1, Qi1, z4, li, lk), li, lk);
        if (i >= 2) tile_st<TL, F32>(T2, tile_mma<KBQ, false, TL, F32>(A2, Qi2, z4, li, lk), li, lk);
        lds_sync();
        KPROF(2)
        // ---- P3: Y_ii, Y_i,i-1, L2_i = -T2 L0_{i-2}^-T, beta_i -------------------------------
        y0 = tile_ld<TL, F32>(Qi0, li, lk);
        tile_add_diag<NB, F32>(y0, rho, nd, li, lk);
        y0 = tile_mma<KBU, false, TL, F32>(T0, A0, y0, li, lk);
        y1a = z4;
        if (i >= 1) {
            y0 = tile_mma<KBQ, false, TL, F32>(T1, A1, y0, li, lk);
            y1a = tile_neg<NB, F32>(tile_ld<TL, F32>(T1, li, lk));
        }
        if (i >= 2) {
            y0 = tile_mma<KBQ, false, TL, F32>(T2, A2, y0, li, lk);
            y1a = tile_mma<KBQ, false, TL, F32>(T2, A1p, y1a, li, lk);
            tile_st<TL, F32>(L2c, tile_mma<KBD, true, TL, F32>(T2, Li2, z4, li, lk), li, lk);
        }
        if (lane < nd) {   // beta_i = T0 rpu - Qinv_i rq_i + T1 rq_{i-1} + T2 rq_{i-2} - rd_i
            double s = tile_mv<nu, false, TL>(T0, rpu, lane) - tile_mv<nq, false, TL>(Qi0, q0r, lane);
            if (i >= 1) s += tile_mv<nq, false, TL>(T1, q1r, lane);
            if (i >= 2) s += tile_mv<nq, false, TL>(T2, q2r, lane);
            bet[lane] = s - rd_i;
        }
        if constexpr (PIPE >= 2) { tile_st<TL, F32>(t.Y0h, y0, li, lk); tile_st<TL, F32>(t.Y1h, y1a, li, lk); }
        KPROF(3)
    };
    // ---- stage C of step i: what the backward pass runs on, formed off the dependency chain and spilled ------------
    //   dnu_j = L0_j^-T (y_j - L1_{j+1}^T dnu_{j+1} - L2_{j+2}^T dnu_{j+2})  =  yhat_j - W1_j^T dnu_{j+1} - W2_j^T dnu_{j+2}
    //   W1_j = L1_{j+1} L0_j^-1 ,  W2_j = L2_{j+2} L0_j^-1 ,  yhat_j = L0_j^-T y_j
    // so that a backward step is ONE level of two independent mat-vecs instead of two dependent levels.  Step i knows L1_i and
    // L2_i: it completes the records of steps i-1 (W1) and i-2 (W2) and writes yhat_i.  Record of a step: [W1 | W2 | - | yhat].
    auto stageC = [&](int i) {
        const Slots t = slots(i);
        // (bottom chain, second trace step i = nbot + 1: there is no L1 of that step)
        if (i >= 1 && !(TW == 2 && i == nbot + 1)) put(rec(i - 1), tile_mma<KBD, false, TL, F32>(t.L1c, t.LiT1, z4, li, lk));
        if (i >= 2) put(rec(i - 2) + n2, tile_mma<KBD, false, TL, F32>(t.L2c, t.LiT2, z4, li, lk));
        if constexpr (TW == 2) {
            if (i >= nbot) {      // trace steps: what the eliminated rows contribute to the right-hand side of rows m+1 (c0), m (c1)
                if (lane < nd) {
                    double s = tile_mv<nd, false, TL>(t.L2c, t.y2, lane);           // L2'_i y'_{i-2}
                    if (i == nbot) s += tile_mv<nd, false, TL>(t.L1c, t.y1, lane);  // + L1'_nb y'_{nb-1}
                    xc[(i - nbot) * nd + lane] = s;
                }
                if (i == nbot + 1) {      // everything the top chain waits for is written: release, raise the flag
                    __threadfence();
                    if (lane == 0) astore(xfl + 0, S.kkt_tw_epoch);
                }
                return;
            }
        }
        if constexpr (OFFCHAIN) {       // forward substitution of step i: y_i = L0_i^-1 (beta_i - L1_i y_{i-1} - L2_i y_{i-2})
            [[maybe_unused]] double cm = 0.0;
            if constexpr (TW == 1) {
                if (i >= msp) {       // middle rows: the bottom chain's share of the right-hand side (c1 for row m, c0 for row m+1)
                    if (i == msp) { tw_ok = kkt_tw_wait(xfl + 0, S.kkt_tw_epoch, S.kkt_tw_spins); if (!tw_ok) kkt_tw_give_up(xfl, S.kkt_tw_epoch, lane == 0); }
                    if (lane < nd) cm = tw_ok ? xc[(msp + 1 - i) * nd + lane] : __builtin_nan("");
                }
            }
            if (lane < nd) {
                double s = t.bet[lane];
                if constexpr (TW == 1) s -= cm;
                if (i >= 1) s -= tile_mv<nd, false, TL>(t.L1c, t.y1, lane);
                if (i >= 2) s -= tile_mv<nd, false, TL>(t.L2c, t.y2, lane);
                tv[lane] = s;
            }
            lds_sync();
            if (lane < nd) t.yc[lane] = tile_mv<nd, false, TL>(t.Li, tv, lane);
            lds_sync();
        }
        if (lane < nd) rec(i)[3 * n2 + lane] = tile_mv<nd, true, TL>(t.Li, t.yc, lane);
    };
    // ---- stage B of step i: L1_i, the Cholesky factor L0_i and its inverse, y_i (the recursion's dependency chain) ---
    auto stageB = [&](int i) {
        const Slots t = slots(i);
        double* const Li = t.Li; double* const Li1 = t.Li1; double* const L1c = t.L1c; double* const L1p = t.L1p;
        [[maybe_unused]] double* const yc = t.yc; [[maybe_unused]] double* const y1 = t.y1; [[maybe_unused]] double* const y2 = t.y2;
        double* const L2c = t.L2c; [[maybe_unused]] double* const bet = t.bet;
        if constexpr (PIPE >= 2) { y0 = tile_ld<TL, F32>(t.Y0h, li, lk); y1a = tile_ld<TL, F32>(t.Y1h, li, lk); }
        if constexpr (TW == 1) {
            if (i >= msp) {       // middle rows: minus what the rows eliminated from the bottom contribute
                if (i == msp) { TWSTAMPW(0) tw_ok = kkt_tw_wait(xfl + 0, S.kkt_tw_epoch, S.kkt_tw_spins); TWSTAMPW(1) if (!tw_ok) kkt_tw_give_up(xfl, S.kkt_tw_epoch, lane == 0); }
                sub_g(y0, xS + (i == msp ? n2 : 0), tw_ok);                 // Y_mm -= S11 ,  Y_{m+1,m+1} -= S00
                if (i == msp + 1) sub_g(y1a, xS + 2 * n2, tw_ok);           // Y_{m+1,m} -= S10^T
            }
        }
        if constexpr (TW == 2) {
            if (i == nbot + 1) {  // trace step of row m: S11 = L2 L2^T, S10^T = L1'_nb L2^T
                put(xS + n2, tile_mma<KBD, false, TL, F32>(L2c, L2c, z4, li, lk));
                put(xS + 2 * n2, tile_mma<KBD, false, TL, F32>(L1p, L2c, z4, li, lk));
                __threadfence();
                if constexpr (PIPE != 3) stageC(i);      // (one-wave chain: c1 and the flag are this wave's too)
                return;
            }
        }
        // ---- P4: Y1 -= L2 L1_{i-1}^T ; stage Y1 as an operand ------------------------------
        if (i >= 1) {
            if (i >= 2) y1a = tile_mma<KBD, true, TL, F32>(L2c, L1p, y1a, li, lk);
            tile_st<TL, F32>(Y1, y1a, li, lk);
            lds_sync();
            // ---- P5: L1_i = Y1 L0_{i-1}^-T -----------------------------------------------------
            tile_st<TL, F32>(L1c, tile_mma<KBD, false, TL, F32>(Y1, Li1, z4, li, lk), li, lk);
            lds_sync();
        }
        KPROF(4)
        if constexpr (TW == 2) {
            if (i == nbot) {      // trace step of row m+1: S00 = L1 L1^T + L2 L2^T
                put(xS, tile_mma<KBD, false, TL, F32>(L2c, L2c, tile_mma<KBD, false, TL, F32>(L1c, L1c, z4, li, lk), li, lk));
                __threadfence();
                if constexpr (PIPE != 3) stageC(i);
                return;
            }
        }
        // ---- P6: Lc = Y0 - L1 L1^T - L2 L2^T ; rhs of the forward substitution ---------------
        if (i >= 1) y0 = tile_mma<KBD, true, TL, F32>(L1c, L1c, y0, li, lk);
        if (i >= 2) y0 = tile_mma<KBD, true, TL, F32>(L2c, L2c, y0, li, lk);
        tile_st<TL, F32>(Lc, y0, li, lk);
        if constexpr (!OFFCHAIN) {
            if (lane < nd) {
                double s = bet[lane];
                if constexpr (TW == 1) { if (i >= msp) s -= tw_ok ? xc[(msp + 1 - i) * nd + lane] : __builtin_nan(""); }      // c1 (row m), c0 (row m + 1)
                if (i >= 1) s -= tile_mv<nd, false, TL>(L1c, y1, lane);
                if (i >= 2) s -= tile_mv<nd, false, TL>(L2c, y2, lane);
                tv[lane] = s;
            }
        }
        lds_sync();
        KPROF(5)
        // ---- P7: Cholesky of Lc and L0^-1 in registers (lane = row, DPP broadcasts) -----------
        {
            using LG = KktBcast<TL>;
            const int rl = TL <= 16 ? li : lane;
            double a[nd], invd[nd];
            static_for<0, nd>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                a[c] = (rl < nd) ? Lc[rl + c * TL] : ((c == rl) ? 1.0 : 0.0);
            });
            static_for<0, nd>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const double dk = LG::template bcast<k>(a[k]);
                const double inv = fast_rsqrt(dk);
                invd[k] = inv;
                a[k] *= inv;
                static_for<k + 1, nd>([&](auto cc) {
                    constexpr int c = decltype(cc)::value;
                    a[c] = fma(-a[k], LG::template bcast<c>(a[k]), a[c]);
                });
            });
            double xc[nd];
            static_for<0, nd>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                double acc = 0.0;
                static_for<0, r>([&](auto kc) {
                    constexpr int k = decltype(kc)::value;
                    acc = fma(LG::template bcast<r>(a[k]), xc[k], acc);
                });
                xc[r] = (((r == rl) ? 1.0 : 0.0) - acc) * invd[r];
            });
            if (lane < nd) {
                static_for<0, nd>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    Li[r + lane * TL] = xc[r];
                    t.LiT[lane + r * TL] = xc[r];       // L0^-T: consecutive lanes on consecutive addresses
                });
            }
        }
        lds_sync();
        KPROF(6)
        // ---- P8: y_i = L0^-1 tv -------------------------------------------------------------------
        if constexpr (!OFFCHAIN) {
            if (lane < nd) yc[lane] = tile_mv<nd, false, TL>(Li, tv, lane);
            lds_sync();
        }
        if constexpr (PIPE != 3) stageC(i);
        KPROF(7)
    };
    if constexpr (PIPE >= 2) {
        if constexpr (REV) {
            if (wave == 0) {      // preamble of the bottom chain: weights and r_p(q) of rows H-1, H-2 -> ring slots 0, 1 (chain steps 0, 1)
                for (int k = lane; k < 2 * nq * nq; k += 64) {
                    const int q = k / (nq * nq), e = k - q * nq * nq;
                    sm[(tix(15) + q) * TSZ + (e % nq) + (e / nq) * TL] = S.Qinv[(size_t)(H - 1 - q) * nq * nq + e];
                }
                for (int k = lane; k < 2 * nq; k += 64) {
                    const int q = k / nq, e = k - q * nq;
                    vec[(6 + q) * VS + e] = rb[(H - 1 - q) * nr + nu + e];
                }
            }
        }
        if (wave == 0) prefetch(0);
#ifdef CIMPC_KKT_WPROF
        long long w_stage = 0, w_bar = 0;
#endif
        for (int tck = 0; tck < NS + PIPE - 1; ++tck) {      // tick: A(tck) on wave 0, B(tck - 1) on wave 1, (PIPE 3) C(tck - 2) on wave 2
#ifdef CIMPC_KKT_WPROF
            const long long w0 = clock64();
#endif
            if (wave == 0 && tck < NS) stageA(tck);
            if (wave == 1 && tck >= 1 && tck <= NS) stageB(tck - 1);
            if constexpr (PIPE == 3) { if (wave == 2 && tck >= 2) stageC(tck - 2); }
#ifdef CIMPC_KKT_WPROF
            const long long w1 = clock64();
#endif
            __syncthreads();
#ifdef CIMPC_KKT_WPROF
            w_stage += w1 - w0; w_bar += clock64() - w1;
#endif
        }
#ifdef CIMPC_KKT_WPROF
        // (diagnostic builds) per wave: time inside its stage, time at the tick barrier -> statistics words 17 + 2 wave, 18 + 2 wave
        if (lane == 0 && b == 0) { ((long long*)S.stats)[(TW != 0 ? 32 + 8 * (TW - 1) : 8 + 9) + 2 * wave] = w_stage; ((long long*)S.stats)[(TW != 0 ? 33 + 8 * (TW - 1) : 8 + 10) + 2 * wave] = w_bar; }
#endif
        __threadfence_block();                    // the other waves' spill stores are read back by wave 0
        __syncthreads();
        // (twisted chains: the other two waves come back for the primal recovery - a third of the rows each)
        if constexpr (TW == 0) { if (wave != 0) return; }
        if (wave == 0) { TWSTAMP(2) }
    } else {
        if constexpr (REV) {      // preamble of the bottom chain (as above): weights and r_p(q) of rows H-1, H-2 -> ring slots 0, 1
            for (int k = lane; k < 2 * nq * nq; k += 64) {
                const int q = k / (nq * nq), e = k - q * nq * nq;
                sm[(tix(15) + q) * TSZ + (e % nq) + (e / nq) * TL] = S.Qinv[(size_t)(H - 1 - q) * nq * nq + e];
            }
            for (int k = lane; k < 2 * nq; k += 64) {
                const int q = k / nq, e = k - q * nq;
                vec[(6 + q) * VS + e] = rb[(H - 1 - q) * nr + nu + e];
            }
            lds_sync();
        }
        prefetch(0);
        for (int i = 0; i < NS; ++i) {
            stageA(i);
            lds_sync();
            stageB(i);
        }
    }
    __threadfence_block();
    lds_sync();
    // =========================================================================================
    // backward substitution, step i = H-1 .. 0 (sequential):
    //   dnu_i = L0_i^-T (y_i - L1_{i+1}^T dnu_{i+1} - L2_{i+2}^T dnu_{i+2}) = yhat_i - W1_i^T dnu_{i+1} - W2_i^T dnu_{i+2}   (stage C)
    // then the primal recovery for ALL steps at once (no dependency between steps):
    //   Du_i  = Rinv_i (rpu_i - du1_i^T dnu_i)
    //   Dq_i  = Qinv_i (rpq_i + dnu_i - dq1_{i+1}^T dnu_{i+1} - dq0_{i+2}^T dnu_{i+2})
    // =========================================================================================
    double* D = K.delta + (size_t)b * S.N;
    // per step: W1_i -> tile 0, W2_i -> tile 3, yhat_i -> vec; dnu_i = yhat_i - W1_i^T dnu_{i+1} - W2_i^T dnu_{i+2}
    double* W1t = tile(0); double* W2t = tile(3);
    double* yb = vec;                                 // yhat_i
    double* dn_all = tile(16);                        // [H][16] all dnu (tiles 16..18; H <= 48)
    double* t_all = tile(7);                          // [H][nr] first level of the recovery (tiles 7..)
    constexpr int RW = (TW != 0 && !DUO) ? 3 : 1;        // waves that share the primal recovery
    [[maybe_unused]] auto rsync = [&] { if constexpr (TW != 0 && !DUO) __syncthreads(); else lds_sync(); };
    if (TW == 0 || wave == 0) {                          // ---- the backward pass runs on wave 0 ----
    constexpr int PF_W = (2 * n2 + nd + 63) / 64;
    // The records come from global memory (L2): a load takes ~1 us, a backward step ~0.3 us - the loads run THREE steps ahead
    // (three register sets, the loop unrolled by three so that the sets are indexed statically; a record is 4 doubles per lane).
    constexpr int DEPTH = 3;
    double pf_w[DEPTH][PF_W];
    // marshalling plan of the backward pass: source element of the step's record [W1 | W2 | - | yhat], LDS destination
    int bw_src[PF_W], bw_dst[PF_W];
#pragma unroll
    for (int j = 0; j < PF_W; ++j) {
        const int k = lane + 64 * j, ok = k < 2 * n2 + nd, kk = ok ? k : 0;
        const int t = kk / n2, e = kk - t * n2, r = e % nd, c = e / nd;
        bw_src[j] = t < 2 ? kk : 3 * n2 + (kk - 2 * n2);
        bw_dst[j] = !ok ? TRASH : (t == 0) ? r + c * TL : (t == 1) ? 3 * TSZ + r + c * TL : NTILES * TSZ + (kk - 2 * n2);      // (tiles 0, 3: the same in every packing)
    }
    // chain-local indices: the chain substitutes its steps NBK-1 .. 0 and knows dnu of its steps 0 .. NS-1 afterwards (the bottom
    // chain receives the two middle rows - its steps nbot, nbot + 1 - from the top chain)
    const int NBK = TW == 2 ? nbot : NS;
    if constexpr (TW == 2) {
        tw_ok = kkt_tw_wait(xfl + 1, S.kkt_tw_epoch, S.kkt_tw_spins);
        if (!tw_ok) kkt_tw_give_up(xfl, S.kkt_tw_epoch, lane == 0);
        TWSTAMP(3)
        if (lane < 2 * nd) dn_all[(nbot + lane / nd) * VS + lane % nd] = tw_ok ? xdn[lane] : __builtin_nan("");
        lds_sync();
    }      // (epoch-valued flags: nothing is reset - the next launch waits for ITS stamp)
    auto prefetch_b = [&](auto setc, int i) {
        constexpr int set = decltype(setc)::value;
        if (i < 0) return;
        const double* wsi = rec(i);
#pragma unroll
        for (int j = 0; j < PF_W; ++j) pf_w[set][j] = wsi[bw_src[j]];       // (W1 of the last step, W2 of the last two: never written, never used)
    };
    auto back_step = [&](auto setc, int i) {
        constexpr int set = decltype(setc)::value;
        if (i < 0) return;
#pragma unroll
        for (int j = 0; j < PF_W; ++j) sm[bw_dst[j]] = pf_w[set][j];
        lds_sync();
        prefetch_b(setc, i - DEPTH);
        if (lane < nd) {
            double s = yb[lane];
            if (i + 1 < NS) s -= tile_mv<nd, true, TL>(W1t, dn_all + (i + 1) * VS, lane);
            if (i + 2 < NS) s -= tile_mv<nd, true, TL>(W2t, dn_all + (i + 2) * VS, lane);
            dn_all[i * VS + lane] = s;
            D[H * nr + row(i) * nd + lane] = s;
            if constexpr (TW == 1) { if (i >= msp) xdn[(msp + 1 - i) * nd + lane] = s; }      // the middle rows go to the bottom chain
        }
        if constexpr (TW == 1) {
            if (i == msp) {
                __threadfence();
                if (lane == 0) astore(xfl + 1, S.kkt_tw_epoch);
            }
        }
        lds_sync();
    };
    static_for<0, DEPTH>([&](auto kc) { prefetch_b(kc, NBK - 1 - decltype(kc)::value); });
    for (int i = NBK - 1; i >= 0; i -= DEPTH) {
        static_for<0, DEPTH>([&](auto kc) { back_step(kc, i - decltype(kc)::value); });
    }
    TWSTAMP(4)
    }                                                    // ---- end of the backward pass ----
    rsync();
    // ---- primal recovery, level 1: t = r_p - C^T dnu, one output per lane and trip.  In the q rows both sensitivity columns are
    //      requested before the first multiply-add (indices clamped, the terms dropped afterwards): one memory round trip per row
    //      instead of three dependent ones.  (Two loops, one per kind of row, trip an instruction-selection bug of this compiler
    //      in the translation unit where the tiles are reached through a generic pointer - the single loop stays.) ------------------
    // (twisted: the top chain recovers the primal rows 0 .. m-1 - they need dnu up to row m+1 -, the bottom chain rows m .. H-1;
    //  dnu of matrix row j sits at the chain-local index: dnl(j))
    const int rr0 = TW == 2 ? msp : 0, rrn = TW == 1 ? msp : TW == 2 ? H - msp : H;      // first row, number of rows
    auto dnl = [&](int j) { return dn_all + (REV ? H - 1 - j : j) * VS; };
    for (int idx = lane + 64 * wave; idx < rrn * nr; idx += 64 * RW) {
        const int il = idx / nr, c = idx - il * nr, i = rr0 + il;
        double s = rb[rr0 * nr + idx];
        if (c < nu) {
            const double* a0 = dzrow(i) + (size_t)(2 * nq + c) * nd;
            const double* dn0 = dnl(i);
            double t0 = 0.0;
#pragma unroll
            for (int k = 0; k < nd; ++k) t0 = fma(a0[k], dn0[k], t0);
            s -= t0;
        } else {
            const int cq = c - nu;
            const int i1 = min(i + 1, H - 1), i2 = min(i + 2, H - 1);
            const double* a1 = dzrow(i1) + (size_t)(nq + cq) * nd;
            const double* a2 = dzrow(i2) + (size_t)cq * nd;
            double v1[nd], v2[nd];
#pragma unroll
            for (int k = 0; k < nd; ++k) { v1[k] = a1[k]; v2[k] = a2[k]; }
            const double* dn1 = dnl(i1);
            const double* dn2 = dnl(i2);
            double t1 = 0.0, t2 = 0.0;
#pragma unroll
            for (int k = 0; k < nd; ++k) { t1 = fma(v1[k], dn1[k], t1); t2 = fma(v2[k], dn2[k], t2); }
            s += dnl(i)[cq];
            if (i + 1 < H) s -= t1;
            if (i + 2 < H) s -= t2;
        }
        t_all[idx] = s;
    }
    rsync();
    // ---- level 2: Delta_x = P^-1 t: a u row and a q row per trip, indices clamped instead of branched on --------------------
    constexpr int RJ = nq > nu ? nq : nu;
    for (int j = lane + 64 * wave; j < rrn * RJ; j += 64 * RW) {
        const bool on_u = j < rrn * nu, on_q = j < rrn * nq;
        const int ju = on_u ? j : 0, iul = ju / nu, cu = ju - iul * nu, iu = rr0 + iul;
        const int jq = on_q ? j : 0, il = jq / nq, cq = jq - il * nq, i = rr0 + il;
        const double* Rm = S.Rinv + (size_t)iu * nu * nu + cu;
        const double* Qm = S.Qinv + (size_t)i * nq * nq + cq;
        double rv[nu], qv[nq];
#pragma unroll
        for (int k = 0; k < nu; ++k) rv[k] = Rm[k * nu];
#pragma unroll
        for (int k = 0; k < nq; ++k) qv[k] = Qm[k * nq];
        double su = 0.0, sq = 0.0;
#pragma unroll
        for (int k = 0; k < nu; ++k) su = fma(rv[k], t_all[iul * nr + k], su);
#pragma unroll
        for (int k = 0; k < nq; ++k) sq = fma(qv[k], t_all[il * nr + nu + k], sq);
        D[iu * nr + cu] = su;              // (surplus lanes recompute and rewrite row 0: same value, same address)
        D[i * nr + nu + cq] = sq;
    }
    // committed: the index goes back to "in dz_good".  Twisted chains reset the rows only THEY read (top: 0 .. m-1, bottom: m+2 .. H-1);
    // the two middle rows, which both chains stream, are reset by whichever finishes last - a chain that starts late must still
    // find the index of every row it reads as the accept left it, or as reset AFTER the blocks reached memory (the finish fence)
    if (lazy) for (int idx = lane + 64 * wave; idx < (TW != 0 ? NS - 2 : NS); idx += 64 * RW) S.good_src[(size_t)b * H + row(idx)] = -1;
    if constexpr (DUO) {
        // both chains of the rollout are the two waves of this workgroup: they meet here, and the step is complete in front of both
        __threadfence_block();
        __syncthreads();
        if (lazy && TW == 1 && lane < 2) S.good_src[(size_t)b * H + msp + lane] = -1;      // the middle rows (both chains streamed them)
        if (__builtin_amdgcn_readfirstlane(aload(xfl + 3)) == S.kkt_tw_epoch) {               // a hand-over timed out (forced in the tests): poisoned numbers
            if (TW == 1 && lane == 0) kkt_tw_requeue(S, b, K.finish);
            return;
        }
        if (K.finish) start_line_search<BlockSync>(S, b, K.finish, lane + 64 * (TW - 1), 128);
        return;
    } else if constexpr (TW != 0) {      // the three waves' rows are complete (and visible) before wave 0 reports the chain finished
        __threadfence_block();
        __syncthreads();
    } else lds_sync();
    KPROF(8)
#ifdef CIMPC_KKT_PROF
    if (lane == 0 && b == 0) for (int j = 0; j < 9 ; ++j) ((long long*)S.stats)[8 + j] = pt[j];      // (diagnostic builds: overwrites the statistics of rollouts 2..5)
#endif
    if constexpr (TW != 0) {
        // The chain that finishes last has the whole step in front of it: it starts the line search - with all three of its waves
        // (after the acquire every line of the step, the trajectory and the reference comes from memory: one wave took 33 us for it,
        // profiles/r05/twisted_prof_e.log).  The verdict of the counter reaches the other waves through an LDS word.
        if (wave == 0) {
            TWSTAMP(5)
            __threadfence();
            if (lane == 0) vec[11 * VS] = (double)atomicAdd(xfl + 2, 1);
            TWSTAMP(6)
        }
        __syncthreads();
        if ((int)vec[11 * VS] == 0) return;
        __threadfence();
        if (wave == 0 && lane == 0) astore(xfl + 2, 0);
        if (lazy && wave == 0 && lane < 2) S.good_src[(size_t)b * H + msp + lane] = -1;      // the middle rows (see above)
        if (__builtin_amdgcn_readfirstlane(aload(xfl + 3)) == S.kkt_tw_epoch) {      // a hand-over of this solve timed out: the numbers are poisoned
            if (wave == 0 && lane == 0) kkt_tw_requeue(S, b, K.finish);
            return;
        }
        if (K.finish) start_line_search<BlockSync>(S, b, K.finish, lane + 64 * wave, 192);
        if (wave == 0) { TWSTAMP(7) }
        return;
    }
    if (K.finish) {
        __threadfence_block();
        start_line_search<Sync>(S, b, K.finish, lane, 64);
    }
    TWSTAMP(7)
}


// Packed launch of the Newton loop: KKT_PACK rollouts per workgroup (one wavefront each, private LDS slice),
// taken from the compact list the residual kernel built.  A lone KKT wave on a CU evicts one of the two
// sweep workgroups the CU could hold (it needs a 256-VGPR slot on one SIMD, the sweep workgroup one on
// each).  Two per workgroup (94 KB of LDS) leave room for exactly one sweep workgroup next to them, so
// half as many CUs lose a sweep workgroup (measured B = 512: pack 1 / 2 / 3 -> 14.0 / 13.3 / 13.6 ms).
#define CIMPC_KKT_PACK 2      // (a constant of the build: the -D override is gone with the experiment it served)
#define CIMPC_KKT_PACK_WAVES_PER_SIMD 2      // (a constant of the build: the -D override is gone with the experiment it served)
constexpr int KKT_PACK = CIMPC_KKT_PACK;
template <int NQ, int NU>
constexpr int kkt_pack() {
    constexpr int fit = (160 * 1024 / 8) / kkt_lds_doubles<NQ, NU, 1>();
    return fit < 1 ? 1 : (fit < KKT_PACK ? fit : KKT_PACK);
}
template <int NQ, int NU>
__global__ __launch_bounds__((64 * kkt_pack<NQ, NU>()), (kkt_tld<NQ, NU>() <= 16 ? CIMPC_KKT_PACK_WAVES_PER_SIMD : 1)) void kkt_kernel_packed(NewtonDev S, KktArgs K, const int* list, int n, const int* n_dev) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (n_dev != nullptr) n = *n_dev;        // rounds enqueued ahead of the host: the count is only known on the device
    const int wave = (int)threadIdx.x >> 6, slot = (int)blockIdx.x * kkt_pack<NQ, NU>() + wave;
    if (slot >= n) return;
    __builtin_amdgcn_s_setprio(CIMPC_KKT_PRIO);      // (see ip_kernel_impl.h: CIMPC_SWEEP_PRIO)
    kkt_body<NQ, NU, WaveSync>(S, K, list[slot], sm + (size_t)wave * kkt_lds_doubles<NQ, NU, 1>(), (int)threadIdx.x & 63);
}

// Pipelined launch: one rollout per workgroup of three wavefronts (kkt_body<..., PIPE = 3>), from the compact list.
template <int NQ, int NU>
__global__ __launch_bounds__(192, (kkt_tld<NQ, NU>() <= 16 ? 2 : 1)) void kkt_kernel_pipe(NewtonDev S, KktArgs K, const int* list, int n, const int* n_dev) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if (n_dev != nullptr) n = *n_dev;
    if ((int)blockIdx.x >= n) return;
    kkt_body<NQ, NU, WaveSync, 3>(S, K, list[blockIdx.x], sm, (int)threadIdx.x & 63, (int)threadIdx.x >> 6);
}

// Twisted launch (round 5): TWO workgroups of three wavefronts per rollout - block 2k runs the bottom chain of rollout list[k]
// (kkt_body<..., TW = 2>), block 2k + 1 the top chain (TW = 1).  The chains exchange the traces of the two middle rows and
// their dnu through global memory (flags at agent scope).  The bottom chain has the LOWER block index: it waits for nothing
// until its forward pass is done, so whatever the dispatch order, a resident bottom chain's partner is the next block to start.
// list == nullptr: rollout blockIdx.x / 2 + S.b0 (stage filter as kkt_kernel).
// The two chains are compiled as REAL functions (noinline), each with its own register allocation: inlined side by side into one
// kernel both bodies ran 15 - 30 % slower per step than the same stages of the one-ended kernel (492 v_readlane reloads of spilled
// scalar registers against 81; stage A 2.43 us per step against 1.88 - profiles/r05/twisted_prof_b.log).  A chain reads its
// arguments from the kernel-argument segment (scalar loads from the constant address space) and names the dynamic LDS itself,
// so that the tiles keep their address space (ds_ instructions, not flat ones): the LDS base travels as an address_space(3) pointer.
struct KktTwArgs { NewtonDev S; KktArgs K; const int* list; int n; const int* n_dev; };
template <class T, size_t OFFSET>
__device__ __forceinline__ T kkt_kernarg(unsigned long long v) {
    T out;
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    const unsigned long long u = (((unsigned long long)hi << 32) | lo) + OFFSET;
    __builtin_memcpy(&out, (const __attribute__((address_space(4))) T*)u, sizeof(T));
#else
    (void)v;
#endif
    return out;
}
using lds_double_ptr = __attribute__((address_space(3))) double*;
// (noreturn + s_endpgm inside: the kernel has nothing left to do after its chain, and a function that never returns need not
//  save the caller's registers - with the saves the kernel needed 416 bytes of scratch per lane, and a dispatch that needs scratch
//  of this size per wave slot pays for its allocation: kernel 98 us against 68 us between the chains' first and last clock stamp)
template <int NQ, int NU, int TW>
static __device__ __attribute__((noinline, noreturn)) void kkt_tw_chain(unsigned long long ka, int b, lds_double_ptr sm3) {
    const NewtonDev S = kkt_kernarg<NewtonDev, offsetof(KktTwArgs, S)>(ka);
    const KktArgs K = kkt_kernarg<KktArgs, offsetof(KktTwArgs, K)>(ka);
    kkt_body<NQ, NU, WaveSync, 3, false, TW>(S, K, b, (double*)sm3, (int)threadIdx.x & 63, (int)threadIdx.x >> 6);
    __builtin_amdgcn_endpgm();
}
template <int NQ, int NU>
__global__ __launch_bounds__(192, (kkt_tld<NQ, NU>() <= 16 ? 2 : 1)) void kkt_kernel_twisted(KktTwArgs A) {
#ifdef CIMPC_KKT_TWPROF
    if (threadIdx.x == 0 && blockIdx.x < 2) ((long long*)A.S.stats)[48 + blockIdx.x] = (long long)wall_clock64();      // kernel entry of the two chains of rollout 0
#endif
    int n = A.n;
    if (A.n_dev != nullptr) n = *A.n_dev;
    const int k = (int)blockIdx.x >> 1;
    if (k >= n) return;
    const int b = A.list != nullptr ? A.list[k] : k + A.S.b0;
    if (A.list == nullptr) {
        if (A.K.stage != nullptr && A.K.stage[b] != STAGE_KKT) return;
        if (A.K.only_flag != nullptr && A.K.only_flag[b] == 0) return;
    }
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    if (((int)blockIdx.x & 1) == 0) kkt_tw_chain<NQ, NU, 2>(ka, b, (lds_double_ptr)sm);
    else kkt_tw_chain<NQ, NU, 1>(ka, b, (lds_double_ptr)sm);
}

// Duo launch (round 6): ONE workgroup of two wavefronts per rollout - wave 0 runs the bottom chain (kkt_body<..., PIPE = 1, TW = 2>),
// wave 1 the top chain (TW = 1), each a one-wave body on its own LDS slice (23 + 20 tiles = 90 KB: what the packed kernel's
// workgroup of two rollouts takes, so one sweep workgroup still fits next to it on the CU).  The kernel for the rounds' KKT stage
// NEXT TO the sweep: the packed one-wave recursion is a chain of H steps (0.27 - 0.33 ms whatever the number of systems, longer
// than most sweeps of the late rounds: DESIGN.md section 6), two chains of H / 2 steps are not.  The chains are real functions
// (own register allocation, see kkt_tw_chain) that end the program; they meet at a workgroup barrier inside kkt_body and start
// the line search together.  list == nullptr: rollout blockIdx.x + S.b0 (stage filter as kkt_kernel).
template <int NQ, int NU, int TW>
static __device__ __attribute__((noinline, noreturn)) void kkt_duo_chain(unsigned long long ka, int b, lds_double_ptr sm3) {
    const NewtonDev S = kkt_kernarg<NewtonDev, offsetof(KktTwArgs, S)>(ka);
    const KktArgs K = kkt_kernarg<KktArgs, offsetof(KktTwArgs, K)>(ka);
    kkt_body<NQ, NU, WaveSync, 1, false, TW>(S, K, b, (double*)sm3, (int)threadIdx.x & 63, 0);
    __builtin_amdgcn_endpgm();
}
template <int NQ, int NU>
__global__ __launch_bounds__(128, CIMPC_KKT_PACK_WAVES_PER_SIMD) void kkt_kernel_duo(KktTwArgs A) {
    int n = A.n;
    if (A.n_dev != nullptr) n = *A.n_dev;
    const int k = (int)blockIdx.x;
    if (k >= n) return;
    const int b = A.list != nullptr ? A.list[k] : k + A.S.b0;
    if (A.list == nullptr) {
        if (A.K.stage != nullptr && A.K.stage[b] != STAGE_KKT) return;
        if (A.K.only_flag != nullptr && A.K.only_flag[b] == 0) return;
    }
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __builtin_amdgcn_s_setprio(CIMPC_KKT_PRIO);
    const unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    if (((int)threadIdx.x >> 6) == 0) kkt_duo_chain<NQ, NU, 2>(ka, b, (lds_double_ptr)sm);
    else kkt_duo_chain<NQ, NU, 1>(ka, b, (lds_double_ptr)(sm + kkt_duo_slice_doubles(2)));
}

// (wide tiles: 105 KB of LDS allow one workgroup per CU anyway - let it use the 512-register budget)
template <int NQ, int NU, bool F32 = false>
__global__ __launch_bounds__(64, (kkt_tld<NQ, NU>() <= 16 ? 2 : 1)) void kkt_kernel(NewtonDev S, KktArgs K) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int b = blockIdx.x + S.b0;
    if (K.stage != nullptr && K.stage[b] != STAGE_KKT) return;
    if (K.only_flag != nullptr && K.only_flag[b] == 0) return;
    kkt_body<NQ, NU, BlockSync, 1, F32>(S, K, b, sm, (int)threadIdx.x);
}

}  // namespace cimpc
