// Internal (non-ABI) declarations shared by the host runtime and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/cimpc.h"

namespace cimpc {

// Device-side work queues of the interior-point sweep.  Problems (evaluation slot sb, horizon
// position i) are bucketed by REFERENCE KNOT (window[b][i]) because all problems of a knot share one
// linearization table: the persistent workgroups of a knot stage that table once into LDS and then
// every 16-lane group pulls problems from the knot's queue until it is empty (dynamic load
// balance: solves take 4..100 iterations).  Queues are double buffered by round parity: round r
// consumes items[par], solves parked after iter_cap iterations and the evaluations requested by
// the line search are appended to items[par ^ 1].
constexpr int QPAD = 32;   // ints between two queue counters
constexpr int CPAD = 32;   // ints between two ROUND counters (NewtonDev::counters[k * CPAD]): every counter on its own 128-byte line -
                           // same-line atomics serialise in the L2 at ~50 ns each, and the decision kernel issues several per workgroup

struct IpQueues {
    int* items;         // [2][K][cap]  problem id = sb*H + i
    int* count;         // [2][K] * QPAD   (every counter on its own 128-byte line: same-line atomics
    int* head;          // [K] * QPAD       serialise at ~50 ns each and all lane groups of a knot hit them)
    int* done_count;    // [B*slots] finished problems of the evaluation held by each slot
    const int* window;  // [B][H+2] 0-based reference-knot indices
    int cap, K, par;
};

// Asynchronous (single-launch) Newton solve: every rollout advances on its own dependency chain
// sweep -> residual/decision -> KKT -> sweep ... inside ONE persistent kernel.  The stages hand work
// to each other through these device queues.  Queues never wrap within a solve (capacity = upper
// bound of the pushes of one solve); entries are pre-set to -1 so that a consumer that claimed an
// index can wait for the producer's store.
struct AsyncQ {
    int on;             // 0: lock-step rounds (queues above are snapshots), 1: asynchronous solve
    int* rq_items;      // residual/decision jobs: rollout index
    int* rq_head;
    int* rq_tail;
    int* kq_items;      // KKT jobs: rollout index
    int* kq_head;
    int* kq_tail;
    int* evals_left;    // [B] evaluation slots of the rollout's running line-search batch not yet complete
    int* n_done;        // rollouts whose Newton solve has ended
    int* epoch;         // wake-up words, 64 B apart: [0,16) interior-point work, [16,32) jobs, 32 = any job; bucket = rollout % 16.
                        // An idle workgroup polls only the two words of its own bucket (blockIdx % 16).
    volatile int* abort_flag;   // host-mapped: nonzero = time budget exhausted, leave
    long long* dbg;     // [16] diagnostics (busy ticks / job counts per kind of work) or null
    int n_service;      // workgroups [0, n_service) only take residual / KKT jobs
    int flags;          // reserved (0)
    int idle_sleep;     // back-off of an idle workgroup between polls, units of s_sleep(64) (~2 us)
    int idle_spins;     // polls before an idle workgroup looks around unprompted
    int wake_fan;       // buckets woken per evaluation request (each bucket = 1/16 of the workgroups)
    int B;
    int kkt_tw;         // 1: a rollout's KKT stage is TWO cooperating jobs - the chains of the twisted condensed solve, pushed bottom chain
                        // first (push_kkt_job) - instead of one one-ended recursion
};
// KKT job words: rollout index in the low 24 bits, kind above - 0 one-ended recursion, 1 top chain, 2 bottom chain of the twisted
// solve, 3 one-ended recursion of a rollout whose twisted hand-over timed out
constexpr int KJOB_SHIFT = 24, KJOB_MASK = (1 << KJOB_SHIFT) - 1;

// Hand-offs of the asynchronous solve follow the HIP memory model literally: the producer issues an
// agent-scope release fence (__threadfence) after writing its results and before the queue push /
// counter update, the consumer an agent-scope acquire fence after claiming the entry.  (Measured on
// MI355X: replacing the per-problem fences by write-through stores + s_waitcnt was neither faster
// nor reliably correct across XCDs, so the plain, portable form stays.)
// (`flags` is a reserved word of AsyncQueues, always 0: the timing experiments it once switched - workgroup-scope fences, no
// in-serve retry of idle lane groups - are gone.)
__device__ __forceinline__ void xfence(int /*flags*/) { __threadfence(); }
template <bool X>
__device__ __forceinline__ double xld(const double* p) { return *p; }
template <bool X>
__device__ __forceinline__ void xst(double* p, double v) { *p = v; }
template <bool X>
__device__ __forceinline__ void xst(int* p, int v) { *p = v; }

__device__ __forceinline__ int aload(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void astore(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// multi-producer push: reserve a position, then store the entry (consumers wait on the -1 sentinel)
__device__ __forceinline__ void aq_push(int* items, int* tail, int v) {
    const int pos = atomicAdd(tail, 1);
    astore(items + pos, v);
}
// multi-consumer pop by ONE lane: -1 when the queue is empty.  dbg (optional): [0] CAS attempts, [1] sentinel spins,
// [2] ticks in the claim loop, [3] ticks waiting for the entry
__device__ __forceinline__ int aq_pop(int* items, int* head, const int* tail, long long* dbg = nullptr,
                                      volatile int* abort_flag = nullptr) {
    const long long t0 = dbg ? wall_clock64() : 0;
    int h = aload(head);
    int cas = 0;
    while (true) {
        if (h >= aload(tail)) { if (dbg) { dbg[0] += cas; dbg[2] += wall_clock64() - t0; } return -1; }
        const int old = atomicCAS(head, h, h + 1);
        ++cas;
        if (old == h) break;
        h = old;
    }
    const long long t1 = dbg ? wall_clock64() : 0;
    int v, spins = 0;
    while ((v = aload(items + h)) < 0) {
        __builtin_amdgcn_s_sleep(1);
        // (the producer reserved this entry before we could claim it, so the wait is short; the host's
        //  watchdog / time budget can still end it)
        if ((++spins & 0xFFFF) == 0 && abort_flag != nullptr && *abort_flag != 0) return -1;
    }
    astore(items + h, -1);        // the queue is clean again when every entry has been consumed
    if (dbg) { dbg[0] += cas; dbg[1] += spins; dbg[2] += t1 - t0; dbg[3] += wall_clock64() - t1; }
    return v;
}

// an evaluation spreads over up to H knots and every knot needs a workgroup of its own: wake `wake_fan` buckets
__device__ __forceinline__ void wake_ip(const AsyncQ& A, int b) {
    for (int k = 0; k < A.wake_fan; ++k) atomicAdd(A.epoch + ((b + k * (16 / A.wake_fan)) & 15) * 16, 1);
}
// jobs head the rollout's chain: the (few) dedicated service workgroups all wake on word 32, the
// interior-point workgroups of the rollout's bucket on their job word
__device__ __forceinline__ void wake_job(const AsyncQ& A, int b) {
    atomicAdd(A.epoch + 32 * 16, 1);
    atomicAdd(A.epoch + (16 + (b & 15)) * 16, 1);
}
// A rollout enters its KKT stage.  Twisted form: two consecutive entries, the BOTTOM chain first.  Claims are FIFO (aq_pop advances
// the head by compare-and-swap), so whoever claims the top chain knows that the bottom chain already has a workgroup - every
// workgroup of the persistent kernel is resident and a claimed job is started without waiting for anything - and the bottom chain,
// which waits for nothing until its forward pass is done, is followed by its partner as soon as any workgroup looks at the queue
// (jobs are taken before interior-point work).  Both waits are bounded anyway (kkt_tw_wait): a time-out re-queues kind 3.
__device__ __forceinline__ void push_kkt_job(const AsyncQ& A, int b) {
    if (A.kkt_tw) {
        const int pos = atomicAdd(A.kq_tail, 2);
        astore(A.kq_items + pos, b | (2 << KJOB_SHIFT));
        astore(A.kq_items + pos + 1, b | (1 << KJOB_SHIFT));
        atomicAdd(A.epoch + 32 * 16, 1);
        atomicAdd(A.epoch + (16 + (b & 15)) * 16, 1);
        atomicAdd(A.epoch + (16 + ((b + 1) & 15)) * 16, 1);      // (a second bucket: two workgroups are wanted)
    } else {
        aq_push(A.kq_items, A.kq_tail, b);
        wake_job(A, b);
    }
}
__device__ __forceinline__ void wake_all(const AsyncQ& A) { for (int k = 0; k <= 32; ++k) atomicAdd(A.epoch + k * 16, 1); }

__device__ __forceinline__ int* qcount(const IpQueues& Q, int par, int k) { return Q.count + ((size_t)par * Q.K + k) * QPAD; }
__device__ __forceinline__ int* qhead(const IpQueues& Q, int k) { return Q.head + (size_t)k * QPAD; }

struct IpParams {
    const double* tab;     // [H_ref][LinLayout::size]   packed linearization tables
    IpQueues Q;
    int wpk;               // persistent workgroups of the launch (each serves one knot at a time)
    const double* q;       // [B*slots][H+2][nq]   trajectory being evaluated (q_{i+2} = IP start)
    const double* theta;   // [B*slots][H][nth]
    const double* gam;     // [B*slots][H][nc]   (configurationforce mode) or null
    const double* bfr;     // [B*slots][H][nb]   (configurationforce mode) or null
    const double* alt;     // [B][nc] altitude offsets (RLin.alt) or null
    double* d;             // [B*slots][H][nd]            dynamics violation
    double* dz;            // [B*slots][H][nths][nd]      column-major nd x (2nq+nu) sensitivities
    // delta^T nu: the products residual! forms with the sensitivities (newton_residual.jl:125-127: dq0[t]^T nu_i, dq1[t]^T nu_i,
    // du1[t]^T nu_i), emitted by the solve that produced them while the columns are still in registers - the decision stage then
    // reads nths doubles per solve instead of the whole nths x nd block (58 MB per launch at B = 512, L1-thrashing column reads)
    const double* nu;      // [B*slots][H][nd] dual candidate of the evaluation (null: no products, e.g. the B3 seam)
    double* dtn;           // [B*slots][H][dtn_ld]  (dz column c)^T nu, c = 0 .. nths-1; null: not produced
    int dtn_ld;
    int* status;           // [B*slots][H]  1 = converged
    int* iters;            // [B*slots][H]  IP iterations
    double* zout;          // [B*slots][H][nz] converged z (optional, null = skip)
    // resumable solves: a problem runs at most iter_cap IP iterations per launch; an unfinished
    // solve parks its exact state and is re-queued for the next round, so that one hard instance
    // (up to max_iter = 100 iterations) never holds a launch
    int* pflag;            // [B*slots][H] 1 = parked (resume from pstate)
    double* pstate;        // [B*slots][H][2nx + 4ny + 4]  parked iterate / converged z for the sens pass
    int* pending_count;    // device counter, incremented once per parked problem
    int iter_cap;
    // drain parking (experiment, CIMPC_DRAIN_PCT): once `drain_thresh` workgroups of the launch have run out of work, a solve
    // that has had at least `drain_min` iterations in this launch is parked instead of holding the launch (and the round)
    int* drain_count;      // workgroups of this launch that found no work any more (null = off); zeroed with the round's counters
    int drain_thresh, drain_min;
    int slots;             // evaluation slots per rollout: rollout = slot index / slots
    int H;
    cimpc_ip_opts o;
    // uniform constants of an iteration, formed on the host (IEEE division / subtraction: the values the kernel's own expressions
    // give) so that they arrive as scalars: computed in the kernel they were loop-invariant VECTOR registers that spilled to
    // scratch and were re-loaded in every interior-point trip
    double kc_floor;    // o.kappa_tol / o.undercut
    double tau_floor;   // 1 - o.eps_min
    double reg_floor;   // o.kappa_tol * o.gamma_reg (regularisation floor of differentiate_solution!)
    // per-solve time budget (cimpc_ip_opts::max_time) in ticks of the constant-rate device clock; 0 = unlimited (no clock read)
    long long budget_ticks;
    int generic_static;    // runtime-dimension sweep (ip_generic.hip): 1 = static partition of a knot's queue instead of the dynamic pull
    int direct;            // 1: workgroup k serves knot k's queue and leaves (launches of single rollouts: at most a handful of problems per knot -
                           // the two remaining-work scans of the persistent form, ~3 us each, are a tenth of such a launch); grid = K
    AsyncQ A;
};

// B2 seam (cimpc_ip_residual / cimpc_ip_linear_solve): one reference-side callback on n caller-supplied points of knot `knot`
struct IpCallbackArgs {
    const double* tab;     // packed linearization tables (device)
    int knot, n, op;       // op 0: rlin!   op 1: rzlin! + linear_solve!
    const double* z;       // [n][nz]
    const double* theta;   // [n][nth]  (op 0)
    const double* alt;     // [n][nc] or null (op 0)
    const double* r;       // [n][nz]   (op 1)
    double kappa, reg;
    double* out;           // [n][nz]
};

struct KernelInfo {
    int G;            // lanes per problem
    int lds_table;    // doubles
    int lds_group;    // doubles of per-problem scratch
    int tab_size;     // doubles per knot
    int dtn_ld;       // leading dimension of the delta^T nu products the sweep emits (0: the kernel does not emit them)
    int adj;          // 1: the table carries the constants of the adjoint sensitivity pass (lin_table.h: oK0, oAiB)
    int generic;      // 1: no compiled lane-group kernel for these dimensions - the runtime-dimension kernel (ip_generic.hip) serves them
};

// Opt a kernel in to more than 64 KiB of dynamic LDS (gfx950: 160 KiB per CU).  The attribute call is not free,
// so it is made once per (kernel instantiation, device) - `cache` is a function-local static of the caller.
struct LdsOptIn { size_t set[16] = {0}; };
inline int lds_opt_in(LdsOptIn& cache, const void* kernel, size_t lds) {
    if (lds <= 64 * 1024) return CIMPC_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return CIMPC_ERR_HIP;
    dev &= 15;
    if (lds <= cache.set[dev]) return CIMPC_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return CIMPC_ERR_HIP;
    cache.set[dev] = lds;
    return CIMPC_OK;
}

// dims-dispatching launchers (ip_kernel.hip / newton_kernels.hip)
int ip_kernel_info(const cimpc_dims* dm, KernelInfo* info);
// launches the queue kernel followed by the sensitivity kernel
int launch_ip_sweep(const cimpc_dims* dm, const IpParams& p, int waves, hipStream_t s);
// B2 seam: compiled lane-group models only (CIMPC_ERR_INVALID otherwise)
int launch_ip_callback(const cimpc_dims* dm, const IpCallbackArgs& a, hipStream_t s);
// runtime-dimension fallback (ip_generic.hip): nx, ny <= 64, one problem per wavefront
bool ip_generic_available(const cimpc_dims* dm);
void ip_generic_info(const cimpc_dims* dm, KernelInfo* info);
int launch_ip_generic(const cimpc_dims* dm, const IpParams& p, hipStream_t s);

}  // namespace cimpc
