// Single-launch asynchronous Newton solve.
//
// The lock-step driver (cimpc_host.cpp) advances ALL rollouts of a batch round by round:
// {sweep || KKT} -> residual/decision, one launch each, ~30-50 rounds per MPC step, every round
// paying the tail of its slowest interior-point solve plus launch gaps.  Here the whole
// newton_solve! (/root/reference/src/controller/newton.jl:169-288) of the batch is ONE persistent
// kernel: workgroups loop over three kinds of work,
//     KKT job        (rollout)   one wavefront, kkt_body            -> pushes the alpha = 1 evaluation
//     residual job   (rollout)   whole workgroup, resid_decide_body -> pushes line-search evaluations,
//                                                                     a KKT job, or ends the rollout
//     knot service   (knot)      serve_knot: interior-point solves + sensitivities; the group that
//                                completes a rollout's outstanding evaluations pushes its residual job
// so every rollout advances along its own dependency chain and the GPU is kept busy by the other
// rollouts.  The results per rollout are those of the lock-step rounds (same bodies, same arithmetic;
// no solve is ever parked).  Hand-offs: release fence -> queue push / counter, claim -> acquire fence
// (agent scope; the queues are in device memory).
//
// Deadlock freedom: no unit ever waits for a specific other unit - only for queue entries whose
// producer has already reserved the slot (aq_pop) - and any workgroup can execute any job, so the
// solve progresses with however many workgroups the hardware keeps resident.
#pragma once
#include <algorithm>
#include <cstddef>
#include "ip_kernel_impl.h"
#ifdef CIMPC_KKT_PROF
#undef CIMPC_KKT_PROF        // the KKT phase clocks are read from the lock-step kernels only (and this translation unit does not compile with them)
#endif
#include "newton_impl.h"

namespace cimpc {

struct AsyncArgs {     // the kernel's only argument
    IpParams p;
    NewtonDev S;
};

// The two horizon-level jobs are compiled as real functions: their register allocation stays out of
// the interior-point loop (which needs all 256 VGPRs of the 2-waves-per-SIMD budget for itself).  They
// read the solver state straight from the kernel-argument segment (scalar loads), exactly like a
// kernel of their own would.
// (the kernel passes the address of its argument segment; the callee makes it wave-uniform again so
// that the fields are fetched with scalar loads from the constant address space)
template <class T, size_t OFFSET>
__device__ __forceinline__ T uniform_arg(unsigned long long v) {
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
__device__ __forceinline__ NewtonDev uniform_state(unsigned long long v) { return uniform_arg<NewtonDev, offsetof(AsyncArgs, S)>(v); }
template <int NQ, int NU>
__device__ __noinline__ void async_kkt_job(unsigned long long ka, int b, double* smem, int tid) {
    // entered by the WHOLE workgroup: waves 0 and 1 run the software-pipelined recursion (kkt_body PIPE = 2: the
    // KKT solve heads a rollout's dependency chain here), the other waves only keep the workgroup barriers of
    // the pipeline company (one after the LDS clear, one per tick, one before the backward pass)
    const NewtonDev S = uniform_state(ka);
    if ((int)blockDim.x >= 192) {         // three-stage pipeline on waves 0..2; the fourth wave keeps the barriers company
        // Round 6: the line search is started by the WHOLE workgroup after the solve (finish = 0 inside kkt_body): on wave 0 alone the seven
        // candidates of a tail rollout (apply_steps + 280 queue pushes) took longer than a third of the recursion itself
        if (tid < 192) {
            const KktArgs K{S.res, S.delta, S.beta, 0.0, S.stage, 0};
            kkt_body<NQ, NU, WaveSync, 3>(S, K, b, smem, tid & 63, tid >> 6);
        } else {
            for (int k = 0; k < S.dm.H + 4; ++k) __syncthreads();      // 1 (LDS clear) + H + 2 ticks + 1 (before the backward pass)
        }
        __threadfence_block();
        __syncthreads();                   // wave 0's step (global memory, this workgroup's own stores) is in front of every wave
        start_line_search<BlockSync>(S, b, 2, tid, (int)blockDim.x);
    } else if ((int)blockDim.x >= 128) {
        const KktArgs K{S.res, S.delta, S.beta, 0.0, S.stage, 2};
        kkt_body<NQ, NU, WaveSync, 2>(S, K, b, smem, tid & 63, tid >> 6);
    } else if (tid < 64) {
        const KktArgs K{S.res, S.delta, S.beta, 0.0, S.stage, 2};
        kkt_body<NQ, NU, WaveSync>(S, K, b, smem, tid);
    }
}
// Twisted form of the KKT job (round 6): ONE chain of the two-ended condensed solve (kkt_body<..., PIPE = 3, TW>: the
// three-stage pipeline on waves 0..2 from its own end of the matrix), the rollout's other chain being another workgroup's job
// (push_kkt_job: bottom chain first, FIFO claims).  The hand-over flags are epoch-valued; inside one persistent launch the stamp of a
// rollout's KKT stage is the launch's stamp + 1 + the Newton iterations the rollout has done (the host advances its stamp by
// max_iter + 2 per launch).  The chain that finishes last starts the line search with the whole workgroup; a hand-over that timed
// out re-queues the stage as a one-ended job (kind 3).
template <int NQ, int NU, int TW>
__device__ __noinline__ void async_kkt_tw_job(unsigned long long ka, int b, double* smem, int tid) {
    NewtonDev S = uniform_state(ka);
    S.kkt_tw_epoch += 1 + __builtin_amdgcn_readfirstlane(S.newton_l[b]);
    constexpr int TL = kkt_tld<NQ, NU>();
    const int H = S.dm.H;
    const int msp = kkt_tw_split(H, S.kkt_tw_nb, TL > 16, false);
    const int NS = TW == 1 ? msp + 2 : (H - msp - 2) + 2;
    if (tid < 192) {
        const KktArgs K{S.res, S.delta, S.beta, 0.0, S.stage, 0};
        kkt_body<NQ, NU, WaveSync, 3, false, TW>(S, K, b, smem, tid & 63, tid >> 6);
    } else {
        // the fourth wave keeps the chain's workgroup barriers company: 1 (LDS clear) + NS + 2 ticks + 1 (before the backward pass) +
        // 2 (primal recovery) + 1 (rows complete) + 1 (verdict of the finish counter)
        for (int k = 0; k < NS + 8; ++k) __syncthreads();
    }
    __syncthreads();
    // verdict of the finish counter (kkt_body: vec[11 VS], written by wave 0 before its last barrier): non-zero = the partner had finished
    const int last = (int)smem[KKT_TW_TILES * TL * TL + 11 * (TL <= 16 ? 16 : 32)];
    if (last == 0) return;
    __threadfence();                      // acquire the partner's share of the step in every wave
    int* xfl = S.kkt_tw_flags + (size_t)b * KKT_TW_FLAGS;
    if (__builtin_amdgcn_readfirstlane(aload(xfl + 3)) == S.kkt_tw_epoch) {      // poisoned numbers: the stage again, one-ended
        if (tid == 0) { aq_push(S.A.kq_items, S.A.kq_tail, b | (3 << KJOB_SHIFT)); wake_job(S.A, b); }
        return;
    }
    start_line_search<BlockSync>(S, b, 2, tid, (int)blockDim.x);
}
template <int NQ, int NU>
__device__ __noinline__ void async_resid_job(unsigned long long ka, int b, double* red, double* rc, int* sh) {
    const NewtonDev S = uniform_state(ka);
    resid_decide_body<NQ, NU, false, true>(S, b, red, rc, sh);
}

// The knot service is a real function too: the interior-point loop then gets the register allocation it has
// in the lock-step kernel, undisturbed by what the job functions clobber (with the loop inlined next to the
// job calls the allocator spilled loop-invariant values "everywhere": +30 % interior-point time).
template <class M>
__device__ __noinline__ void async_serve_job(unsigned long long ka, double* smem, int knot, int tid) {
    const IpParams p = uniform_arg<IpParams, offsetof(AsyncArgs, p)>(ka);
    serve_knot<M, true>(p, smem, knot, tid);
}

template <class M>
__global__ __launch_bounds__(256, (M::G == 16 ? 2 : 1)) void newton_async_kernel(AsyncArgs args) {
    const IpParams& p = args.p;
    const unsigned long long ka = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    static_assert(M::MODE == CIMPC_MODE_CONFIGURATION, "the KKT stage implements :configuration");
    constexpr int NQ = M::NQ, NU = M::NU;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int s_knot, s_total, s_rem[PICK_MAXK], s_job[2], s_epoch[2];
    __shared__ double rc[CS];
    __shared__ int sh[4];
    const int tid = (int)threadIdx.x;
    const AsyncQ& A = p.A;
    const bool service = (int)blockIdx.x < A.n_service;        // these only take residual / KKT jobs
    const int nwork = max(1, (int)gridDim.x - A.n_service), wg = max(0, (int)blockIdx.x - A.n_service);
    unsigned trip = 0;
    long long tk = wall_clock64();
    auto account = [&](int slot) {          // busy time per kind of work, 100 MHz ticks (diagnostics)
        if (tid == 0 && A.dbg != nullptr) {
            const long long tn = wall_clock64();
            atomicAdd((unsigned long long*)A.dbg + slot, (unsigned long long)(tn - tk));
            atomicAdd((unsigned long long*)A.dbg + 8 + slot, 1ull);
            tk = tn;
        }
    };
    while (true) {
        __syncthreads();
        if (tid == 0) {
            int type = 0, job = -1;
            // wake-up words of this workgroup's bucket, read BEFORE looking for work (a push during the scan changes them)
            s_epoch[0] = aload(A.epoch + ((int)blockIdx.x & 15) * 16);
            s_epoch[1] = aload(A.epoch + (service ? 32 : 16 + ((int)blockIdx.x & 15)) * 16);
            // the abort flag lives in host memory (one PCIe read): looked at now and then only
            if ((trip++ & 31u) == 31u && *A.abort_flag != 0) {
                type = 3;
            } else if ((job = aq_pop(A.kq_items, A.kq_head, A.kq_tail, nullptr, A.abort_flag)) >= 0) {
                type = 1;       // KKT first: it heads the longest chain of a rollout
            } else if ((job = aq_pop(A.rq_items, A.rq_head, A.rq_tail, nullptr, A.abort_flag)) >= 0) {
                type = 2;
            }
            s_job[0] = type; s_job[1] = job;
        }
        __syncthreads();
        const int type = s_job[0], job = s_job[1];
        if (type == 3) break;
        if (type == 1) {
            xfence(A.flags);                             // acquire res / traj / dz of the rollout
            account(0);
            const int kind = job >> KJOB_SHIFT, jb = job & KJOB_MASK;
            if constexpr (kkt_tld<NQ, NU>() == 16) {
                if ((kind == 1 || kind == 2) && (int)blockDim.x >= 256) {
                    if (kind == 2) async_kkt_tw_job<NQ, NU, 2>(ka, jb, smem, tid);
                    else async_kkt_tw_job<NQ, NU, 1>(ka, jb, smem, tid);
                    __syncthreads();
                    account(1);
                    continue;
                }
            }
            async_kkt_job<NQ, NU>(ka, jb, smem, tid);
            __syncthreads();
            account(1);
            continue;
        }
        if (type == 2) {
            xfence(A.flags);                             // acquire d / dz / status of the evaluations
            account(0);
            async_resid_job<NQ, NU>(ka, job, smem, rc, sh);      // reduction scratch [CS][256] + [N] in the (idle) table area
            __syncthreads();
            account(2);
            continue;
        }
        if (!service) {
            const int knot = pick_knot(p, s_rem, &s_total, &s_knot, tid, wg, nwork);
            if (knot >= 0) {
                account(0);
                async_serve_job<M>(ka, smem, knot, tid);
                __syncthreads();
                account(3);
                continue;
            }
        }
        // Nothing to do right now.  Idle workgroups must stay off the queue counters (hundreds of
        // pollers on a handful of cache lines inflate the memory latency of the working ones ~10x:
        // measured SQ_INST_LEVEL_VMEM / SQ_INSTS_VMEM 16 -> 200).  They wait on the two wake-up words of
        // their bucket, which producers bump after publishing, and look around anyway every ~100 us.
        if (tid == 0) {
            int leave = aload(A.n_done) >= A.B;
            const int* wi = A.epoch + ((int)blockIdx.x & 15) * 16;
            const int* wj = A.epoch + (service ? 32 : 16 + ((int)blockIdx.x & 15)) * 16;
            unsigned spins = 0;
            while (!leave && (service || aload(wi) == s_epoch[0]) && aload(wj) == s_epoch[1]) {
                for (int k = 0; k < A.idle_sleep; ++k) __builtin_amdgcn_s_sleep(64);
                if (++spins >= (unsigned)A.idle_spins) break;
            }
            if (!leave && *A.abort_flag != 0) leave = 1;
            s_job[0] = leave;
        }
        __syncthreads();
        if (s_job[0]) break;
    }
}

template <class M>
int launch_async_model(const IpParams& p, const NewtonDev& S, int waves, int grid, hipStream_t s) {
    constexpr LinLayout L(M::NX, M::NY, M::NTH, M::G, M::NTHS, M::ADJ);
    if (waves != 1 && waves != 2 && waves != 4) return CIMPC_ERR_INVALID;
    const int ppw = 64 / M::G;
    const size_t lds_ip = (size_t)(L.size + waves * ppw * M::LDS_GROUP) * sizeof(double);
    size_t lds_kkt = (size_t)(waves >= 3 ? kkt_lds_doubles<M::NQ, M::NU, 3>() : waves >= 2 ? kkt_lds_doubles<M::NQ, M::NU, 2>() : kkt_lds_doubles<M::NQ, M::NU, 1>()) * sizeof(double);
    if (p.A.kkt_tw) {      // twisted KKT jobs: four-wave workgroups, 16-wide tiles (the host sets the flag only where both hold)
        if (waves != 4 || kkt_tld<M::NQ, M::NU>() != 16) return CIMPC_ERR_INVALID;
        lds_kkt = std::max(lds_kkt, (size_t)kkt_tw_lds_doubles<M::NQ, M::NU>() * sizeof(double));
    }
    const size_t lds_res = (size_t)(CS * 256 + S.N) * sizeof(double);       // residual job: partial sums of every slot + |r_e| scratch
    const size_t lds = std::max(std::max(lds_ip, lds_kkt), lds_res);
    static LdsOptIn optin;
    if (lds_opt_in(optin, (const void*)newton_async_kernel<M>, lds) != CIMPC_OK) return CIMPC_ERR_HIP;
    // every workgroup of this kernel must be resident (they wait for each other's jobs): clamp the grid to what the
    // device really holds for this kernel / block size / LDS footprint
    static int max_resident[5] = {0, 0, 0, 0, 0};
    if (max_resident[waves] == 0) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)newton_async_kernel<M>, 64 * waves, lds) != hipSuccess ||
            hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || per_cu < 1)
            return CIMPC_ERR_HIP;
        max_resident[waves] = per_cu * prop.multiProcessorCount;
    }
    if (grid > max_resident[waves]) grid = max_resident[waves];
    if (grid <= S.A.n_service) return CIMPC_ERR_INVALID;          // no workgroup left for the interior-point queues
    AsyncArgs args{p, S};
    hipLaunchKernelGGL((newton_async_kernel<M>), dim3(grid), dim3(64 * waves), lds, s, args);
    return hipGetLastError() == hipSuccess ? CIMPC_OK : CIMPC_ERR_HIP;
}

#define CIMPC_DEFINE_ASYNC_MODEL(name, q, u, w, c, b)                                                   \
    int async_launch_##name(const IpParams& p, const NewtonDev& S, int waves, int grid, hipStream_t s) { \
        return launch_async_model<Model<q, u, w, c, b, 0>>(p, S, waves, grid, s);                       \
    }

}  // namespace cimpc
