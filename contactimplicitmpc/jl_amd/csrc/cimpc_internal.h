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
struct IpQueues {
    int* items;         // [2][K][cap]  problem id = sb*H + i
    int* count;         // [2][K]
    int* head;          // [K]      consumption cursor of items[par]
    int* s_items;       // [K][cap] converged problems waiting for their sensitivities
    int* s_count;       // [K]
    int* s_head;        // [K]
    int* done_count;    // [B*slots] finished problems of the evaluation held by each slot
    const int* window;  // [B][H+2] 0-based reference-knot indices
    int cap, K, par;
};

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
    int slots;             // evaluation slots per rollout: rollout = slot index / slots
    int H;
    cimpc_ip_opts o;
};

struct KernelInfo {
    int G;            // lanes per problem
    int lds_table;    // doubles
    int lds_group;    // doubles of per-problem scratch
    int tab_size;     // doubles per knot
};

// dims-dispatching launchers (ip_kernel.hip / newton_kernels.hip)
int ip_kernel_info(const cimpc_dims* dm, KernelInfo* info);
// launches the queue kernel followed by the sensitivity kernel
int launch_ip_sweep(const cimpc_dims* dm, const IpParams& p, int waves, hipStream_t s);

}  // namespace cimpc
