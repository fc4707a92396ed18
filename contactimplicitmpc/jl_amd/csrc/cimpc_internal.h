// Internal (non-ABI) declarations shared by the host runtime and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/cimpc.h"

namespace cimpc {

// Work descriptor of the batched interior-point sweep: problems (rollout b, horizon
// position i) are bucketed by reference knot so that one workgroup stages ONE knot's
// linearization table into LDS and serves up to PW problems that use it.
struct IpParams {
    const double* tab;     // [H_ref][LinLayout::size]   packed linearization tables
    const int* wg_desc;    // [n_wg][4]  {knot, first index into plist, count, 0}
    const int* plist;      // [n_prob]   b*H + i, sorted by knot
    const double* q;       // [B][H+2][nq]   trajectory being evaluated (q_{i+2} = IP start)
    const double* theta;   // [B][H][nth]
    const double* gam;     // [B][H][nc]   (configurationforce mode) or null
    const double* bfr;     // [B][H][nb]   (configurationforce mode) or null
    const double* alt;     // [B][nc] altitude offsets (RLin.alt) or null
    const int* need_sweep; // [B] per-rollout flag (null = all rollouts)
    double* d;             // [B][H][nd]            dynamics violation
    double* dz;            // [B][H][nths][nd]      column-major nd x (2nq+nu) sensitivities
    int* status;           // [B][H]  1 = converged
    int* iters;            // [B][H]  IP iterations
    double* zout;          // [B][H][nz] converged z (optional, null = skip)
    // resumable solves: a launch runs at most iter_cap IP iterations per problem; unfinished
    // problems park their iterate here and continue in the next launch, so that one hard
    // instance (up to max_iter = 100 iterations) never stalls the whole batch
    int* pflag;            // [B][H] 0 = fresh, 1 = pending (resume), 2 = done for this evaluation
    double* pstate;        // [B][H][2nx + 4ny + 4]
    int* pending_count;    // device counter, incremented once per parked problem
    int iter_cap;
    int slots;             // evaluation slots per rollout: the index b below is a slot, rollout = b / slots
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
int launch_ip_sweep(const cimpc_dims* dm, const IpParams& p, int n_wg, int waves, hipStream_t s);

}  // namespace cimpc
