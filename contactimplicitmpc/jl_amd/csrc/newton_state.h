// Device-resident state of the batched Newton solver (one slot per rollout), the HBM
// counterpart of the reference's `Newton` workspace (/root/reference/src/controller/newton.jl:17-35)
// and `ContactTraj` containers (trajectory.jl:1-19).  Plain pointers, SoA, every array has a
// leading rollout dimension B.
#pragma once
#include "cimpc_internal.h"

namespace cimpc {
constexpr int NLOG = 16;       // Newton iterations kept per rollout in the status log

enum Stage : int { STAGE_DONE = 0, STAGE_INIT = 1, STAGE_LS0 = 2, STAGE_KKT = 3, STAGE_LS1 = 4, STAGE_LS2 = 5, STAGE_LS7 = 6, STAGE_LS3 = 7 };
constexpr int CS = 7;   // evaluation slots per rollout (speculative line search, newton_impl.h)

struct TrajDev {
    double* q;    // [B][H+2][nq]
    double* u;    // [B][H][nu]
    double* w;    // [B][H][nw]
    double* g;    // [B][H][nc]
    double* b;    // [B][H][nb]
    double* th;   // [B][H][nth]
};

struct NewtonDev {
    cimpc_dims dm;
    int b0;            // first rollout served by this launch (sub-batch offset)
    int nb_launch;     // rollouts served by this launch
    int kkt_same_round; // 1: KKT runs before the sweep on the same stream (small batches); 2: chained round - KKT next to the first sweep,
                        //    its candidates are evaluated by the round's second sweep (queue par ^ 1)
    int kkt_scalar;     // 1: the non-MFMA condensed kernel everywhere (CIMPC_KKT_SCALAR, diagnostic)
    int nd, nr, nth, nths, N;
    TrajDev traj, cand, ref;   // cand: [B*CS] evaluation slots; traj, ref: [B]
    double* nu;        // [B][H][nd]
    double* nu_cand;   // [B*CS][H][nd]
    // implicit dynamics of the last sweep
    double* d;         // [B*CS][H][nd]
    double* dz;        // [B*CS][H][nths][nd]
    double* dtn;       // [B*CS][H][dtn_ld] delta^T nu products of every solve, emitted by the sweep (IpParams::dtn); null: not produced
    int dtn_ld;
    double* dz_good;   // [B][H][nths][nd]  sensitivities of the rollout's ACCEPTED evaluation (the reference's ip[t].dz
                       // after the last accepted implicit_dynamics!): Jacobian data of the KKT stage and the value a
                       // failed solve falls back to (see dz_eff in newton_impl.h)
    // Round 6: the copy into dz_good is LAZY where the KKT stage is the condensed MFMA solve (good_src != null).  An accept records, per
    // step, WHICH slot holds the block (good_src[b][i] = slot, -1 = dz_good itself); the KKT kernel that follows - it streams every
    // block of the rollout through its registers anyway - reads through the index and writes the blocks to dz_good on its way
    // (fire-and-forget stores off the recursion's chain), then resets the index.  Invariant: outside a rollout's window
    // [accept, end of its next KKT stage] every entry is -1; dz_commit_kernel flushes what a solve leaves behind at its end
    // (rollouts that finished without another KKT stage).  The decision kernel's accepting blocks - the slowest of every round -
    // no longer move 105 KB each.
    int* good_src;     // [B][H] or null (copy at the accept, as rounds 3-5)
    int* ip_status;    // [B*CS][H]
    int* ip_iters;     // [B*CS][H]
    int* pflag;        // [B*CS][H] resumable-solve flags (see IpParams)
    int* cur_slot;     // [B] slot holding the current im_traj (d, dz) of the rollout
    IpQueues WQ;       // device work queues of the sweep (WQ.par = parity of the running round)
    AsyncQ A;          // hand-off queues of the asynchronous single-launch solve (A.on = 0: lock-step rounds)
    // Newton vectors, reference layout (newton_residual.jl:69-98)
    double* res;       // [B][N]
    double* res_cand;  // [B*CS][N]
    double* delta;     // [B][N]
    // per-rollout scalars
    double* r_norm;    // [B]  |res|_1
    double* r_cand;    // [B*CS]
    double* alpha;     // [B]
    double* beta;      // [B]
    int* ls_iter;      // [B]
    int* newton_l;     // [B]  Newton iterations done
    int* stage;        // [B]
    int* kkt_list;     // [2][B] rollouts that entered STAGE_KKT, per round parity (compact list for the packed KKT launch)
    int* need_sweep;   // [B*CS]
    int* slot_list;    // [2][B*CS] evaluation slots requested for a round (by queue parity, count in counters[4]): the residual launch
                       // of that round gets one workgroup per ENTRY instead of one per (rollout, slot) pair - idle blocks are not free
                       // (3584 blocks, 15 % live: 50 us per launch, most of it block dispatch)
    int* counters;     // [8 * CPAD], counter k at k * CPAD: 0 = #rollouts needing a sweep, 1 = #needing KKT, 2 = parked solves, 3 = drained workgroups, 4 = evaluation slots requested of the next round, 7 = block ticket
    int* counters_next; // counter block of the next round (zeroed by the residual kernel)
    int* host_flag;    // host-mapped pinned {n_sweep, n_kkt, stamp}
    int round_stamp;   // value published to host_flag[2] when the round is complete
    int band_reduce;   // banded LDL^T backend: 1 = eliminate the controls first (every R_t inverted on the host), 0 = full interleaved form
    int band_form;     // ... test handle (CIMPC_BANDED_FORM): bit 0 = four pivots per block, bit 1 = no power-of-two padding of the window (bit 2, host only: controls kept)
    long long* stats;  // [B][4] per rollout: sweeps, ip_solves, ip_iters, ip_failures of the running solve, speculative evaluations included
                       // (plain stores of the rollout's decision workgroup - a global counter would be 4 same-line atomics per workgroup; the host sums)
    int* ro_sweeps;    // [B] implicit_dynamics! evaluations of the last solve
    int* ro_ip_iters;  // [B] interior-point iterations of the last solve
    int* ro_ip_fail;   // [B] failed interior-point solves of the last solve
    // objective (objective.jl), per horizon step, column-major blocks
    const double* Q;     // [H][nq*nq]
    const double* R;     // [H][nu*nu]
    const double* Qinv;  // [H][nq*nq]
    const double* Rinv;  // [H][nu*nu]
    const double* Cg;    // [H][nc*nc] (cf) or null
    const double* Cb;    // [H][nb*nb] (cf) or null
    const double* V;        // [H][nq*nq] TrackingVelocityObjective weights or null (objective.jl:18-47)
    const double* q_target; // [H][nq] (velocity objective) or null
    const double* v_target; // [H][nq] (velocity objective) or null
    // KKT workspace: per rollout H * (3*nd*nd + nd) doubles (L1, L2, L0inv, y)
    double* kkt_ws;
    // twisted condensed solve (newton_impl.h: kkt_body<..., TW>): per rollout the exchange block of the two chains
    // ([S00 | S11 | S10^T | c0 | c1 | dnu_{m+1} | dnu_m], kkt_tw_xch_doubles(nd)) and one line of flags (KKT_TW_FLAGS ints)
    double* kkt_tw_xch;
    int* kkt_tw_flags;
    // The flags are EPOCH-valued (round 6): a producer stores the launch's stamp `kkt_tw_epoch` (the host bumps it for every
    // KKT stage), a consumer waits for equality and never resets - a partner that arrives after its waiter gave up cannot leave a
    // raised flag behind for the next solve of the rollout.  A wait is bounded by `kkt_tw_spins` polls; a chain that gives up marks
    // the rollout (flag word 3 = epoch) and counts in `kkt_tw_fail` (host-mapped): the chain that finishes last then re-queues the
    // rollout's KKT stage instead of starting the line search on poisoned numbers, and the host serves it with the one-ended kernel.
    int kkt_tw_epoch;
    int kkt_tw_spins;
    int* kkt_tw_fail;  // host-mapped counter of timed-out hand-overs (null: not counted)
    int kkt_tw_nb;     // rows the bottom chain eliminates (0: kkt_tw_split's default; CIMPC_KKT_TW_NB)
    int kkt_tw_raw;    // 1: the B1 seam (cimpc_kkt_solve: a lone solve, latency-bound) takes the twisted kernel too
    int kkt_tw_band;   // 1: the banded LDL^T takes its twisted form (two workgroups per rollout) where it is available (kkt_dense.hip)
    // options
    double r_tol, beta_init, kappa;
    int max_iter;
    double* nlog;      // [B][NLOG][4] per accepted Newton iteration: alpha, |r|_1/N before, after, line-search iter (print_status, newton.jl:290-301)
    int spec_first;    // candidates of the FIRST line-search round of a solve's first Newton iteration (1, 3 or 7): no history yet
    int spec_mid;      // previous search ended at iter >= spec_mid: the first round evaluates 1, 1/2, 1/4 together (STAGE_LS3)
    int spec_all;      // a rollout whose previous line search ended at iter >= spec_all evaluates all 7 step lengths at once
};

// single-launch asynchronous solve (newton_async_impl.h)
bool newton_async_available(const cimpc_dims* dm);
int launch_newton_async(const cimpc_dims* dm, const IpParams& p, const NewtonDev& S, int waves, int grid, hipStream_t s);
int launch_async_handoff(const NewtonDev& S, const IpQueues& lockstep_next, hipStream_t s);
int launch_mpc_advance(const NewtonDev& S, int* window, const double* stride, int H_ref, hipStream_t s);
// the controller's full reference trajectory (shared by the rollouts) and the per-rollout step counter
struct GaitDev {
    const double *q, *u, *w, *g, *b, *th;   // [H_ref+2][nq], [H_ref][nu | nw | nc | nb | nth]
    const double* stride;                    // [nq]
    int* phase;                              // [B] rot_n_stride! steps applied so far
    int H_ref;
};
int launch_gait_window(const NewtonDev& S, const GaitDev& G, int* window, int advance, hipStream_t s);
int launch_reset(const NewtonDev& nd, const double* q0, const double* q1, int warm, hipStream_t s);
int launch_dz_commit(const NewtonDev& nd, hipStream_t s);      // flush of the lazily committed sensitivities (no-op without good_src)
bool kkt_lazy_commit_available(const NewtonDev& nd);            // every KKT stage of these dimensions runs kkt_body (fp64 tiles)
int launch_dz_rekey(const NewtonDev& nd, double* knot, const int* window, int which, int dir, hipStream_t s);
int launch_resid_decide(const NewtonDev& nd, hipStream_t s, int n_slots = -1, int phase = 0);   // n_slots: entries of slot_list[WQ.par] (-1: every (rollout, slot) pair gets a block)
int launch_enqueue_all(const NewtonDev& nd, hipStream_t s);   // B3 seam: queue slot 0 of every rollout
int launch_kkt(const NewtonDev& nd, hipStream_t s);
bool kkt_condensed_available(const NewtonDev& nd);
// mixed precision: block products on the fp32 MFMA, fp64 matrix-free refinement, fp64 fallback (newton_kernels.hip)
bool kkt_mixed_available(const NewtonDev& nd);
size_t kkt_mixed_workspace_doubles(const NewtonDev& nd);
int launch_kkt_mixed_newton(const NewtonDev& nd, double* ws, int* n_fallback, hipStream_t s);
int launch_kkt_mixed_raw(const NewtonDev& nd, const double* r_dev, double beta, double* delta_dev, double* ws, int* n_fallback, hipStream_t s);
// packed variant: n_kkt rollouts from kkt_list[list_par], KKT_PACK per workgroup (dedicates whole CUs to the recursion)
// (pipe: 0 packed one-wave kernel, 1 three-wave pipeline, 2 twisted = two three-wave chains per rollout, 3 duo = two one-wave chains in one
//  workgroup, -1 by kkt_same_round)
int launch_kkt_packed(const NewtonDev& nd, int n_kkt, int list_par, hipStream_t s, const int* n_dev = nullptr, int pipe = -1);
bool kkt_twisted_available(const NewtonDev& nd);
bool kkt_duo_available(const NewtonDev& nd);      // ... as one workgroup of two one-wave chains (pipe = 3): next to the sweep
int launch_queue_recycle(const IpQueues& Q, int par, hipStream_t s);
int launch_solve_finish(const NewtonDev& nd, double* out, hipStream_t s);   // end-of-solve result block, see solve_finish_kernel
// reference-default backend (dense jacobian! + LU with partial pivoting), any mode / objective (kkt_dense.hip)
size_t kkt_dense_workspace_doubles(const NewtonDev& S, bool banded);
int launch_kkt_dense_newton(const NewtonDev& S, double* ws, hipStream_t s, bool banded);
struct KktArgs;
int launch_kkt_dense_args(const NewtonDev& S, const KktArgs& K, double* ws, hipStream_t s, bool banded);
// :configurationforce mode reduced to the :configuration solvers (newton_kernels.hip)
struct CfReduce { double* dzq; double* r2; double* d2; };   // packed q rows of the sensitivities, reduced rhs, reduced solution
NewtonDev cf_shadow(const NewtonDev& S);
size_t cf_reduce_doubles(const NewtonDev& S);
bool kkt_cf_reduce_available(const NewtonDev& S);
int launch_kkt_cf_reduced_newton(const NewtonDev& S, double* ws, double* dense_ws, hipStream_t s);
int launch_kkt_cf_reduced_raw(const NewtonDev& S, const double* r_dev, double beta, double* delta_dev, double* ws, double* dense_ws, hipStream_t s);
bool kkt_banded_available(const NewtonDev& S);
bool kkt_banded_twisted_available(const NewtonDev& S);
int launch_kkt_dense_raw(const NewtonDev& S, const double* r_dev, double beta, double* delta_dev, double* ws, hipStream_t s, bool banded);
// B1 seam: solve with caller-provided residual / beta for all rollouts, no state change
int launch_kkt_raw(const NewtonDev& nd, const double* r_dev, double beta, double* delta_dev,
                   hipStream_t s);

}  // namespace cimpc
