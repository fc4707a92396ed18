// Host runtime behind the C ABI of include/cimpc.h: handle, HBM residency, the per-knot
// linearization packer (A1), window bucketing, and the lock-step driver of newton_solve!
// (/root/reference/src/controller/newton.jl:169-288).  HIP runtime only - no torch, no BLAS.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cimpc_internal.h"
#include "lin_table.h"
#include "newton_state.h"

using namespace cimpc;

namespace {

thread_local std::string g_create_error;

enum ProfClass { PC_IP = 0, PC_KKT = 1, PC_RESID = 2, PC_OTHER = 3, PC_ASYNC = 4, PC_COUNT = 5 };

struct ProfRec {
    hipEvent_t a, b;
    int cls;
};

}  // namespace

// Streams of the lock-step rounds: the sweep / residual chain runs on `st`, the KKT recursion of the
// rollouts that start a Newton iteration on `st_kkt` next to it (joined by an event before the residual).
struct RoundStreams {
    hipStream_t st = nullptr;
    hipStream_t st_kkt = nullptr;
    hipEvent_t ev_join = nullptr;   // end of the round's KKT kernel
    hipEvent_t ev_start = nullptr;  // the caller's stream at the start of a solve (uploads of q0 / q1 precede the reset kernel)
};

// Schedule settings.  The defaults are the measured optima.  A dozen of them keep an environment
// override for the A/B scripts under scripts/ (read ONCE, in cimpc_create: a handle's configuration never changes
// afterwards and no solve-path call touches the environment); the rest are constants of the build - their overrides were
// removed in round 3, every one an untested configuration (the experiments they served are recorded in DESIGN.md).
struct Knobs {
    // ---- with an override ----
    int async_mode = 2;          // CIMPC_ASYNC: 0 lock-step rounds only, 1 always the single launch, 2 auto
    int async_tail = -1;         // CIMPC_ASYNC_TAIL: hybrid hand-over threshold (-1: by batch size)
    int spec_first = -1;         // CIMPC_SPEC_FIRST: step lengths of the first line-search round of a solve's first Newton iteration (-1: by batch size)
    int iter_cap = 28;           // CIMPC_ITER_CAP (B = 512: 20 / 24 / 28 / 32 -> 8.73 / 8.65 / 8.74 / 8.56 ms, inside the spread: profiles/r03/knobs3.log)
    bool async_debug = false;    // CIMPC_ASYNC_DEBUG
    double watchdog_s = 30.0;    // CIMPC_ASYNC_WATCHDOG_S
    bool debug_rounds = false;   // CIMPC_DEBUG_ROUNDS
    int tail_div = 8;            // CIMPC_TAIL_DIV
    int drain_pct = 95;          // CIMPC_DRAIN_PCT: drain parking once this percentage of the sweep's workgroups has left (0 = off; B = 512: 0 / 75 / 90 / 95 / 97 -> 10.68 / 10.97 / 10.48 / 10.45 / 10.47 ms)
    int drain_min = 4;           // CIMPC_DRAIN_MIN: ... for solves that have had at least this many iterations in the launch
    int async_full_max = 32;     // CIMPC_ASYNC_FULL_MAX: largest batch solved by the single persistent launch alone (larger: hybrid).
                                 // Measured 128 -> 64 (round 3): B = 96 8.88 -> 8.11 ms, B = 128 9.72 -> 9.48 ms, B = 64 unchanged (6.9 ms); 64 -> 32 (round 6,
                                 // after the rounds' sweep and KKT stage got faster): B = 64 5.0-5.7 -> 4.35-4.4 ms, B = 48 4.54 -> 4.07 ms with the hand-over at 32
    bool generic_static = false; // CIMPC_GENERIC_STATIC: runtime-dimension sweep with the static queue partition of rounds 2-3 instead of the dynamic pull
    int kkt_pipe = -1;           // CIMPC_KKT_PIPE: three-wave pipelined KKT kernel 0 never, 1 always, -1 where the solve is on the critical path
    int kkt_twisted = -1;        // CIMPC_KKT_TWISTED: twisted (two-ended) condensed solve, two workgroups per rollout: 0 never, 1 wherever the
                                 // pipelined kernel would run, -1 = 1
    int kkt_tw_nb = 0;           // CIMPC_KKT_TW_NB: rows eliminated from the bottom (0: the default split)
    int kkt_duo = 1;             // CIMPC_KKT_DUO: 0 = packed one-wave kernel next to the sweep (rounds 2-5), 1 = the duo kernel where it applies
    int kkt_duo_hint = 20000;    // CIMPC_KKT_DUO_HINT: ... in rounds whose sweep has at most this many problems queued (a shorter launch than the packed recursion)
    int kkt_duo_max = 256;       // ... for at most this many systems per launch (one workgroup per CU at a time)
    bool lazy_dz = true;         // CIMPC_LAZY_DZ: 0 = the decision kernel copies the accepted sensitivities itself (rounds 3-5)
    int kkt_tw_spins = 0;        // bound of a chain's wait for its partner, in polls (0: 2^21); the tests force the time-out path through cimpc_debug_set_tw_spins
    int kkt_tw_max = 120;        // twisted kernel for at most this many rollouts per launch (two workgroups each must be resident together)
    int async_kkt_tw = 1;        // CIMPC_ASYNC_KKT_TW: the persistent kernel's KKT stage as TWO cooperating jobs (the chains of the twisted solve): 0 never,
                                 // 1 where it was measured a gain - single launches of at most 32 rollouts (quadruped H = 40, B = 4 / 16 / 32: 2.08 -> 1.85,
                                 // 2.43 -> 2.27, 3.55 -> 3.41 ms; B = 64: 4.4 -> 4.9 ms - twice the jobs when every rollout reaches its KKT stage at once;
                                 // hybrid tail of B = 512: no difference in six alternating pairs, the tail's chain is its interior-point evaluations) and hybrid
                                 // tails of at most 32 rollouts (B = 48 / 64 / 96 / 128: -1.5 / -3 / -1 / 0 % in alternating pairs) -, 2 always
    // ---- constants ----
    int async_mem = 0;           // exchange buffers: ordinary device memory (uncached / fine-grained variants lost)
    int spec_all = -1;           // speculative slots of later line-search rounds: by batch size
    int spec_tail = 3;
    int spec_mid = -1;           // previous search depth from which a rollout starts with three candidates: by batch size
    int waves = 0;               // sweep workgroup size: by batch size
    int kkt_overlap = -1;        // KKT on its own stream next to the sweep: from 64 rollouts on
    int waves32 = 0;             // CIMPC_WAVES32: waves per sweep workgroup of the 32-lane models (0: by batch size; 4 = latency build, 8 = throughput build)
    int banded_form = 0;         // CIMPC_BANDED_FORM: banded LDL^T variants the sizes in the tree never reach - bit 0: four pivots per block, bit 1: window
                                 // of w + RB slots instead of the next power of two, bit 2: controls not eliminated (the form a singular R_t
                                 // falls back to) - tests/test_gpu_round4.py runs every form against form 0
    int sweep_wgs = 0;           // CIMPC_SWEEP_WGS: persistent sweep workgroups (0: computed from the resident set) - sub-batch experiments
    int async_service = 0;       // job-only workgroups of the asynchronous kernel: computed
    int async_flags = 0;         // reserved
    int async_sleep = 2;         // idle back-off of the asynchronous kernel (units of ~2 us), polls before looking around, wake-up fan
    int async_spins = 48;
    int async_fan = 1;
    bool kkt_packed = true;      // two rollouts per KKT workgroup next to the sweep
    bool kkt_scalar = false;     // scalar KKT kernel only where the MFMA one does not apply (tiles beyond 24, long horizons)
    int async_tail_grid = -1;    // workgroups of the hybrid tail's persistent kernel: 3 per rollout handed over (B = 512: 512 / 320 / 256 / 192 / 96
                                 // workgroups -> 10.85 / 10.42 / 10.35 / 10.38 / 10.6 ms)
    int kkt_pipe_max = 128;      // pipelined KKT kernel for at most this many rollouts per round (it takes three times the waves)

    static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
    void read_environment() {
        async_mode = env_int("CIMPC_ASYNC", async_mode);
        async_tail = env_int("CIMPC_ASYNC_TAIL", async_tail);
        spec_first = env_int("CIMPC_SPEC_FIRST", spec_first);
        iter_cap = std::max(1, env_int("CIMPC_ITER_CAP", iter_cap));
        async_debug = getenv("CIMPC_ASYNC_DEBUG") != nullptr;
        if (const char* v = getenv("CIMPC_ASYNC_WATCHDOG_S")) watchdog_s = atof(v);
        debug_rounds = getenv("CIMPC_DEBUG_ROUNDS") != nullptr;
        tail_div = env_int("CIMPC_TAIL_DIV", tail_div);
        async_full_max = env_int("CIMPC_ASYNC_FULL_MAX", async_full_max);
        drain_pct = env_int("CIMPC_DRAIN_PCT", drain_pct);
        drain_min = std::max(1, env_int("CIMPC_DRAIN_MIN", drain_min));
        kkt_pipe = env_int("CIMPC_KKT_PIPE", kkt_pipe);
        kkt_twisted = env_int("CIMPC_KKT_TWISTED", kkt_twisted);
        kkt_tw_nb = env_int("CIMPC_KKT_TW_NB", kkt_tw_nb);
        lazy_dz = env_int("CIMPC_LAZY_DZ", 1) != 0;
        kkt_duo = env_int("CIMPC_KKT_DUO", kkt_duo);
        kkt_duo_hint = env_int("CIMPC_KKT_DUO_HINT", kkt_duo_hint);
        async_kkt_tw = env_int("CIMPC_ASYNC_KKT_TW", async_kkt_tw);
        generic_static = env_int("CIMPC_GENERIC_STATIC", generic_static ? 1 : 0) != 0;
        sweep_wgs = env_int("CIMPC_SWEEP_WGS", sweep_wgs);
        waves32 = env_int("CIMPC_WAVES32", waves32);
        banded_form = env_int("CIMPC_BANDED_FORM", banded_form);
    }
};

struct cimpc_ctx {
    cimpc_dims dm{};
    Knobs kn{};
    cimpc_ip_opts ip{};
    cimpc_newton_opts nt{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    KernelInfo ki{};
    int nz = 0, nth = 0, nths = 0, nd = 0, nr = 0, nx = 0, ny = 0, N = 0;
    // device memory
    std::vector<void*> allocs;
    double* d_tab = nullptr;
    IpQueues Q{};                // device work queues (par is set per launch)
    int* d_window = nullptr;
    // cimpc_set_gait: the controller's full reference trajectory + per-rollout step counters
    double *g_q = nullptr, *g_u = nullptr, *g_w = nullptr, *g_g = nullptr, *g_b = nullptr, *g_th = nullptr, *g_stride = nullptr;
    int* g_phase = nullptr;
    bool gait_set = false;
    int dz_producer = 0;            // who wrote the per-step sensitivity memory last: 0 nobody, 1 newton_solve (dz_good), 2 implicit_dynamics (slot 0)
    int* d_good_src = nullptr;      // [B][H] NewtonDev::good_src (in use where the KKT stage is the condensed MFMA solve: select_kkt_backend)
    double* d_dz_knot = nullptr;    // [B][H_ref][nths*nd] per-knot archive, allocated at the first window change that needs it
    int wpk = 1;                 // persistent workgroups of a sweep launch
    double* d_alt = nullptr;
    double* d_zout = nullptr;
    double *d_Q = nullptr, *d_R = nullptr, *d_Qinv = nullptr, *d_Rinv = nullptr, *d_Cg = nullptr,
           *d_Cb = nullptr;
    double *d_q0 = nullptr, *d_q1 = nullptr;      // one allocation: d_q1 = d_q0 + B nq (uploaded by one copy)
    double* h_qin = nullptr;     // pinned staging of [q0 | q1], then nq doubles for the stride of cimpc_mpc_advance
    std::vector<int> win_mirror; // 0-based window as the HOST last uploaded it (empty: unknown - cimpc_set_gait / cimpc_mpc_advance moved it on the device)
    double* h_stage = nullptr;   // pinned staging of cimpc_mpc_solve (window, reference, altitude in; trajectory out), allocated at its first call
    size_t h_stage_doubles = 0;
    bool advance_pending = false; // a cimpc_mpc_advance was queued on `stream` and not waited for (its staging word is in flight)
    double* d_result = nullptr;  // end-of-solve result block (solve_finish_kernel): 8 + B (nu + 2) doubles
    double* h_result = nullptr;  // ... its pinned host copy, valid from the end of a solve to the next call that touches the state
    double* d_rhs = nullptr;   // B1 seam staging
    double* d_pstate = nullptr;   // parked interior-point iterates
    long long ip_budget_ticks = 0;  // cimpc_ip_opts::max_time in ticks of the device's constant-rate clock (0 = unlimited)
    int iter_cap = 28;            // (Knobs::iter_cap) measured B = 512: 16 / 20 / 24 / 32 / 48 -> 13.57 / 12.73 / 12.85 / 12.97 / 14.41 ms per batch step
    NewtonDev S{};
    int* h_counters = nullptr;   // pinned
    // bookkeeping
    std::vector<char> knot_set;
    int n_knots_set = 0;
    bool objective_set = false, window_set = false, reference_set = false, alt_set = false;
    bool velocity_objective = false;
    bool use_dense = false;        // KKT through kkt_dense.hip: the reference-default dense LU (any mode / objective) ...
    bool use_banded = false;       // ... or its banded LDL^T (:configuration mode, velocity objective / on request)
    int kkt_tw_overlap_max = 0;    // overlapped rounds: twisted KKT kernel when at most this many rollouts need a solve (0: never)
    long long adapt32 = 0;         // 32-lane models: problems per sweep launch from which the throughput build is taken (0: the handle's one build)
    long long n_kkt_twisted = 0;   // KKT launches that took the twisted kernel (cimpc_get_kkt_twisted)
    int* h_twfail = nullptr;       // host-mapped: hand-overs of the twisted kernels that timed out (NewtonDev::kkt_tw_fail), never reset
    int twfail_seen = 0;           // ... its value when the host last looked
    long long n_kkt_tw_fallbacks = 0;   // KKT stages repeated on the one-ended kernels after such a time-out (cimpc_get_kkt_twisted_fallbacks)
    bool band_reduce_ok = false;   // every R_t could be inverted: the banded LDL^T may eliminate the controls first (NewtonDev::band_reduce)
    bool use_mixed = false;        // condensed solve in mixed precision (CIMPC_KKT_CONDENSED_MIXED)
    double* d_mix_ws = nullptr;
    int* d_mix_nfb = nullptr;
    size_t dense_ws_doubles = 0;
    // :configurationforce with negligible gamma / b weights: eliminated onto the :configuration solvers (newton_kernels.hip)
    bool cf_reduce = false, cf_tiny = true;
    double* d_cf_ws = nullptr;
    double* d_dense_ws = nullptr;  // [B][N*N + 2N], allocated on first use
    double *d_V = nullptr, *d_qt = nullptr, *d_vt = nullptr;
    int waves = 4;
    bool kkt_overlap = true;
    int* d_ring = nullptr;       // [MAX_DEPTH][8] device counters per in-flight round (behind the queue counters: Q.count .. +ctl_ints)
    size_t ctl_ints = 0;
    int* h_ring = nullptr;       // pinned, host-mapped: {n_sweep, n_kkt, stamp}
    int* h_ring_dev = nullptr;   // device pointer of h_ring
    // asynchronous single-launch solve (newton_async_impl.h)
    bool async_on = false, async_dirty = true;   // async_on: buffers allocated, kernel available
    int async_mode = 2;          // 0 lock-step only, 1 always the single launch, 2 auto (by batch size / hybrid tail)
    int async_tail = 96;         // hybrid: hand over to the asynchronous kernel when at most this many rollouts are active
    int* a_items = nullptr;      // [K][a_cap] live interior-point queues
    int* a_jobs = nullptr;       // residual job entries, then KKT job entries
    int* a_ctrl = nullptr;       // [count K][head K][rq_head rq_tail kq_head kq_tail n_done ...]
    int* a_evals = nullptr;      // [B]
    long long* a_dbg = nullptr;  // [16] diagnostics (CIMPC_ASYNC_DEBUG)
    size_t a_cap = 0, a_rq_cap = 0, a_kq_cap = 0;
    int a_grid = 0, a_service = 0;
    RoundStreams rs;
    bool external_stream = false;
    std::vector<double> h_tab;       // one knot staging
    cimpc_stats last_stats{};
    // profiling
    int prof_on = 0;             // 0 off, 1 every launch, 2 the interior-point sweep launches only (cimpc_profile_enable)
    bool prof_open = false;      // the last prof_begin recorded an event pair (prof_end closes it)
    std::vector<ProfRec> prof_recs;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms[PC_COUNT] = {0, 0, 0, 0, 0};
    long long prof_n[PC_COUNT] = {0, 0, 0, 0, 0};
    long long prof_async_problems = 0;
    long long prof_ip_problems = 0, prof_kkt_systems = 0;
};

namespace {

int fail(cimpc_ctx* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    else g_create_error = msg;
    return code;
}

#define HIP_TRY(h, call)                                                                     \
    do {                                                                                     \
        hipError_t e__ = (call);                                                             \
        if (e__ != hipSuccess)                                                               \
            return fail(h, CIMPC_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

template <class T>
int dev_alloc(cimpc_ctx* h, T** p, size_t count, int mem = 0) {
    // mem: 0 = ordinary device memory; 1 = uncached, 2 = fine-grained device memory (experiments only)
    void* v = nullptr;
    if (count == 0) count = 1;
    hipError_t e = mem == 0 ? hipMalloc(&v, count * sizeof(T))
                            : hipExtMallocWithFlags(&v, count * sizeof(T), mem == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
    if (e != hipSuccess) return fail(h, CIMPC_ERR_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    e = hipMemset(v, 0, count * sizeof(T));
    // the fill runs on the NULL stream, which the library's non-blocking streams do not wait for: a buffer
    // allocated lazily right before a launch (dense KKT workspace) must be clean before anything touches it
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) return fail(h, CIMPC_ERR_HIP, std::string("hipMemset: ") + hipGetErrorString(e));
    h->allocs.push_back(v);
    *p = static_cast<T*>(v);
    return CIMPC_OK;
}

// dense inverse by Gauss-Jordan with partial pivoting (column-major n x n); false if singular
bool invert(const double* A, double* Ai, int n) {
    std::vector<double> M((size_t)n * 2 * n);
    auto at = [&](int r, int c) -> double& { return M[(size_t)r * 2 * n + c]; };
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
            at(r, c) = A[r + (size_t)c * n];
            at(r, n + c) = (r == c) ? 1.0 : 0.0;
        }
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int r = k + 1; r < n; ++r)
            if (std::fabs(at(r, k)) > std::fabs(at(p, k))) p = r;
        if (at(p, k) == 0.0 || !std::isfinite(at(p, k))) return false;
        if (p != k)
            for (int c = 0; c < 2 * n; ++c) std::swap(at(p, c), at(k, c));
        const double piv = at(k, k);
        for (int c = 0; c < 2 * n; ++c) at(k, c) /= piv;
        for (int r = 0; r < n; ++r) {
            if (r == k) continue;
            const double f = at(r, k);
            if (f == 0.0) continue;
            for (int c = 0; c < 2 * n; ++c) at(r, c) -= f * at(k, c);
        }
    }
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) Ai[r + (size_t)c * n] = at(r, n + c);
    return true;
}

void prof_begin(cimpc_ctx* h, int cls, hipStream_t st = nullptr) {
    h->prof_open = false;
    if (!h->prof_on || (h->prof_on == 2 && cls != PC_IP)) return;
    h->prof_open = true;
    if (!st) st = h->stream;
    ProfRec r;
    auto get = [&]() {
        if (!h->ev_pool.empty()) {
            hipEvent_t e = h->ev_pool.back();
            h->ev_pool.pop_back();
            return e;
        }
        hipEvent_t e;
        (void)hipEventCreate(&e);
        return e;
    };
    r.a = get();
    r.b = get();
    r.cls = cls;
    (void)hipEventRecord(r.a, st);
    h->prof_recs.push_back(r);
}
void prof_end(cimpc_ctx* h, hipStream_t st = nullptr) {
    if (!h->prof_on || !h->prof_open) return;
    h->prof_open = false;
    (void)hipEventRecord(h->prof_recs.back().b, st ? st : h->stream);
}
void prof_collect(cimpc_ctx* h) {
    if (h->prof_recs.empty()) return;
    (void)hipStreamSynchronize(h->stream);
    if (h->rs.st) (void)hipStreamSynchronize(h->rs.st);
    if (h->rs.st_kkt) (void)hipStreamSynchronize(h->rs.st_kkt);
    for (auto& r : h->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            h->prof_ms[r.cls] += ms;
            h->prof_n[r.cls] += 1;
        }
        h->ev_pool.push_back(r.a);
        h->ev_pool.push_back(r.b);
    }
    h->prof_recs.clear();
}

IpParams make_ip_params(cimpc_ctx* h, const TrajDev& T, int par, int* pending_counter, double* zout) {
    IpParams p{};
    p.tab = h->d_tab;
    p.Q = h->Q;
    p.Q.par = par;
    p.wpk = h->wpk;
    p.q = T.q;
    p.theta = T.th;
    p.gam = T.g;
    p.bfr = T.b;
    p.alt = h->alt_set ? h->d_alt : nullptr;
    p.d = h->S.d;
    p.dz = h->S.dz;
    p.nu = nullptr; p.dtn = nullptr; p.dtn_ld = h->S.dtn_ld;      // (the Newton loop switches the products on, see cimpc_newton_solve_dev)
    p.status = h->S.ip_status;
    p.iters = h->S.ip_iters;
    p.zout = zout;
    p.pflag = h->S.pflag;
    p.pstate = h->d_pstate;
    p.pending_count = pending_counter;
    p.iter_cap = h->iter_cap;
    p.slots = CS;
    p.H = h->dm.H;
    p.o = h->ip;
    p.kc_floor = h->ip.kappa_tol / h->ip.undercut;
    p.tau_floor = 1.0 - h->ip.eps_min;
    p.reg_floor = h->ip.kappa_tol * h->ip.gamma_reg;
    p.budget_ticks = h->ip_budget_ticks;
    p.generic_static = h->kn.generic_static ? 1 : 0;
    return p;
}

// queue kernel + sensitivity kernel of one round
int run_sweep(cimpc_ctx* h, int par, int* pending_counter, double* zout, hipStream_t st, int iter_cap = 0, int* drain_counter = nullptr,
              bool with_products = false, long long hint = -1) {
    IpParams p = make_ip_params(h, h->S.cand, par, pending_counter, zout);
    int waves = h->waves;
    // 32-lane models: the build of the sweep is chosen PER LAUNCH from the problems the host knows to be queued (round 5; the choice
    // used to be made once per handle from B H): the latency build (4 waves per workgroup) below `adapt32` problems, the throughput
    // build (8 waves, two per SIMD) from there on.  Parked iterates are the model's, not the build's: a solve may change builds.
    // (only where the host's count is complete: with the KKT stage in the same round as the evaluation of its candidates - small
    //  batches - the requests of that round are not known at launch; a first version sized a B = 1 sweep for ONE workgroup: 1.1 -> 20 ms)
    if (h->ki.G == 32 && h->adapt32 > 0 && hint > 0 && h->kkt_overlap && h->kn.sweep_wgs <= 0 && !(h->kn.waves32 == 4 || h->kn.waves32 == 8)) {      // (an explicit CIMPC_SWEEP_WGS keeps its grid)
        waves = hint >= h->adapt32 ? 8 : 4;
        const size_t groups_per_wg = 2 * (size_t)waves;
        size_t w = ((size_t)hint + 2 * groups_per_wg - 1) / (2 * groups_per_wg);
        w = std::max<size_t>(w, std::min<size_t>((size_t)h->dm.H_ref, (size_t)std::max<long long>(hint, 1)));
        p.wpk = (int)std::max<size_t>(1, std::min<size_t>(w, 256));
    }
    // (round 4, measured and removed: one sweep workgroup per CU in rounds with few problems - <= 4 k / 8 k / 16 k - so that every
    //  wave has its SIMD to itself: 7.97 -> 7.96 / 7.99 / 8.01 ms per step, no effect: profiles/r04/knob_small_round.log)
    if (with_products && h->S.dtn != nullptr) { p.nu = h->S.nu_cand; p.dtn = h->S.dtn; }
    // single rollouts (B < 4: at most 7 B problems per knot): one workgroup per knot, no remaining-work scans (IpParams::direct)
    if (h->dm.B < 4 && h->kn.sweep_wgs <= 0) { p.direct = 1; p.wpk = h->dm.H_ref; }
    if (iter_cap > 0) p.iter_cap = iter_cap;
    if (drain_counter != nullptr && h->kn.drain_pct > 0 && p.iter_cap < h->ip.max_iter) {
        p.drain_count = drain_counter;
        p.drain_thresh = std::max(1, (int)((long long)p.wpk * h->kn.drain_pct / 100));
        p.drain_min = h->kn.drain_min;
    }
    prof_begin(h, PC_IP, st);
    int rc = launch_ip_sweep(&h->dm, p, waves, st);
    prof_end(h, st);
    if (rc != CIMPC_OK) return fail(h, rc, "ip sweep launch failed");
    return CIMPC_OK;
}

int ensure_dense_ws(cimpc_ctx* h) {
    if (h->cf_reduce && !h->d_cf_ws && dev_alloc(h, &h->d_cf_ws, cf_reduce_doubles(h->S)) != CIMPC_OK) return CIMPC_ERR_HIP;
    if (h->cf_reduce && h->S.V == nullptr) return CIMPC_OK;            // reduced problem goes to the condensed MFMA solve
    const size_t need = h->cf_reduce ? kkt_dense_workspace_doubles(cf_shadow(h->S), true) : kkt_dense_workspace_doubles(h->S, h->use_banded);
    if (h->d_dense_ws && h->dense_ws_doubles >= need) return CIMPC_OK;
    h->d_dense_ws = nullptr;                      // (an earlier, smaller workspace stays in h->allocs until destroy)
    h->dense_ws_doubles = need;
    return dev_alloc(h, &h->d_dense_ws, need);
}

// which kkt_dense.hip path serves the current mode / objective / requested backend
static void select_kkt_backend(cimpc_ctx* h) {
    const bool cfg = h->dm.mode == CIMPC_MODE_CONFIGURATION;
    const int want = h->nt.kkt_backend;
    const bool velocity = h->S.V != nullptr;
    // (dimension sets without a compiled condensed solve - the runtime-dimension path - take the banded LDL^T / dense LU)
    h->use_dense = !cfg || want == CIMPC_KKT_DENSE_LU || want == CIMPC_KKT_BANDED_LDL || velocity || !kkt_condensed_available(h->S);
    // Form of the banded LDL^T.  The reduced form (controls eliminated first) needs LDS for its pre-pass as well as the
    // window: where that does not fit (long horizons of wide models) the full interleaved form is tried, and the dense LU
    // serves what neither fits.  avail() = the availability test of whoever would run the banded kernel on h->S.
    auto pick_band_form = [&](auto avail) {
        h->S.band_reduce = h->band_reduce_ok ? 1 : 0;
        if (avail()) return true;
        if (h->S.band_reduce == 0) return false;
        h->S.band_reduce = 0;
        if (avail()) return true;
        h->S.band_reduce = 1;
        return false;
    };
    h->S.band_reduce = h->band_reduce_ok ? 1 : 0;
    h->use_banded = h->use_dense && cfg && want != CIMPC_KKT_DENSE_LU && pick_band_form([&] { return kkt_banded_available(h->S); });
    h->cf_reduce = !cfg && want != CIMPC_KKT_DENSE_LU && h->cf_tiny &&
                   (velocity ? pick_band_form([&] { return kkt_cf_reduce_available(h->S); }) : kkt_cf_reduce_available(h->S));
    h->use_mixed = want == CIMPC_KKT_CONDENSED_MIXED && !h->use_dense && kkt_mixed_available(h->S);
    // lazy commit of the accepted sensitivities (NewtonDev::good_src): where every KKT stage is the condensed fp64 MFMA solve
    // (CIMPC_LAZY_DZ=0 keeps the copy at the accept, for the A/B lines)
    h->S.good_src = (!h->use_dense && !h->use_mixed && !h->cf_reduce && h->kn.lazy_dz && kkt_lazy_commit_available(h->S)) ? h->d_good_src : nullptr;
}

int ensure_mixed_ws(cimpc_ctx* h) {
    if (!h->use_mixed || h->d_mix_ws) return CIMPC_OK;
    if (dev_alloc(h, &h->d_mix_ws, kkt_mixed_workspace_doubles(h->S)) != CIMPC_OK) return CIMPC_ERR_HIP;
    return dev_alloc(h, &h->d_mix_nfb, 1);
}

// KKT stage of the Newton loop for everything that is not the plain condensed solve
static int launch_kkt_general(cimpc_ctx* h, const NewtonDev& Sk, hipStream_t st) {
    if (h->use_mixed) return launch_kkt_mixed_newton(Sk, h->d_mix_ws, h->d_mix_nfb, st);
    if (h->cf_reduce) {
        if (h->velocity_objective && kkt_banded_twisted_available(cf_shadow(Sk))) h->n_kkt_twisted++;
        return launch_kkt_cf_reduced_newton(Sk, h->d_cf_ws, h->d_dense_ws, st);
    }
    if (h->use_banded && kkt_banded_twisted_available(Sk)) h->n_kkt_twisted++;      // (launch_kkt_dense's own test)
    return launch_kkt_dense_newton(Sk, h->d_dense_ws, st, h->use_banded);
}

// statistics of the running / last solve: the per-rollout records of the decision stage, summed (NewtonDev::stats)
int read_stats(cimpc_ctx* h, long long out[4]) {
    std::vector<long long> v((size_t)h->dm.B * 4);
    HIP_TRY(h, hipMemcpy(v.data(), h->S.stats, v.size() * sizeof(long long), hipMemcpyDeviceToHost));
    out[0] = out[1] = out[2] = out[3] = 0;
    for (size_t b = 0; b < (size_t)h->dm.B; ++b)
        for (int k = 0; k < 4; ++k) out[k] += v[b * 4 + k];
    return CIMPC_OK;
}

// cimpc_mpc_advance returns with its copy / kernels still queued on h->stream (a private NON-blocking stream): every entry point that
// rewrites reference / gait / window / objective state with blocking NULL-stream copies must first let that work finish, or the
// pending gait_window / rekey kernels race with the rewrite
int drain_pending(cimpc_ctx* h) {
    if (!h->advance_pending) return CIMPC_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->advance_pending = false;
    return CIMPC_OK;
}

int check_ready(cimpc_ctx* h, bool need_newton) {
    if (!h) return CIMPC_ERR_INVALID;
    if (h->n_knots_set != h->dm.H_ref)
        return fail(h, CIMPC_ERR_STATE, "set_linearization has not been called for every knot");
    if (!h->window_set) return fail(h, CIMPC_ERR_STATE, "set_window has not been called");
    if (need_newton) {
        if (!h->objective_set) return fail(h, CIMPC_ERR_STATE, "set_objective has not been called");
        if (!h->reference_set) return fail(h, CIMPC_ERR_STATE, "set_reference has not been called");
    }
    return CIMPC_OK;
}

}  // namespace

extern "C" {

int cimpc_version(void) { return 106; }      // 1.06: round 6 (+ cimpc_mpc_solve; no struct changed).  1.05: round 5 (+ cimpc_get_kkt_twisted; no struct changed).  1.04: round 4 (cimpc_ip_opts::max_time - the struct grew by one double at its end)

void cimpc_default_ip_opts(cimpc_ip_opts* o) {
    if (!o) return;
    o->r_tol = 1.0e-8;
    o->kappa_tol = 2.0e-4;
    o->undercut = 5.0;
    o->gamma_reg = 0.1;
    o->kappa_reg = 1.0e-3;
    o->eps_min = 0.05;
    o->ls_scale = 0.5;
    o->max_iter = 100;
    o->max_ls = 3;
    o->stall_alpha = 1.0e-13;
    o->max_time = 0.0;       // unlimited (the reference's default, 1e5 s, is treated as unlimited too)
}

void cimpc_default_newton_opts(cimpc_newton_opts* o) {
    if (!o) return;
    o->r_tol = 3.0e-4;
    o->beta_init = 1.0e-5;
    o->max_time = 0.0;
    o->kappa = 2.0e-4;
    o->max_iter = 5;
    o->kkt_backend = CIMPC_KKT_CONDENSED;
}

const char* cimpc_last_error(cimpc_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int cimpc_create(const cimpc_dims* dims, const cimpc_ip_opts* ip, const cimpc_newton_opts* nt,
                 int device, cimpc_handle* out) {
    if (!dims || !out) return fail(nullptr, CIMPC_ERR_INVALID, "null argument");
    *out = nullptr;
    const cimpc_dims& d = *dims;
    if (d.nq <= 0 || d.nu < 0 || d.nw < 0 || d.nc <= 0 || d.nb < 0 || d.H <= 0 || d.H_ref <= 0 ||
        d.B <= 0 || (d.mode != 0 && d.mode != 1))
        return fail(nullptr, CIMPC_ERR_INVALID, "invalid dimensions");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, CIMPC_ERR_NO_DEVICE,
                    "no HIP device visible: this library has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(nullptr, CIMPC_ERR_NO_DEVICE, "device index out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess)
        return fail(nullptr, CIMPC_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, CIMPC_ERR_NO_DEVICE,
                    std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, CIMPC_ERR_NO_DEVICE, "hipSetDevice failed");

    cimpc_ctx* h = new cimpc_ctx();
    h->dm = d;
    if (ip) h->ip = *ip; else cimpc_default_ip_opts(&h->ip);
    if (nt) h->nt = *nt; else cimpc_default_newton_opts(&h->nt);
    h->device = device;
    if (h->ip.max_time > 0.0 && h->ip.max_time < 1000.0) {      // per-solve budget of the interior point (policy.jl:9,61)
        if (h->ip.max_iter >= 128) { delete h; return fail(nullptr, CIMPC_ERR_INVALID, "ip max_time needs max_iter < 128 (the parked state packs both into one word)"); }
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) != hipSuccess || khz <= 0) {
            delete h;      // no guessing: a budget in ticks of an unknown clock would silently be the wrong budget
            return fail(nullptr, CIMPC_ERR_HIP, "ip max_time: the device does not report hipDeviceAttributeWallClockRate");
        }
        h->ip_budget_ticks = std::max<long long>(1, (long long)std::llround(h->ip.max_time * 1.0e3 * (double)khz));
    }
    h->kn.read_environment();       // the only place the environment is consulted
    h->iter_cap = h->kn.iter_cap;
    if (ip_kernel_info(&h->dm, &h->ki) != CIMPC_OK) {
        delete h;
        return fail(nullptr, CIMPC_ERR_INVALID,
                    "model dimensions outside the runtime-dimension kernel (nx, ny <= 64) and without a compiled set (built: pushbot, hopper_2D, hopper_3D, walledcartpole, particle, particle_2D, "
                    "quadruped, flamingo, centroidal_quadruped are built)");
    }
    h->nx = d.nq;
    h->ny = 2 * d.nc + d.nb;
    h->nz = d.nq + 4 * d.nc + 2 * d.nb;
    h->nth = 2 * d.nq + d.nu + d.nw + 2;
    h->nths = 2 * d.nq + d.nu;
    h->nd = d.mode ? d.nq + d.nc + d.nb : d.nq;
    h->nr = d.mode ? d.nq + d.nu + d.nc + d.nb : d.nq + d.nu;
    h->N = d.H * (h->nr + h->nd);
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return fail(nullptr, CIMPC_ERR_HIP, "hipStreamCreate failed");
    }
    h->own_stream = true;
    h->knot_set.assign(d.H_ref, 0);
    h->h_tab.assign(h->ki.tab_size, 0.0);

    const size_t B = d.B, H = d.H;
    int rc = CIMPC_OK;
    {   // asynchronous single-launch solve? (queues sized for every push of one solve: they never wrap)
        // CIMPC_ASYNC: 0 = lock-step rounds only, 1 = always the single launch, unset / 2 = auto.  Measured on
        // MI355X (quadruped, H = 40): the single launch wins for 4 <= B <= 128 rollouts
        // (B = 64: 8.1 vs 11.2 ms), the rounds win for large batches (B = 512: 16.5 vs 21.7 ms) - except for
        // their sparse tail, which auto mode hands over to the asynchronous kernel.
        h->async_mode = h->kn.async_mode;
        h->async_tail = std::min(96, std::max(64, d.B / 6));      // measured: B = 512 -> 48 .. 96 flat; B = 1024 / 2048 -> 96 (11.1 / 18.85 ms against 11.4 / 19.5 at 170 / 256:
                                                                   // the persistent kernel's chains do not get shorter with more rollouts handed over, the rounds do)
        // (round 6, mid-size batches: B = 48 / 64 / 96 best at 32 - 4.07 / 4.35 / 4.41 ms -, B = 128 at 32-48 - 4.91 -, B = 192 at 48 - 4.76 ms against 5.30
        //  with the rule above: scripts/dbg/headline_b.sh)
        if (d.B < 256) h->async_tail = std::max(32, d.B / 4);
        if (h->kn.async_tail >= 0) h->async_tail = h->kn.async_tail;
        const bool want = h->async_mode != 0;
        const size_t K = d.H_ref;
        const size_t evals = 1 + 7 * (size_t)std::max(1, h->nt.max_iter);       // per rollout and solve
        h->a_cap = B * evals * ((H + K - 1) / K + 1);
        h->a_rq_cap = B * (2 + 3 * (size_t)std::max(1, h->nt.max_iter));
        h->a_kq_cap = B * (2 + 3 * (size_t)std::max(1, h->nt.max_iter));      // (twisted KKT jobs: two entries per stage, one more after a timed-out hand-over)
        const size_t bytes = (K * h->a_cap + h->a_rq_cap + h->a_kq_cap) * sizeof(int);
        h->async_on = want && newton_async_available(&d) && K <= 256 && bytes <= ((size_t)1 << 30);
    }
    // 32-lane models (centroidal: one workgroup per CU, 512-register waves): the persistent kernel wins up to 32 rollouts only and the
    // rounds keep their lead far into the tail - measured on BASELINE configs[4] (H = 60): B = 8 / 16 / 32 / 64 / 128 -> single launch
    // 10.7 / 14.1 / 19.3 / 29.5 / - ms, lock-step rounds 12.1 / 17.7 / 20.5 / 22.4 / 34.9 ms, hand-over at 16 active rollouts 22.3 / 35.7 ms
    if (h->ki.G == 32) {
        if (getenv("CIMPC_ASYNC_FULL_MAX") == nullptr) h->kn.async_full_max = std::min(h->kn.async_full_max, 32);
        if (h->kn.async_tail < 0) h->async_tail = 16;
    }
    const int xm = h->async_on ? h->kn.async_mem : 0;   // experiments: 1 uncached, 2 fine-grained
    auto A = [&](auto** p, size_t n) { if (rc == CIMPC_OK) rc = dev_alloc(h, p, n); };
    auto AX = [&](auto** p, size_t n) { if (rc == CIMPC_OK) rc = dev_alloc(h, p, n, xm); };   // exchanged state
    A(&h->d_tab, (size_t)d.H_ref * h->ki.tab_size);
    const int ppw = 64 / h->ki.G;
    {   // work queues: a knot can appear ceil(H / H_ref) times in one window
        const size_t K = d.H_ref;
        const size_t cap = B * CS * ((H + K - 1) / K + 1);
        h->Q.K = (int)K; h->Q.cap = (int)cap; h->Q.par = 0;
        A(&h->Q.items, 2 * K * cap);
        // queue counters, queue heads and the round counters: ONE block, cleared by one memset at the start of a solve
        h->ctl_ints = 3 * K * QPAD + 2 * 8 * CPAD;
        A(&h->Q.count, h->ctl_ints);
        if (rc == CIMPC_OK) { h->Q.head = h->Q.count + 2 * K * QPAD; h->d_ring = h->Q.count + 3 * K * QPAD; }
        AX(&h->Q.done_count, B * CS);
        A(&h->d_window, B * (H + 2));
        h->Q.window = h->d_window;
    }
    A(&h->d_alt, B * d.nc);
    A(&h->d_zout, B * CS * H * h->nz);
    A(&h->d_Q, H * d.nq * d.nq);
    A(&h->d_R, H * d.nu * d.nu);
    A(&h->d_Qinv, H * d.nq * d.nq);
    A(&h->d_Rinv, H * d.nu * d.nu);
    A(&h->d_Cg, H * d.nc * d.nc);
    A(&h->d_Cb, H * d.nb * d.nb);
    A(&h->d_V, H * d.nq * d.nq);
    A(&h->d_qt, H * d.nq);
    A(&h->d_vt, H * d.nq);
    A(&h->d_q0, 2 * B * d.nq);
    if (rc == CIMPC_OK) h->d_q1 = h->d_q0 + B * d.nq;
    A(&h->d_result, 8 + B * (d.nu + 2));
    if (rc == CIMPC_OK && (hipHostMalloc((void**)&h->h_qin, (2 * B + 1) * d.nq * sizeof(double), hipHostMallocDefault) != hipSuccess ||
                           hipHostMalloc((void**)&h->h_result, (8 + B * (d.nu + 2)) * sizeof(double), hipHostMallocDefault) != hipSuccess))
        rc = CIMPC_ERR_HIP;
    A(&h->d_rhs, B * h->N);
    NewtonDev& S = h->S;
    S.dm = d;
    S.b0 = 0;
    S.nb_launch = d.B;
    S.nd = h->nd; S.nr = h->nr; S.nth = h->nth; S.nths = h->nths; S.N = h->N;
    const size_t BS = B * CS;     // evaluation slots (speculative line search)
    for (TrajDev* T : {&S.traj, &S.cand, &S.ref}) {
        const size_t n = (T == &S.cand) ? BS : B;
        if (T == &S.ref) {       // read-only during a solve
            A(&T->q, n * (H + 2) * d.nq); A(&T->u, n * H * d.nu); A(&T->w, n * H * d.nw);
            A(&T->g, n * H * d.nc); A(&T->b, n * H * d.nb); A(&T->th, n * H * h->nth);
        } else {
            AX(&T->q, n * (H + 2) * d.nq); AX(&T->u, n * H * d.nu); AX(&T->w, n * H * d.nw);
            AX(&T->g, n * H * d.nc); AX(&T->b, n * H * d.nb); AX(&T->th, n * H * h->nth);
        }
    }
    AX(&S.nu, B * H * h->nd);
    AX(&S.nu_cand, BS * H * h->nd);
    AX(&S.d, BS * H * h->nd);
    AX(&S.dz, BS * H * h->nths * h->nd);
    AX(&S.dz_good, B * H * h->nths * h->nd);
    AX(&h->d_good_src, B * H);
    S.dtn_ld = h->ki.generic ? 0 : h->ki.dtn_ld;
    if (S.dtn_ld > 0) AX(&S.dtn, BS * H * (size_t)S.dtn_ld);
    AX(&S.ip_status, BS * H);
    AX(&S.ip_iters, BS * H);
    AX(&S.pflag, BS * H);
    AX(&S.cur_slot, B);
    A(&h->d_pstate, BS * H * (2 * (size_t)h->nx + 4 * (size_t)h->ny + 4));
    AX(&S.res, B * h->N);
    AX(&S.res_cand, BS * h->N);
    AX(&S.delta, B * h->N);
    AX(&S.r_norm, B); AX(&S.r_cand, BS); AX(&S.alpha, B); AX(&S.beta, B);
    AX(&S.ls_iter, B); AX(&S.newton_l, B); AX(&S.stage, B); AX(&S.need_sweep, BS);
    A(&S.kkt_list, 2 * B);
    A(&S.slot_list, 2 * BS);
    AX(&S.counters, 8 * CPAD);
    AX(&S.stats, std::max<size_t>(B * 4, 64));      // (>= 32 entries: diagnostic builds with -DCIMPC_KKT_PROF park their phase clocks at [8..23])
    AX(&S.ro_sweeps, B); AX(&S.ro_ip_iters, B); AX(&S.ro_ip_fail, B);
    AX(&S.nlog, B * NLOG * 4);
    A(&S.kkt_ws, B * H * (3 * (size_t)h->nd * h->nd + h->nd));
    A(&S.kkt_tw_xch, B * (3 * (size_t)h->nd * h->nd + 4 * h->nd));      // (= kkt_tw_xch_doubles(nd))
    A(&S.kkt_tw_flags, B * 32);                                          // (= KKT_TW_FLAGS per rollout, one line each)
    (void)ppw;
    if (rc == CIMPC_OK && hipHostMalloc((void**)&h->h_counters, 8 * CPAD * sizeof(int)) != hipSuccess)
        rc = fail(h, CIMPC_ERR_HIP, "hipHostMalloc failed");
    if (rc != CIMPC_OK) {
        g_create_error = h->err;
        cimpc_destroy(h);
        return rc;
    }
    S.Q = h->d_Q; S.R = h->d_R; S.Qinv = h->d_Qinv; S.Rinv = h->d_Rinv; S.Cg = h->d_Cg; S.Cb = h->d_Cb;
    if (hipMemset(h->d_good_src, 0xFF, B * H * sizeof(int)) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {      // -1: every block is in dz_good
        g_create_error = "hipMemset failed"; cimpc_destroy(h); return CIMPC_ERR_HIP;
    }
    S.kkt_scalar = h->kn.kkt_scalar ? 1 : 0;      // (select_kkt_backend reads it)
    S.band_form = h->kn.banded_form;
    S.r_tol = h->nt.r_tol; S.beta_init = h->nt.beta_init; S.kappa = h->nt.kappa; S.max_iter = h->nt.max_iter;
    // all-seven-step-lengths speculation: shortens the chain of rollouts that exhaust their line search; pays
    // when the solve is latency-bound (small batches), costs throughput otherwise (B = 2048: -8 %)
    // condensed MFMA/scalar solve: :configuration + TrackingObjective; everything else (and kkt_backend = dense LU)
    // goes through the reference-default dense LU
    select_kkt_backend(h);
    S.spec_all = h->kn.spec_all >= 0 ? h->kn.spec_all : (d.B <= 128 ? 3 : 8);
    // first line search of a solve (no history yet): 1, 1/2, 1/4 evaluated together up to mid-size batches (B = 512: 9.65 -> 9.50 ms,
    // B = 128: 8.14 -> 8.03 ms for 0.3 more evaluated sweeps per step; B = 2048: 25.4 -> 25.6 ms, throughput-bound: one candidate there)
    S.spec_first = h->kn.spec_first >= 0 ? h->kn.spec_first : (d.B <= 1024 ? 3 : 1);
    S.kkt_scalar = h->kn.kkt_scalar ? 1 : 0;
    S.kkt_tw_nb = h->kn.kkt_tw_nb;
    S.kkt_tw_epoch = 0;
    S.kkt_tw_spins = h->kn.kkt_tw_spins > 0 ? h->kn.kkt_tw_spins : (1 << 21);
    {
        int* dv = nullptr;
        if (hipHostMalloc((void**)&h->h_twfail, 4 * sizeof(int), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void**)&dv, h->h_twfail, 0) != hipSuccess) {
            g_create_error = "mapped counter allocation failed"; cimpc_destroy(h); return CIMPC_ERR_HIP;
        }
        h->h_twfail[0] = 0;
        S.kkt_tw_fail = dv;
    }
    S.kkt_tw_raw = (h->kn.kkt_twisted != 0 && d.B <= h->kn.kkt_tw_max) ? 1 : 0;
    if (h->kn.kkt_duo == 2 && h->kn.kkt_twisted != 0) S.kkt_tw_raw = 2;      // CIMPC_KKT_DUO=2: the duo kernel at the B1 seam (tests)
    // Overlapped rounds (B >= 64, hybrid schedule): the twisted kernel where EVERY round's KKT set fits the pair bound, i.e. batches
    // of up to 120 rollouts - B = 96: 5.56 -> 4.8-5.1 ms per batch step; with a bound below the batch size the rounds mix kernels
    // and lose (B = 128: 6.04 -> 6.8 ms at 48 / 96; B = 256 and 512 within the noise: profiles/r05/ab_tw_overlap.log).
    // (CIMPC_KKT_TWISTED >= 2: the bound itself, for the A/B lines)
    h->kkt_tw_overlap_max = h->kn.kkt_twisted >= 2 ? h->kn.kkt_twisted : (h->kn.kkt_twisted != 0 && d.B <= h->kn.kkt_tw_max ? h->kn.kkt_tw_max : 0);
    S.kkt_tw_band = S.kkt_tw_raw;      // (the banded LDL^T is launched over all rollouts of the handle: the same bound applies)
    // large batches: a rollout whose previous search needed a back-off starts the next one with 1, 1/2, 1/4 together (one
    // round less per Newton iteration for 0.9 more evaluated sweeps per step: B = 512 11.4 -> 10.9 ms); small batches already
    // evaluate all seven step lengths from depth 3 on
    S.spec_mid = h->kn.spec_mid >= 0 ? h->kn.spec_mid : (d.B > 128 ? 1 : 8);
    // 4 waves share one staged table (throughput); measured for single rollouts too (one problem per knot anyway): B = 1
    // quadruped H = 40 cold 0.945 -> 0.926 ms, warm MPC loop 3.19 -> 3.03 ms, hopper H = 20 1.02 -> 0.97 ms against 1 wave
    h->waves = 4;
    // 32-lane models (round 4): from about three problems per lane group and sweep on, the throughput build of the sweep - eight waves per
    // workgroup, two per SIMD (ip_kernel_impl.h: ip_queue_kernel<M, WIDE>; centroidal H = 60: 64 rollouts 7.9 -> 11.6 ms of sweeps per
    // step, 128 rollouts 14.0 -> 12.2 ms, 256 rollouts 26.0 -> 22.4 ms - profiles/r04/cent_w8b.log).  CIMPC_WAVES32 = 4 / 8 forces one.
    // (A first form that ALSO read the MGS column twice instead of keeping it in registers was LDS-bound: cent_8wave_experiment.log.)
    if (h->ki.G == 32) {
        if (h->kn.waves32 > 0 && h->kn.waves32 < 1000) h->waves = h->kn.waves32;
        else if ((size_t)B * H >= 6000) h->waves = 8;
        // Round 5: the build is chosen PER LAUNCH (run_sweep) from the problems queued for it - a step of 64 rollouts has launches of
        // 3.8 k (one candidate per rollout) to 11.5 k problems (three): centroidal H = 60, 64 rollouts 7.87 -> 6.7 ms of sweeps per
        // step (10.7 -> 9.6 ms per batch step), 128 rollouts 12.0 -> 11.5 ms; thresholds 5 / 7 / 9 / 11 k are equivalent at 64,
        // 5 - 7 k best at 128 (profiles/r05/cent_adapt32.log).  CIMPC_WAVES32 >= 1000 sets the threshold, 4 / 8 force one build.
        h->adapt32 = h->kn.waves32 >= 1000 ? h->kn.waves32 : (h->kn.waves32 == 4 || h->kn.waves32 == 8) ? 0 : 7000;
    }
    // the single-launch solve runs its residual jobs on the whole workgroup: 4 waves also for small batches
    // (measured B = 8: 5.2 -> 4.4 ms, B = 64: 8.5 -> 7.5 ms)
    if (h->async_on && (h->async_mode == 1 || (h->async_mode == 2 && B >= 4 && B <= h->kn.async_full_max))) h->waves = 4;
    if (h->kn.waves == 1 || h->kn.waves == 2 || (h->kn.waves >= 4 && h->kn.waves <= 8)) h->waves = h->kn.waves;   // (5..8: builds with CIMPC_SWEEP_THREADS > 256)
    h->kkt_overlap = B >= 64;
    if (h->kn.kkt_overlap >= 0) h->kkt_overlap = h->kn.kkt_overlap != 0;
    if (hipHostMalloc((void**)&h->h_ring, 32 * sizeof(int), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void**)&h->h_ring_dev, h->h_ring, 0) != hipSuccess) {
        g_create_error = "ring allocation failed"; cimpc_destroy(h); return CIMPC_ERR_HIP;
    }
    {
        const size_t K = d.H_ref;
        if (h->async_on) {
            if (dev_alloc(h, &h->a_items, K * h->a_cap, xm) != CIMPC_OK || dev_alloc(h, &h->a_jobs, h->a_rq_cap + h->a_kq_cap, xm) != CIMPC_OK ||
                dev_alloc(h, &h->a_ctrl, 2 * K * QPAD + 64 + 33 * 16, xm) != CIMPC_OK || dev_alloc(h, &h->a_evals, B, xm) != CIMPC_OK) {
                g_create_error = h->err; cimpc_destroy(h); return CIMPC_ERR_HIP;
            }
            if (h->kn.async_debug && dev_alloc(h, &h->a_dbg, 16) != CIMPC_OK) { g_create_error = h->err; cimpc_destroy(h); return CIMPC_ERR_HIP; }
            h->async_dirty = true;
        }
    }
    {   // persistent workgroups of a sweep launch: all resident (256 VGPRs -> 8 waves per CU), about two
        // problems per lane group in a full round, and never fewer workgroups than busy knots (a
        // workgroup serves one knot at a time)
        const size_t groups_per_wg = (64 / h->ki.G) * h->waves;
        const size_t nprob = B * H;
        size_t w = (nprob + 2 * groups_per_wg - 1) / (2 * groups_per_wg);
        w = std::max<size_t>(w, std::min<size_t>(d.H_ref, nprob));
        // (32-lane models run one wave per SIMD - 512 registers - i.e. 4 waves per CU; the asynchronous launcher
        //  clamps its grid to the occupancy the runtime reports in any case)
        const size_t resident = (size_t)256 * std::max(1, (h->ki.G == 16 ? (h->waves > 4 ? 2 * h->waves : 8) : 4) / h->waves);
        h->wpk = (int)std::max<size_t>(1, std::min<size_t>(w, resident));
        if (h->kn.sweep_wgs > 0) h->wpk = h->kn.sweep_wgs;
        // asynchronous solve: the same resident set plus dedicated residual/KKT workgroups
        // (a line-search burst is up to 7 evaluations x H knots per rollout and every knot needs a workgroup of
        //  its own: small batches get at least 240 - measured B = 8: 6.4 -> 5.1 ms, B = 128: 11.1 -> 10.3 ms)
        const int a_wgs = h->kn.sweep_wgs > 0 ? h->wpk : std::max(h->wpk, 240);
        h->a_service = std::max(1, a_wgs / 8);
        if (h->kn.async_service > 0) h->a_service = h->kn.async_service;
        h->a_grid = (int)std::min<size_t>(resident, (size_t)a_wgs + h->a_service);
        if (h->a_grid <= h->a_service) h->a_grid = h->a_service + 1;
    }
    {   // streams of the rounds (one batch: sub-batch streams were measured and only multiply the per-launch
        // latency floor - 4 sub-batches 34.8 ms/step vs 28.2 ms single batch at B = 512)
        RoundStreams& r = h->rs;
        if (hipStreamCreateWithFlags(&r.st, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&r.st_kkt, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&r.ev_join, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&r.ev_start, hipEventDisableTiming) != hipSuccess) {
            g_create_error = "stream creation failed"; cimpc_destroy(h); return CIMPC_ERR_HIP;
        }
    }
    *out = h;
    return CIMPC_OK;
}

int cimpc_destroy(cimpc_handle h) {
    if (!h) return CIMPC_ERR_INVALID;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto& r : h->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : h->ev_pool) (void)hipEventDestroy(e);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->h_counters) (void)hipHostFree(h->h_counters);
    if (h->h_ring) (void)hipHostFree(h->h_ring);
    if (h->h_qin) (void)hipHostFree(h->h_qin);
    if (h->h_result) (void)hipHostFree(h->h_result);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    if (h->h_twfail) (void)hipHostFree(h->h_twfail);
    {
        RoundStreams& r = h->rs;
        if (r.st) { (void)hipStreamSynchronize(r.st); (void)hipStreamDestroy(r.st); }
        if (r.st_kkt) { (void)hipStreamSynchronize(r.st_kkt); (void)hipStreamDestroy(r.st_kkt); }
        if (r.ev_join) (void)hipEventDestroy(r.ev_join);
        if (r.ev_start) (void)hipEventDestroy(r.ev_start);
    }
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return CIMPC_OK;
}

int cimpc_set_stream(cimpc_handle h, void* hip_stream) {
    if (!h) return CIMPC_ERR_INVALID;
    (void)hipStreamSynchronize(h->stream);
    h->advance_pending = false;
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = static_cast<hipStream_t>(hip_stream);
    h->own_stream = false;
    h->external_stream = true;   // caller-ordered stream: the solve runs as ONE batch on it
    return CIMPC_OK;
}

int cimpc_synchronize(cimpc_handle h) {
    if (!h) return CIMPC_ERR_INVALID;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CIMPC_OK;
}

int cimpc_set_linearization(cimpc_handle h, int t, const double* z0, const double* th0,
                            const double* r0, const double* rz0, const double* rth0) {
    if (!h || !z0 || !th0 || !r0 || !rz0 || !rth0) return fail(h, CIMPC_ERR_INVALID, "null argument");
    if (t < 1 || t > h->dm.H_ref) return fail(h, CIMPC_ERR_INVALID, "knot index out of range (1-based)");
    if (int dp = drain_pending(h); dp != CIMPC_OK) return dp;
    const int nx = h->nx, ny = h->ny, nz = h->nz, nth = h->nth, G = h->ki.G;
    const LinLayout L(nx, ny, nth, G, h->ki.generic ? 0 : h->nths, h->ki.generic ? 0 : h->ki.adj);
    if (L.size != h->ki.tab_size) return fail(h, CIMPC_ERR_STATE, "table layout of the library and of the kernel differ");
    std::vector<double>& T = h->h_tab;
    std::fill(T.begin(), T.end(), 0.0);
    auto RZ = [&](int r, int c) { return rz0[r + (size_t)c * nz]; };
    auto RTH = [&](int r, int c) { return rth0[r + (size_t)c * nz]; };
    // blocks (linearization_var_index / term_index are contiguous: index.jl:289-327)
    std::vector<double> Dx((size_t)nx * nx), Ai((size_t)nx * nx), CAi((size_t)ny * nx), CAiB((size_t)ny * ny);
    for (int c = 0; c < nx; ++c)
        for (int r = 0; r < nx; ++r) Dx[r + (size_t)c * nx] = RZ(r, c);
    if (!invert(Dx.data(), Ai.data(), nx))
        return fail(h, CIMPC_ERR_INVALID, "Dx = rz0[idyn, ix] is singular");
    // CAi = Rx * Ai ; CAiB = (Rx * Ai) * Dy1   (schur.jl:40-41)
    for (int r = 0; r < ny; ++r)
        for (int c = 0; c < nx; ++c) {
            double s = 0.0;
            for (int k = 0; k < nx; ++k) s += RZ(nx + r, k) * Ai[k + (size_t)c * nx];
            CAi[r + (size_t)c * ny] = s;
        }
    for (int r = 0; r < ny; ++r)
        for (int c = 0; c < ny; ++c) {
            double s = 0.0;
            for (int k = 0; k < nx; ++k) s += CAi[r + (size_t)k * ny] * RZ(k, nx + c);
            CAiB[r + (size_t)c * ny] = s;
        }
    for (int i = 0; i < ny; ++i)        // W[i*G + j] = Ry1[i,j] - CAiB[i,j], diagonal kept apart
        for (int j = 0; j < ny; ++j)
            T[L.oW + i * L.ldw + j] = (i == j) ? 0.0 : RZ(nx + i, nx + j) - CAiB[i + (size_t)j * ny];
    for (int k = 0; k < nx; ++k) {
        for (int i = 0; i < ny; ++i) T[L.oCAi + k * G + i] = CAi[i + (size_t)k * ny];
        for (int i = 0; i < nx; ++i) T[L.oAi + k * G + i] = Ai[i + (size_t)k * nx];
        for (int i = 0; i < nx; ++i) T[L.oDx + k * G + i] = RZ(i, k);
        for (int i = 0; i < ny; ++i) T[L.oRx + k * G + i] = RZ(nx + i, k);
    }
    for (int k = 0; k < ny; ++k) {
        for (int i = 0; i < nx; ++i) T[L.oDy1 + k * G + i] = RZ(i, nx + k);
        for (int i = 0; i < ny; ++i) T[L.oRy1 + k * G + i] = RZ(nx + i, nx + k);
    }
    for (int k = 0; k < nth; ++k) {
        for (int i = 0; i < nx; ++i) T[L.oRthDyn + k * G + i] = RTH(i, k);
        for (int i = 0; i < ny; ++i) T[L.oRthRst + k * G + i] = RTH(nx + i, k);
    }
    // right-hand sides of the sensitivity pass as the QR sees them (lin_table.h: oGs): Gs[i, c] = (CAi rthdyn[:, c])_i - rthrst[i, c]
    // with the kernel's chain - two partial sums over even / odd k, correctly rounded multiply-adds (IpSolver::schur_solve) -
    // so that the columns the sweep writes do not change by a bit
    for (int c = 0; c < L.nths; ++c)
        for (int i = 0; i < ny; ++i) {
            double bq[2] = {0.0, 0.0};
            for (int k = 0; k < nx; ++k) bq[k & 1] = std::fma(RTH(k, c), CAi[i + (size_t)k * ny], bq[k & 1]);
            T[L.gst ? L.oGs + i * L.nths + c : L.oGs + c * G + i] = (bq[0] + bq[1]) - RTH(nx + i, c);
        }
    // constants of the adjoint form of the sensitivity pass (lin_table.h: oK0, oAiB): A^-1 rthdyn, A^-1 B (plain sums)
    if (L.adj) {
        for (int c = 0; c < L.nths; ++c)
            for (int i = 0; i < nx; ++i) {
                double s = 0.0;
                for (int k = 0; k < nx; ++k) s = std::fma(Ai[i + (size_t)k * nx], RTH(k, c), s);
                T[L.oK0 + c * nx + i] = s;
            }
        for (int i = 0; i < nx; ++i)
            for (int k = 0; k < ny; ++k) {
                double s = 0.0;
                for (int m = 0; m < nx; ++m) s = std::fma(Ai[i + (size_t)m * nx], RZ(m, nx + k), s);
                T[L.oAiB + i * ny + k] = s;
            }
    }
    for (int i = 0; i < ny; ++i) {
        T[L.oVec + LinLayout::V_RY2 * G + i] = RZ(nx + i, nx + ny + i);
        T[L.oVec + LinLayout::V_RY1D * G + i] = RZ(nx + i, nx + i);
        T[L.oVec + LinLayout::V_CAIBD * G + i] = CAiB[i + (size_t)i * ny];
        T[L.oVec + LinLayout::V_RRST0 * G + i] = r0[nx + i];
        T[L.oVec + LinLayout::V_Y10 * G + i] = z0[nx + i];
        T[L.oVec + LinLayout::V_Y20 * G + i] = z0[nx + ny + i];
    }
    for (int i = 0; i < nx; ++i) {
        T[L.oVec + LinLayout::V_RDYN0 * G + i] = r0[i];
        T[L.oVec + LinLayout::V_X0 * G + i] = z0[i];
    }
    for (int k = 0; k < nth; ++k) T[L.oTh0 + k] = th0[k];
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpy(h->d_tab + (size_t)(t - 1) * L.size, T.data(), (size_t)L.size * sizeof(double),
                         hipMemcpyHostToDevice));
    if (!h->knot_set[t - 1]) { h->knot_set[t - 1] = 1; h->n_knots_set++; }
    return CIMPC_OK;
}

int cimpc_set_objective(cimpc_handle h, const double* Q, const double* R, const double* Cg,
                        const double* Cb, const double* V, const double* q_target,
                        const double* v_target) {
    if (!h || !Q || !R) return fail(h, CIMPC_ERR_INVALID, "Q and R are required");
    const cimpc_dims& d = h->dm;
    const size_t H = d.H;
    {   // the weights are symmetric in the reference (objective.jl: Diagonal or relative_state_cost) and every backend but the dense
        // LU relies on it (Qinv / Rinv as symmetric operands of the condensed solve, a symmetric banded matrix): refuse, don't guess
        auto symmetric = [&](const double* A, int n) {
            for (size_t t = 0; t < H; ++t) {
                const double* M = A + t * n * n;
                double amax = 0.0, asym = 0.0;
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j) { amax = std::max(amax, std::fabs(M[i + j * n])); asym = std::max(asym, std::fabs(M[i + j * n] - M[j + i * n])); }
                if (asym > 1e-9 * amax) return false;
            }
            return true;
        };
        // (the reference-default backend - dense jacobian! + LU, on request - takes the blocks as they are, like the reference)
        const bool lu = h->nt.kkt_backend == CIMPC_KKT_DENSE_LU;
        if (!lu && !symmetric(Q, d.nq)) return fail(h, CIMPC_ERR_INVALID, "objective block Q[i] is not symmetric");
        if (!lu && d.nu > 0 && !symmetric(R, d.nu)) return fail(h, CIMPC_ERR_INVALID, "objective block R[i] is not symmetric");
        if (!lu && V && !symmetric(V, d.nq)) return fail(h, CIMPC_ERR_INVALID, "objective block V[i] is not symmetric");
    }
    HIP_TRY(h, hipSetDevice(h->device));
    if (int dp = drain_pending(h); dp != CIMPC_OK) return dp;
    {   // contact-impulse weights below fp64 resolution (1e-100 in every example of the reference): the cf-mode KKT
        // system reduces exactly onto the :configuration solvers (newton_kernels.hip: cf_reduce_*)
        double gmax = 0.0;
        if (Cg) for (size_t k = 0; k < H * d.nc * d.nc; ++k) gmax = std::max(gmax, std::fabs(Cg[k]));
        if (Cb) for (size_t k = 0; k < H * d.nb * d.nb; ++k) gmax = std::max(gmax, std::fabs(Cb[k]));
        h->cf_tiny = gmax <= 1e-30;
    }
    if (V != nullptr) {      // TrackingVelocityObjective (objective.jl:18-47): q_target integrates v_target when not given
        std::vector<double> vt(H * d.nq, 0.0), qt(H * d.nq, 0.0);
        if (v_target) std::copy(v_target, v_target + H * d.nq, vt.begin());
        if (q_target) std::copy(q_target, q_target + H * d.nq, qt.begin());
        else {
            bool any = false;
            for (double x : vt) any = any || x != 0.0;
            if (any) for (size_t t = 1; t < H; ++t) for (int k = 0; k < d.nq; ++k) qt[t * d.nq + k] = qt[(t - 1) * d.nq + k] + vt[(t - 1) * d.nq + k];
        }
        HIP_TRY(h, hipMemcpy(h->d_V, V, H * d.nq * d.nq * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_qt, qt.data(), qt.size() * sizeof(double), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(h->d_vt, vt.data(), vt.size() * sizeof(double), hipMemcpyHostToDevice));
        h->S.V = h->d_V; h->S.q_target = h->d_qt; h->S.v_target = h->d_vt;
        h->velocity_objective = true;      // P is block tridiagonal: the condensed solve does not apply (banded LDL^T / dense LU)
    } else {
        h->S.V = nullptr; h->S.q_target = nullptr; h->S.v_target = nullptr;
        h->velocity_objective = false;
    }
    // the inverses first: whether every R_t could be inverted decides the form of the banded LDL^T (kkt_dense.hip:
    // kkt_banded_kernel eliminates the controls when it can), and that form decides whether the backend fits
    std::vector<double> Qi(H * d.nq * d.nq), Ri(H * d.nu * d.nu);
    bool q_inverted = true, r_inverted = true;
    for (size_t i = 0; i < H; ++i) {
        if (!invert(Q + i * d.nq * d.nq, Qi.data() + i * d.nq * d.nq, d.nq)) q_inverted = false;
        if (d.nu > 0 && !invert(R + i * d.nu * d.nu, Ri.data() + i * d.nu * d.nu, d.nu)) r_inverted = false;
    }
    h->band_reduce_ok = r_inverted && d.nu > 0 && (h->kn.banded_form & 4) == 0;
    select_kkt_backend(h);
    // (the inverses feed the condensed solve - also behind the cf-mode reduction - and the control elimination of the banded LDL^T)
    const bool need_inv = !h->use_dense || (h->cf_reduce && V == nullptr);
    // (a refused objective leaves NO objective: the backend selection above is already committed while the device blocks still hold
    //  the previous call's - ADVICE r05 - so the handle goes back to "set_objective has not been called" instead of solving a mix)
    if (need_inv && !q_inverted) { h->objective_set = false; return fail(h, CIMPC_ERR_INVALID, "objective block Q[i] is singular"); }
    if (need_inv && !r_inverted) { h->objective_set = false; return fail(h, CIMPC_ERR_INVALID, "objective block R[i] is singular"); }
    HIP_TRY(h, hipMemcpy(h->d_Q, Q, H * d.nq * d.nq * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_R, R, H * d.nu * d.nu * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_Qinv, Qi.data(), Qi.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_Rinv, Ri.data(), Ri.size() * sizeof(double), hipMemcpyHostToDevice));
    if (Cg) HIP_TRY(h, hipMemcpy(h->d_Cg, Cg, H * d.nc * d.nc * sizeof(double), hipMemcpyHostToDevice));
    if (Cb) HIP_TRY(h, hipMemcpy(h->d_Cb, Cb, H * d.nb * d.nb * sizeof(double), hipMemcpyHostToDevice));
    h->objective_set = true;
    return CIMPC_OK;
}

int cimpc_set_altitude(cimpc_handle h, const double* alt) {
    if (!h) return CIMPC_ERR_INVALID;
    if (!alt) { h->alt_set = false; return CIMPC_OK; }
    HIP_TRY(h, hipSetDevice(h->device));
    if (int dp = drain_pending(h); dp != CIMPC_OK) return dp;
    HIP_TRY(h, hipMemcpy(h->d_alt, alt, (size_t)h->dm.B * h->dm.nc * sizeof(double), hipMemcpyHostToDevice));
    h->alt_set = true;
    return CIMPC_OK;
}

// The sensitivity memory of failed solves belongs to the reference KNOTS (newton_kernels.hip: dz_rekey_kernel).  Called around
// every change of the window: rekey_out with the window still in force, rekey_in with the new one.
// One logical store for both seams.  The reference keeps ONE ip[t].dz per knot that implicit_dynamics! (B3) and the
// evaluations inside newton_solve! (B4) both write; on the device B4 keeps the accepted evaluation in dz_good and B3 writes
// slot 0 of S.dz.  When the producer changes between two calls on one handle, the blocks of the previous producer are copied
// into the store the next call reads / falls back to, so a solve that fails there keeps what the reference would keep.
static int sync_dz_stores(cimpc_ctx* h, int next_producer) {
    if (h->dz_producer == 0 || h->dz_producer == next_producer) return CIMPC_OK;
    const size_t row = (size_t)h->dm.H * h->nths * h->nd * sizeof(double), B = h->dm.B;
    if (next_producer == 2)      // newton_solve ran last: dz_good -> slot 0 of every rollout
        HIP_TRY(h, hipMemcpy2DAsync(h->S.dz, CS * row, h->S.dz_good, row, row, B, hipMemcpyDeviceToDevice, h->stream));
    else                         // implicit_dynamics ran last: slot 0 -> dz_good
        HIP_TRY(h, hipMemcpy2DAsync(h->S.dz_good, row, h->S.dz, CS * row, row, B, hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CIMPC_OK;
}

static int rekey_out(cimpc_ctx* h) {
    if (h->dz_producer == 0 || !h->window_set) return CIMPC_OK;
    if (!h->d_dz_knot && dev_alloc(h, &h->d_dz_knot, (size_t)h->dm.B * h->dm.H_ref * h->nths * h->nd) != CIMPC_OK) return CIMPC_ERR_HIP;
    int rc = launch_dz_rekey(h->S, h->d_dz_knot, h->d_window, h->dz_producer == 1 ? 0 : 1, 0, h->stream);
    return rc == CIMPC_OK ? CIMPC_OK : fail(h, rc, "sensitivity archive launch failed");
}
static int rekey_in(cimpc_ctx* h) {
    if (h->dz_producer == 0 || !h->d_dz_knot) return CIMPC_OK;
    for (int which = 0; which < 2; ++which) {      // both consumers see the knots' blocks
        int rc = launch_dz_rekey(h->S, h->d_dz_knot, h->d_window, which, 1, h->stream);
        if (rc != CIMPC_OK) return fail(h, rc, "sensitivity restore launch failed");
    }
    return CIMPC_OK;
}

int cimpc_set_window(cimpc_handle h, const int* window) {
    if (!h || !window) return fail(h, CIMPC_ERR_INVALID, "null argument");
    const cimpc_dims& d = h->dm;
    const int B = d.B, H = d.H, K = d.H_ref;
    std::vector<int> w0((size_t)B * (H + 2));
    for (size_t k = 0; k < w0.size(); ++k) {
        const int t = window[k];
        if (t < 1 || t > K) return fail(h, CIMPC_ERR_INVALID, "window entry out of range (1-based knot index)");
        w0[k] = t - 1;                 // the device buckets problems by 0-based knot (newton_kernels.hip: enqueue_eval)
    }
    HIP_TRY(h, hipSetDevice(h->device));
    if (int dp = drain_pending(h); dp != CIMPC_OK) return dp;
    if (int rk = rekey_out(h); rk != CIMPC_OK) return rk;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(h->d_window, w0.data(), w0.size() * sizeof(int), hipMemcpyHostToDevice));
    if (int rk = rekey_in(h); rk != CIMPC_OK) return rk;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->window_set = true;
    h->gait_set = false;               // an explicit window / reference replaces the gait-generated ones
    h->win_mirror = w0;
    return CIMPC_OK;
}

int cimpc_set_reference(cimpc_handle h, const double* q_ref, const double* u_ref,
                        const double* w_ref, const double* gamma_ref, const double* b_ref,
                        const double* theta_ref) {
    if (!h || !q_ref || !u_ref || !theta_ref) return fail(h, CIMPC_ERR_INVALID, "q_ref, u_ref, theta_ref are required");
    const cimpc_dims& d = h->dm;
    const size_t B = d.B, H = d.H;
    HIP_TRY(h, hipSetDevice(h->device));
    if (int dp = drain_pending(h); dp != CIMPC_OK) return dp;
    HIP_TRY(h, hipMemcpy(h->S.ref.q, q_ref, B * (H + 2) * d.nq * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->S.ref.u, u_ref, B * H * d.nu * sizeof(double), hipMemcpyHostToDevice));
    if (w_ref) HIP_TRY(h, hipMemcpy(h->S.ref.w, w_ref, B * H * d.nw * sizeof(double), hipMemcpyHostToDevice));
    else {      // NULL-stream fill: make it complete before the (non-blocking) solver streams can read it
        HIP_TRY(h, hipMemset(h->S.ref.w, 0, B * H * d.nw * sizeof(double)));
        HIP_TRY(h, hipStreamSynchronize(nullptr));
    }
    if (gamma_ref) HIP_TRY(h, hipMemcpy(h->S.ref.g, gamma_ref, B * H * d.nc * sizeof(double), hipMemcpyHostToDevice));
    if (b_ref) HIP_TRY(h, hipMemcpy(h->S.ref.b, b_ref, B * H * d.nb * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->S.ref.th, theta_ref, B * H * h->nth * sizeof(double), hipMemcpyHostToDevice));
    h->reference_set = true;
    h->gait_set = false;
    return CIMPC_OK;
}

int cimpc_implicit_dynamics(cimpc_handle h, const double* q, const double* theta,
                            const double* gamma, const double* b, double* d_out, double* dz,
                            int* status, int* iters, double* z) {
    int rc = check_ready(h, false);
    if (rc != CIMPC_OK) return rc;
    if (!q || !theta) return fail(h, CIMPC_ERR_INVALID, "q and theta are required");
    const cimpc_dims& d = h->dm;
    const size_t B = d.B, H = d.H;
    if (d.mode == CIMPC_MODE_CONFIGURATIONFORCE && (!gamma || !b))
        return fail(h, CIMPC_ERR_INVALID, "gamma and b are required in configurationforce mode");
    HIP_TRY(h, hipSetDevice(h->device));
    if (int sy = sync_dz_stores(h, 2); sy != CIMPC_OK) return sy;
    h->dz_producer = 2;
    TrajDev& T = h->S.cand;
    // evaluation slot 0 of every rollout (slot stride = CS rows)
    auto up = [&](double* dst, const double* src, size_t row_doubles) {
        return hipMemcpy2DAsync(dst, CS * row_doubles * sizeof(double), src, row_doubles * sizeof(double),
                                row_doubles * sizeof(double), B, hipMemcpyHostToDevice, h->stream);
    };
    HIP_TRY(h, up(T.q, q, (H + 2) * d.nq));
    HIP_TRY(h, up(T.th, theta, H * h->nth));
    if (gamma) HIP_TRY(h, up(T.g, gamma, H * d.nc));
    if (b) HIP_TRY(h, up(T.b, b, H * d.nb));
    // queue slot 0 of every rollout, then run rounds until no solve is parked
    HIP_TRY(h, hipMemsetAsync(h->Q.count, 0, 2 * h->Q.K * QPAD * sizeof(int), h->stream));
    HIP_TRY(h, hipMemsetAsync(h->Q.head, 0, h->Q.K * QPAD * sizeof(int), h->stream));
    {
        NewtonDev Sk = h->S;
        Sk.WQ = h->Q; Sk.WQ.par = 0;
        rc = launch_enqueue_all(Sk, h->stream);
        if (rc != CIMPC_OK) return fail(h, rc, "enqueue launch failed");
    }
    for (int pass = 0; pass < 64; ++pass) {
        const int par = pass & 1;
        HIP_TRY(h, hipMemsetAsync(h->S.counters, 0, 8 * CPAD * sizeof(int), h->stream));
        rc = run_sweep(h, par, h->S.counters + 2 * CPAD, z ? h->d_zout : nullptr, h->stream);
        if (rc != CIMPC_OK) return rc;
        HIP_TRY(h, hipMemsetAsync(h->Q.count + (size_t)par * h->Q.K * QPAD, 0, h->Q.K * QPAD * sizeof(int), h->stream));
        HIP_TRY(h, hipMemsetAsync(h->Q.head, 0, h->Q.K * QPAD * sizeof(int), h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->h_counters, h->S.counters + 2 * CPAD, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->h_counters[0] == 0) break;      // no solve was parked
    }
    auto down = [&](void* dst, const void* src, size_t row_bytes) {
        return hipMemcpy2DAsync(dst, row_bytes, src, CS * row_bytes, row_bytes, B, hipMemcpyDeviceToHost, h->stream);
    };
    if (d_out) HIP_TRY(h, down(d_out, h->S.d, H * h->nd * sizeof(double)));
    if (dz) HIP_TRY(h, down(dz, h->S.dz, H * h->nths * h->nd * sizeof(double)));
    if (status) HIP_TRY(h, down(status, h->S.ip_status, H * sizeof(int)));
    if (iters) HIP_TRY(h, down(iters, h->S.ip_iters, H * sizeof(int)));
    if (z) HIP_TRY(h, down(z, h->d_zout, H * h->nz * sizeof(double)));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CIMPC_OK;
}

// B2 seam: one reference-side callback on n points of knot t (ip_kernel_impl.h: ip_callback_kernel)
static int run_ip_callback(cimpc_handle h, int t, int n, int op, const double* z, const double* theta, const double* alt,
                           const double* r, double kappa, double reg, double* out) {
    if (!h || !z || !out || n <= 0) return fail(h, CIMPC_ERR_INVALID, "null argument");
    if (t < 1 || t > h->dm.H_ref) return fail(h, CIMPC_ERR_INVALID, "knot index out of range (1-based)");
    if (!h->knot_set[t - 1]) return fail(h, CIMPC_ERR_STATE, "set_linearization has not been called for this knot");
    if (h->ki.generic) return fail(h, CIMPC_ERR_INVALID, "B2 callbacks exist for the compiled lane-group models only");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t nz = h->nz, nth = h->nth, nc = h->dm.nc;
    const size_t naux = std::max(nz, nth);               // theta (op 0) or r (op 1)
    const size_t tot = (size_t)n * (2 * nz + naux + nc);
    double* buf = nullptr;
    HIP_TRY(h, hipMalloc((void**)&buf, tot * sizeof(double)));
    double* d_z = buf; double* d_out = buf + (size_t)n * nz; double* d_aux = d_out + (size_t)n * nz; double* d_alt = d_aux + (size_t)n * naux;
    auto done = [&](int rc) { (void)hipFree(buf); return rc; };
    if (hipMemcpy(d_z, z, (size_t)n * nz * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return done(fail(h, CIMPC_ERR_HIP, "upload failed"));
    const double* aux = op == 0 ? theta : r;
    if (!aux) return done(fail(h, CIMPC_ERR_INVALID, "null argument"));
    if (hipMemcpy(d_aux, aux, (size_t)n * (op == 0 ? nth : nz) * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return done(fail(h, CIMPC_ERR_HIP, "upload failed"));
    if (alt && hipMemcpy(d_alt, alt, (size_t)n * nc * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return done(fail(h, CIMPC_ERR_HIP, "upload failed"));
    IpCallbackArgs a{};
    a.tab = h->d_tab; a.knot = t - 1; a.n = n; a.op = op; a.z = d_z;
    a.theta = op == 0 ? d_aux : nullptr; a.alt = (op == 0 && alt) ? d_alt : nullptr; a.r = op == 1 ? d_aux : nullptr;
    a.kappa = kappa; a.reg = reg; a.out = d_out;
    (void)hipGetLastError();                             // (a stale error of an earlier call must not be read as this launch's)
    const int rc = launch_ip_callback(&h->dm, a, h->stream);
    if (rc != CIMPC_OK) return done(fail(h, rc, "callback launch failed"));
    if (hipStreamSynchronize(h->stream) != hipSuccess ||
        hipMemcpy(out, d_out, (size_t)n * nz * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return done(fail(h, CIMPC_ERR_HIP, "callback kernel failed"));
    return done(CIMPC_OK);
}

int cimpc_ip_residual(cimpc_handle h, int t, int n, const double* z, const double* theta, const double* alt, double kappa, double* r) {
    return run_ip_callback(h, t, n, 0, z, theta, alt, nullptr, kappa, 0.0, r);
}

int cimpc_ip_linear_solve(cimpc_handle h, int t, int n, const double* z, const double* r, double reg, double* delta) {
    return run_ip_callback(h, t, n, 1, z, nullptr, nullptr, r, 0.0, reg, delta);
}

int cimpc_kkt_solve(cimpc_handle h, const double* r, double beta, double* delta) {
    int rc = check_ready(h, true);
    if (rc != CIMPC_OK) return rc;
    if (!r || !delta) return fail(h, CIMPC_ERR_INVALID, "null argument");
    const size_t n = (size_t)h->dm.B * h->N;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(h->d_rhs, r, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (h->use_mixed) { rc = ensure_mixed_ws(h); if (rc != CIMPC_OK) return rc; }
    else if (h->use_dense) { rc = ensure_dense_ws(h); if (rc != CIMPC_OK) return rc; }
    // attempt 0 may take a twisted kernel (two chains per rollout with bounded hand-over waits); if one of them timed out - the
    // mapped counter moved - the numbers are poisoned and the solve is repeated ONCE on the one-ended kernels (round 6)
    for (int attempt = 0; attempt < 2; ++attempt) {
        h->S.kkt_tw_epoch += 1;
        NewtonDev Sk = h->S;
        if (attempt == 1) { Sk.kkt_tw_raw = 0; Sk.kkt_tw_band = 0; }
        const int fails0 = *(volatile int*)h->h_twfail;
        prof_begin(h, PC_KKT);
        if (h->use_mixed) {
            rc = launch_kkt_mixed_raw(Sk, h->d_rhs, beta, Sk.delta, h->d_mix_ws, h->d_mix_nfb, h->stream);
        } else if (h->use_dense) {
            if (h->cf_reduce ? (h->velocity_objective && kkt_banded_twisted_available(cf_shadow(Sk))) : (h->use_banded && kkt_banded_twisted_available(Sk))) h->n_kkt_twisted++;
            rc = h->cf_reduce ? launch_kkt_cf_reduced_raw(Sk, h->d_rhs, beta, Sk.delta, h->d_cf_ws, h->d_dense_ws, h->stream)
                              : launch_kkt_dense_raw(Sk, h->d_rhs, beta, Sk.delta, h->d_dense_ws, h->stream, h->use_banded);
        } else {
            if (Sk.kkt_tw_raw != 0 && kkt_twisted_available(Sk)) h->n_kkt_twisted++;      // (launch_kkt_raw's own test)
            rc = launch_kkt_raw(Sk, h->d_rhs, beta, Sk.delta, h->stream);
        }
        prof_end(h);
        if (rc != CIMPC_OK) return fail(h, rc, "kkt launch failed");
        HIP_TRY(h, hipMemcpyAsync(delta, h->S.delta, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        const int fails1 = *(volatile int*)h->h_twfail;
        if (fails1 == fails0) break;
        h->n_kkt_tw_fallbacks += fails1 - fails0;
        h->twfail_seen = fails1;
    }
    return CIMPC_OK;
}

int cimpc_kkt_solve_rho(cimpc_handle h, const double* r, double rho, double* delta) {
    if (!h) return CIMPC_ERR_INVALID;
    if (!(rho >= 0.0) || !(h->nt.kappa > 0.0)) return fail(h, CIMPC_ERR_INVALID, "rho must be >= 0 (and kappa > 0)");
    // the kernels form rho = H * beta * kappa (newton_jacobian.jl:169-186): hand them the beta that reproduces rho
    const double Hk = (double)h->dm.H, kap = h->nt.kappa;
    auto f = [&](double b) { return Hk * b * kap; };
    double beta = rho / (Hk * kap), err = std::fabs(f(beta) - rho), lo = beta, hi = beta;
    for (int k = 0; k < 4 && err != 0.0; ++k) {
        lo = std::nextafter(lo, -HUGE_VAL); hi = std::nextafter(hi, HUGE_VAL);
        for (double c : {lo, hi}) { const double e = std::fabs(f(c) - rho); if (e < err) { err = e; beta = c; } }
    }
    return cimpc_kkt_solve(h, r, beta, delta);
}

int cimpc_newton_solve_dev(cimpc_handle h, const double* q0_dev, const double* q1_dev, int warm_start) {
    int rc = check_ready(h, true);
    if (rc != CIMPC_OK) return rc;
    if (!q0_dev || !q1_dev) return fail(h, CIMPC_ERR_INVALID, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    if (int sy = sync_dz_stores(h, 1); sy != CIMPC_OK) return sy;
    h->dz_producer = 1;
    NewtonDev& S = h->S;
    const int tw_fail0 = *(volatile int*)h->h_twfail;      // time-outs of the twisted kernels' hand-overs before this solve
    h->twfail_seen = tw_fail0;
    const auto t0 = std::chrono::steady_clock::now();
    auto over_budget = [&]() {
        if (h->nt.max_time <= 0.0) return false;
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return el >= h->nt.max_time;
    };
    // the solve runs on the library's streams: they wait (on the device) for what the caller's stream holds - the uploads of
    // q0 / q1 - instead of the host waiting for it
    if (!h->external_stream) {
        HIP_TRY(h, hipEventRecord(h->rs.ev_start, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(h->rs.st, h->rs.ev_start, 0));
    }
    HIP_TRY(h, hipMemsetAsync(S.stats, 0, (size_t)h->dm.B * 4 * sizeof(long long), h->external_stream ? h->stream : h->rs.st));
    // end of a solve: statistics, Newton iteration counts, r_norm and u_1 of every rollout in one block, one copy (queued on the
    // solve's stream; the caller synchronises)
    auto finish_results = [&](hipStream_t st) -> int {
        if (launch_dz_commit(S, st) != CIMPC_OK) return fail(h, CIMPC_ERR_HIP, "sensitivity flush launch failed");
        int rf = launch_solve_finish(S, h->d_result, st);
        if (rf != CIMPC_OK) return fail(h, rf, "result kernel launch failed");
        HIP_TRY(h, hipMemcpyAsync(h->h_result, h->d_result, (8 + (size_t)h->dm.B * (h->dm.nu + 2)) * sizeof(double), hipMemcpyDeviceToHost, st));
        return CIMPC_OK;
    };
    // ---- one persistent launch: every rollout advances on its own chain (newton_async_impl.h).  Entered
    //      from the start (from_reset) or with the rollouts the lock-step rounds left active (hybrid). ----
    auto run_async = [&](bool from_reset, long long rounds_before, int next_par = -1) -> int {
        IpQueues LQ = h->Q; LQ.par = next_par >= 0 ? next_par : (int)(rounds_before & 1);      // lock-step queue of the round that would come next
        hipStream_t st = h->external_stream ? h->stream : h->rs.st;
        const size_t K = h->Q.K;
        if (h->async_dirty) {     // entries are reset by their consumers; only an aborted solve leaves some behind
            HIP_TRY(h, hipMemsetAsync(h->a_items, 0xFF, K * h->a_cap * sizeof(int), st));
            HIP_TRY(h, hipMemsetAsync(h->a_jobs, 0xFF, (h->a_rq_cap + h->a_kq_cap) * sizeof(int), st));
            h->async_dirty = false;
        }
        HIP_TRY(h, hipMemsetAsync(h->a_ctrl, 0, (2 * K * QPAD + 64 + 33 * 16) * sizeof(int), st));
        volatile int* hm = (volatile int*)h->h_ring;
        hm[3] = 0;
        NewtonDev Sk = S;
        Sk.b0 = 0; Sk.nb_launch = h->dm.B; Sk.counters = h->d_ring; Sk.counters_next = h->d_ring + 8 * CPAD;
        Sk.host_flag = h->h_ring_dev;
        // the tail is latency-bound: its stragglers (rollouts that exhaust the line search in every iteration)
        // evaluate all seven step lengths at once there, whatever the throughput-oriented setting of the rounds
        if (!from_reset) Sk.spec_all = h->kn.spec_tail;
        Sk.WQ = h->Q; Sk.WQ.par = 0;
        Sk.WQ.items = h->a_items; Sk.WQ.cap = (int)h->a_cap;
        Sk.WQ.count = h->a_ctrl; Sk.WQ.head = h->a_ctrl + K * QPAD;
        AsyncQ& A = Sk.A;
        A.on = 1;
        int* c = h->a_ctrl + 2 * K * QPAD;
        A.rq_items = h->a_jobs; A.rq_head = c + 0; A.rq_tail = c + 16;
        A.kq_items = h->a_jobs + h->a_rq_cap; A.kq_head = c + 32; A.kq_tail = c + 48;
        A.n_done = c + 8;
        A.epoch = c + 64;
        A.evals_left = h->a_evals;
        A.abort_flag = (volatile int*)(h->h_ring_dev + 3);
        // (hybrid tail: few rollouts left - a smaller resident set means fewer idle workgroups polling next to the working ones)
        int a_grid = h->a_grid, a_service = h->a_service;
        const int tail_grid = h->kn.async_tail_grid >= 0 ? h->kn.async_tail_grid : std::max(256, 3 * h->async_tail);      // (256 = one workgroup per CU: the tail is latency bound, co-resident workgroups slow each other)
        if (!from_reset && tail_grid > 0 && tail_grid < a_grid) {
            a_grid = tail_grid;
            a_service = std::max(1, a_grid / 8);
        }
        A.n_service = a_service;
        A.flags = h->kn.async_flags;
        A.idle_sleep = h->kn.async_sleep;
        A.idle_spins = h->kn.async_spins;
        A.wake_fan = h->kn.async_fan;
        A.dbg = h->a_dbg;
        if (h->a_dbg) HIP_TRY(h, hipMemsetAsync(h->a_dbg, 0, 16 * sizeof(long long), st));
        A.B = h->dm.B;
        // KKT stage of the persistent kernel as two cooperating jobs (the chains of the twisted solve, newton_async_impl.h): 16-wide tiles,
        // four-wave workgroups, a horizon the twisted form accepts, and no hand-over of this solve has timed out so far
        A.kkt_tw = ((h->kn.async_kkt_tw == 2 || (h->kn.async_kkt_tw == 1 && (from_reset ? h->dm.B <= 32 : h->async_tail <= 32))) && h->kn.kkt_twisted != 0 && h->dm.nq <= 16 && h->dm.nu <= 16 && std::min(h->waves, 4) == 4 && kkt_twisted_available(Sk) &&
                    *(volatile int*)h->h_twfail == tw_fail0) ? 1 : 0;
        if (A.kkt_tw) h->n_kkt_twisted++;
        S.kkt_tw_epoch += h->nt.max_iter + 2;      // the launch's KKT stages take the stamps Sk.kkt_tw_epoch + 1 + (Newton iterations done)
        prof_begin(h, PC_OTHER, st);
        int rc2 = from_reset ? launch_reset(Sk, q0_dev, q1_dev, warm_start, st) : launch_async_handoff(Sk, LQ, st);
        prof_end(h, st);
        if (rc2 != CIMPC_OK) return fail(h, rc2, "reset / hand-off launch failed");
        IpParams p = make_ip_params(h, S.cand, 0, h->d_ring + 2 * CPAD, nullptr);
        if (S.dtn != nullptr) { p.nu = S.nu_cand; p.dtn = S.dtn; }
        p.Q = Sk.WQ;
        p.iter_cap = h->ip.max_iter + 1;      // no solve is parked (solves parked by the lock-step rounds resume and finish)
        p.A = A;
        long long solved_before = 0;      // interior-point problems the lock-step rounds had solved (profiling only)
        if (h->prof_on && !from_reset) { long long sv[4]; if (int rs = read_stats(h, sv); rs != CIMPC_OK) return rs; solved_before = sv[1]; }
        prof_begin(h, PC_ASYNC, st);
        rc2 = launch_newton_async(&h->dm, p, Sk, std::min(h->waves, 4), a_grid, st);
        prof_end(h, st);
        if (rc2 != CIMPC_OK) return fail(h, rc2, "asynchronous newton launch failed");
        if (int rf = finish_results(st); rf != CIMPC_OK) return rf;      // (queued behind the persistent kernel)
        // The host watches the persistent kernel: the reference's wall-clock budget ends the loop silently
        // (newton.jl:187-277), and a watchdog turns a kernel that stopped making progress into an error
        // instead of a hang (the kernel polls the host-mapped abort flag).
        const double watchdog_s = h->kn.watchdog_s;
        const bool budget = h->nt.max_time > 0.0 && h->nt.max_time < 1.0e6;
        bool timed_out = false;
        const auto tw = std::chrono::steady_clock::now();
        long long polls = 0;
        while (hipStreamQuery(st) == hipErrorNotReady) {
            if (budget && over_budget()) { hm[3] = 1; h->async_dirty = true; break; }
            if ((++polls & 0x3FF) == 0 &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count() > watchdog_s) {
                hm[3] = 1; h->async_dirty = true; timed_out = true; break;
            }
        }
        HIP_TRY(h, hipStreamSynchronize(st));
        if (const int twf = *(volatile int*)h->h_twfail; twf != h->twfail_seen) { h->n_kkt_tw_fallbacks += twf - h->twfail_seen; h->twfail_seen = twf; }
        if (timed_out) return fail(h, CIMPC_ERR_HIP, "asynchronous solve: watchdog expired (no completion within CIMPC_ASYNC_WATCHDOG_S)");
        if (h->a_dbg) {
            long long dv[16];
            HIP_TRY(h, hipMemcpy(dv, h->a_dbg, sizeof(dv), hipMemcpyDeviceToHost));
            fprintf(stderr, "[cimpc async] WG-ms: other %.2f kkt %.2f resid %.2f ip %.2f | jobs: kkt %lld resid %lld serve %lld | grid %d service %d | group-trips %.2fM active %.1f%% | group-ms pop %.1f fence %.1f | pop: cas %lld spins %lld claim-ms %.1f wait-ms %.1f\n",
                    dv[0] * 1e-5, dv[1] * 1e-5, dv[2] * 1e-5, dv[3] * 1e-5, dv[9], dv[10], dv[11], h->a_grid, h->a_service, dv[5] * 1e-6, dv[5] ? 100.0 * dv[4] / dv[5] : 0.0, dv[6] * 1e-5, dv[7] * 1e-5, dv[12], dv[13], dv[14] * 1e-5, dv[15] * 1e-5);
        }
        const long long stv[4] = {(long long)h->h_result[0], (long long)h->h_result[1], (long long)h->h_result[2], (long long)h->h_result[3]};
        h->last_stats.sweeps = stv[0];
        h->last_stats.ip_solves = stv[1];
        h->last_stats.ip_iters = stv[2];
        h->last_stats.ip_failures = stv[3];
        h->last_stats.rounds = rounds_before + 1;
        h->last_stats.newton_iters = (long long)h->h_result[4];
        h->prof_ip_problems += solved_before;
        h->prof_async_problems += stv[1] - solved_before;
        if (from_reset) h->prof_kkt_systems += h->last_stats.newton_iters;
        return CIMPC_OK;
    };
    if (h->use_dense) { rc = ensure_dense_ws(h); if (rc != CIMPC_OK) return rc; }
    if (h->use_mixed) { rc = ensure_mixed_ws(h); if (rc != CIMPC_OK) return rc; }
    const bool full_async = h->async_on && !h->use_dense && !h->use_mixed && (h->async_mode == 1 || (h->async_mode == 2 && h->dm.B >= 4 && h->dm.B <= h->kn.async_full_max));
    const bool hybrid = h->async_on && !h->use_dense && !h->use_mixed && h->async_mode == 2 && !full_async && h->dm.B > h->kn.async_full_max;
    if (full_async) return run_async(true, 0);
    // safety net only: every Newton iteration needs at most 3 (speculative) rounds, each evaluation at
    // most ceil(max_iter / iter_cap) launches of the resumable interior-point sweep
    // (drain parking guarantees drain_min iterations of progress per launch, iter_cap parking iter_cap)
    const int min_progress = h->kn.drain_pct > 0 ? std::min(h->iter_cap, h->kn.drain_min) : h->iter_cap;
    const int max_rounds = (h->nt.max_iter * 8 + 2) * ((h->ip.max_iter + min_progress - 1) / min_progress + 1);
    // the rounds run on the library's private streams, or on the caller's stream when one was given
    RoundStreams sb = h->rs;
    if (h->external_stream) sb.st = h->stream;
    // Lock-step rounds.  Every kernel of a round is driven by device-side state (work queues, per-rollout stage); the host only
    // looks at the counters the last decision block publishes, one round at a time.  (Measured and removed again: rounds enqueued
    // one ahead of the host's knowledge - the KKT kernel must then be launched blind at full grid, 15.9 vs 13.7 ms at B = 512 - and
    // chained rounds {sweep || KKT} -> sweep of the new candidates -> residual: every extra sweep launch pays the slowest-solve
    // tail again, 10.7 -> 11.0-11.3 ms.)
    long long launched = 0, completed = 0, rounds = 0;
    const bool dbg_rounds = h->kn.debug_rounds;
    int last_kkt = 0, last_sweep = h->dm.B, last_slots = h->dm.B, last_parked = 0, last_new_slots = h->dm.B;
    // Single rollouts (B < 4: below the persistent kernel's range) keep ONE round queued ahead of the one the host waits for: such a
    // round is launched BLIND - KKT kernel on the full list range with its count read on the device, everything else is
    // device-driven anyway - and costs three empty launches if the solve turns out to be over; in exchange no round waits for the
    // host to see the previous one's stamp and launch (about 10 us per round of a 0.7 ms solve).  Warm-started solves only - the
    // cadence of an MPC loop, five Newton iterations over eight to ten rounds: hopper H = 20 0.652 -> 0.614 ms per MPC step; the
    // four rounds of a cold start lose more to the empty launches than they gain (0.685 -> 0.706 ms).
    // (not with a wall-clock budget: the round queued ahead would still run - and step the trajectory - after the host has stopped
    //  waiting, newton.jl:187-277 ends silently at the check)
    const bool ahead = h->dm.B < 4 && !h->use_mixed && warm_start != 0 && !(h->nt.max_time > 0.0 && h->nt.max_time < 1.0e6);
    auto launch_round = [&](long long r, bool blind) -> int {
        // [KKT for rollouts that start an iteration] || sweep -> residual of every evaluated slot -> line-search decision
        const int slot = (int)(r & 1), par = (int)(r & 1);      // round r consumes Q[par] and leaves the next round's requests in Q[par ^ 1]
        int* d_cnt = h->d_ring + 8 * CPAD * slot;
        S.kkt_tw_epoch += 1;      // stamp of this round's KKT stage (epoch-valued hand-over flags of the twisted kernels)
        // a hand-over of a twisted kernel timed out earlier in this solve (its rollouts were queued again): one-ended kernels from here on
        const int tw_fails = *(volatile int*)h->h_twfail;
        if (tw_fails != h->twfail_seen) { h->n_kkt_tw_fallbacks += tw_fails - h->twfail_seen; h->twfail_seen = tw_fails; }
        const bool tw_off = tw_fails != tw_fail0;
        NewtonDev Sk = S;
        if (tw_off) Sk.kkt_tw_band = 0;
        Sk.b0 = 0; Sk.nb_launch = h->dm.B; Sk.counters = d_cnt;
        Sk.counters_next = h->d_ring + 8 * CPAD * (slot ^ 1);
        Sk.host_flag = h->h_ring_dev + 8 * slot;
        Sk.A.n_done = hybrid ? h->a_ctrl + 2 * (size_t)h->Q.K * QPAD + 8 : nullptr;
        Sk.round_stamp = (int)(r + 1);
        Sk.WQ = h->Q; Sk.WQ.par = par;       // the queue being consumed
        const bool kkt = (r > 0) && (blind || last_kkt > 0);
        const int n_kkt = blind ? h->dm.B : last_kkt;
        const int* n_kkt_dev = blind ? h->d_ring + 8 * CPAD * (slot ^ 1) + 1 * CPAD : nullptr;      // the previous round's count of KKT requests
        // (round 4, measured and removed: the three-wave kernel in rounds whose sweep is light - <= 16 k / 26 k / 60 k problems: KKT kernel
        //  time 3.6 -> 3.4 / 3.0 / 2.7 ms per step, but the sweep and the tail give it back, 7.93 -> 7.88 / 7.88 / 7.96 ms:
        //  profiles/r04/knob_kkt_pipe_small.log)
        // Wide tiles (nq or nu in 17 .. 24: the centroidal sizes) hold one rollout per CU in either kernel (105 KB / 128 KB of LDS), so the
        // three-wave kernel costs no extra CUs there - always taken (round 4, BASELINE configs[4] at 64 rollouts: KKT 6.1 -> 3.2 ms per
        // step, 16.3 -> 13.3 ms; at 128 rollouts 24.2 -> 21.2 ms: profiles/r04/cent_kkt_pipe.log)
        const bool wide_tiles = (h->dm.nq > 16 || h->dm.nu > 16) && h->dm.nq <= 24 && h->dm.nu <= 24;
        // (CIMPC_KKT_PIPE selects between the one-wave and the three-wave kernel only - ADVICE r05: a 2 from the environment went
        //  to the twisted kernel past its availability test and its pair bound; the twisted form is chosen below, where both are checked)
        int pipe = h->kn.kkt_pipe >= 0 ? std::min(h->kn.kkt_pipe, 1) : ((!h->kkt_overlap && n_kkt <= h->kn.kkt_pipe_max) || wide_tiles) ? 1 : 0;
        // Round 5: where the pipelined kernel runs - the solve is on the critical path - the TWISTED solve takes its place: two
        // workgroups per rollout factor the block penta-diagonal matrix from both ends (two chains of about H / 2 steps), as long as
        // every pair is resident at once
        if (pipe == 1 && !tw_off && h->kn.kkt_twisted != 0 && n_kkt <= h->kn.kkt_tw_max && kkt_twisted_available(Sk)) pipe = 2;
        // ... and in the OVERLAPPED rounds of larger batches when few rollouts start a Newton iteration (late rounds): the packed
        // one-wave recursion (266 us) then outlasts the thinning sweep next to it, the twisted pair (two workgroups per rollout) does not
        if (pipe == 0 && !tw_off && h->kkt_overlap && !blind && h->kn.kkt_twisted != 0 && h->kkt_tw_overlap_max > 0 && n_kkt > 0 && n_kkt <= h->kkt_tw_overlap_max &&
            kkt_twisted_available(Sk)) pipe = 2;
        // Round 6: the rounds' KKT stage NEXT TO the sweep (overlapped rounds, one-wave packed kernel so far) takes the DUO kernel - the
        // two-ended solve as two one-wave chains in one workgroup, the LDS footprint of the packed kernel's workgroup - while every
        // system gets a workgroup at once (one per CU): the packed recursion is a chain of H steps, 0.27 - 0.33 ms whatever the
        // number of systems, and outlasted the sweep in 11 of the 16 rounds of the headline step
        // ... in the rounds whose sweep is SHORTER than the packed recursion would be (the host knows the problems queued for it: evaluation
        // slots requested + solves parked; a launch takes about 80 us + 1 us per 112 problems): next to a long sweep the packed kernel is
        // off the critical path anyway and takes half as many CUs from it (B = 1024: 11.7 -> 12.5 ms with the duo kernel everywhere;
        // B = 256: 6.46 -> 5.89 ms, B = 512 unchanged)
        // (the NEWLY REQUESTED evaluation slots only - not the parked solves, not the slots re-listed while they wait for one: which
        //  solves a launch parks depends on timing, and a kernel choice that followed it would make the iterates differ from run to
        //  run - the two kernels agree to 1e-12, not to the bit; scripts/dbg/repro_check.py)
        const long long sweep_problems = blind ? -1 : (long long)last_new_slots * h->dm.H;
        if (pipe == 0 && !tw_off && h->kkt_overlap && h->kn.kkt_duo != 0 && h->kn.kkt_twisted != 0 && n_kkt > 0 && n_kkt <= h->kn.kkt_duo_max && sweep_problems >= 0 &&
            sweep_problems <= h->kn.kkt_duo_hint && kkt_duo_available(Sk)) pipe = 3;
        if (kkt && (pipe == 2 || pipe == 3) && !h->use_dense && !h->use_mixed && (h->kkt_overlap ? h->kn.kkt_packed : true)) h->n_kkt_twisted++;
        if (kkt && !h->kkt_overlap) {   // small batches: latency matters, keep the KKT result in this round
            Sk.kkt_same_round = 1;
            prof_begin(h, PC_KKT, sb.st);
            // (same compact list / packed or pipelined kernel as the overlapped path)
            int rr = (h->use_dense || h->use_mixed) ? launch_kkt_general(h, Sk, sb.st) : launch_kkt_packed(Sk, n_kkt, par ^ 1, sb.st, n_kkt_dev, pipe);
            prof_end(h, sb.st);
            if (rr != CIMPC_OK) return fail(h, rr, "kkt launch failed");
        }
        if (kkt && h->kkt_overlap) {
            // KKT of the rollouts that start a Newton iteration, next to the sweep on its own stream.  No fork
            // event: the host has seen the previous round's stamp, so everything before is complete (every decision
            // block releases its results before it takes its ticket).  (Launched BEFORE the sweep: the other order was
            // measured 10 % slower in round 1 and 20 % slower in round 3 (8.37 -> 10.05 ms, profiles/r03/knob_order.log) - behind the
            // persistent sweep the KKT workgroups wait for sweep workgroups to leave: 3.6 -> 6.2 ms of KKT time per step.)
            prof_begin(h, PC_KKT, sb.st_kkt);
            // the list was built by the decision kernel of the previous round (its queue parity)
            int rk = (h->use_dense || h->use_mixed) ? launch_kkt_general(h, Sk, sb.st_kkt)
                                  : h->kn.kkt_packed ? launch_kkt_packed(Sk, last_kkt, par ^ 1, sb.st_kkt, nullptr, pipe)
                                                     : launch_kkt(Sk, sb.st_kkt);
            prof_end(h, sb.st_kkt);
            if (rk != CIMPC_OK) return fail(h, rk, "kkt launch failed");
            if (hipEventRecord(sb.ev_join, sb.st_kkt) != hipSuccess) return fail(h, CIMPC_ERR_HIP, "join record failed");
        }
        // parking (iter_cap) protects a busy round from one long solve; in the sparse tail of a solve
        // it only adds rounds, so a round that serves few rollouts lets every solve run to the end
        const int tail_div = h->kn.tail_div;
        // (batches below tail_div rollouts never park: a round there is as long as its longest solve whichever way it is cut)
        // (in a blind round `last_sweep` is one round stale: the cap is then chosen without it)
        const bool sparse = tail_div > 0 && (h->dm.B < tail_div || (!blind && (long long)last_sweep * tail_div <= h->dm.B));
        const int cap = sparse ? h->ip.max_iter : h->iter_cap;
        // problems of this round as far as the host knows them: the evaluation slots requested + the solves the last sweep parked
        const long long hint = blind ? -1 : (long long)last_slots * h->dm.H + last_parked;
        if (dbg_rounds && !blind) {      // diagnostics only: the problems this round's sweep will find in its queues (a blocking read)
            std::vector<int> qc((size_t)h->Q.K * QPAD);
            (void)hipStreamSynchronize(sb.st); (void)hipStreamSynchronize(sb.st_kkt);
            if (hipMemcpy(qc.data(), h->Q.count + (size_t)par * h->Q.K * QPAD, qc.size() * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) {
                long long tot = 0;
                for (int k = 0; k < h->Q.K; ++k) tot += qc[(size_t)k * QPAD];
                fprintf(stderr, "[cimpc round %lld] sweep launch: %lld problems queued (host hint %lld)\n", r, tot, hint);
            }
        }
        int rr = run_sweep(h, par, d_cnt + 2 * CPAD, nullptr, sb.st, cap, d_cnt + 3 * CPAD, true, hint);
        if (rr != CIMPC_OK) return rr;
        // the evaluation slots of this round: the compact list its requesters built (small batches run the KKT stage in the same
        // round as the evaluation of its candidates - not known to the host at launch: every (rollout, slot) pair gets a block there)
        const int n_slots = h->kkt_overlap ? last_slots : (h->dm.B < 4 ? -2 : -1);
        // round 4: the per-slot residual kernel goes in FRONT of the join with the overlapped KKT kernel (it reads the sweep's results
        // only; the rollouts the KKT kernel works on have no slot on this round's list), the decision kernel behind it - in the
        // rounds whose KKT recursion outlasts the sweep (about six per step of the headline batch) the 26 us of the slot kernel
        // leave the critical path
        const bool split_join = kkt && h->kkt_overlap && n_slots >= 0;
        prof_begin(h, PC_RESID, sb.st);
        if (split_join) {
            rr = launch_resid_decide(Sk, sb.st, n_slots, 1);
            if (rr != CIMPC_OK) return fail(h, rr, "residual launch failed");
        }
        if (kkt && h->kkt_overlap && hipStreamWaitEvent(sb.st, sb.ev_join, 0) != hipSuccess) return fail(h, CIMPC_ERR_HIP, "join failed");
        rr = launch_resid_decide(Sk, sb.st, n_slots, split_join ? 2 : 0);
        prof_end(h, sb.st);
        if (rr != CIMPC_OK) return fail(h, rr, "residual launch failed");
        return CIMPC_OK;
    };
    HIP_TRY(h, hipMemsetAsync(h->Q.count, 0, h->ctl_ints * sizeof(int), sb.st));      // queue counters, heads, round counters
    if (hybrid) HIP_TRY(h, hipMemsetAsync(h->a_ctrl + 2 * (size_t)h->Q.K * QPAD, 0, 64 * sizeof(int), sb.st));
    ((volatile int*)h->h_ring)[2] = 0;
    ((volatile int*)h->h_ring)[10] = 0;
    {
        NewtonDev Sk = S;
        Sk.b0 = 0; Sk.nb_launch = h->dm.B; Sk.counters = h->d_ring;
        Sk.WQ = h->Q; Sk.WQ.par = 0;
        prof_begin(h, PC_OTHER, sb.st);
        rc = launch_reset(Sk, q0_dev, q1_dev, warm_start, sb.st);
        prof_end(h, sb.st);
        if (rc != CIMPC_OK) return fail(h, rc, "reset launch failed");
    }
    while (true) {
        if (completed >= max_rounds) {          // round limit reached with work left: a scheduling bug, never a silent partial solve
            (void)hipStreamSynchronize(sb.st); (void)hipStreamSynchronize(sb.st_kkt);
            return fail(h, CIMPC_ERR_STATE, "newton_solve: round limit reached with unfinished rollouts");
        }
        while (launched <= completed + (ahead ? 1 : 0)) {
            rc = launch_round(launched, launched > completed);
            if (rc != CIMPC_OK) return rc;
            ++launched;
        }
        {   // the decision kernel's last block stamps the mapped flag when round `completed` is done
            volatile int* hm = (volatile int*)h->h_ring + 8 * (completed & 1);
            const int want = (int)(completed + 1);
            long long spins = 0;
            while (hm[2] != want) {
                if ((++spins & 0xFFFFF) == 0 && hipStreamQuery(sb.st) != hipErrorNotReady && hm[2] != want)
                    return fail(h, CIMPC_ERR_HIP, "round finished without publishing its counters");
            }
        }
        volatile int* hr = (volatile int*)h->h_ring + 8 * (completed & 1);
        const int n_sweep = hr[0];
        last_sweep = n_sweep;
        last_kkt = hr[1];
        last_slots = hr[6];
        last_new_slots = hr[7];      // (deterministic: slots that only wait for a parked solve are re-listed, not re-requested)
        last_parked = hr[4];
        if (dbg_rounds) fprintf(stderr, "[cimpc round %lld] t %.3f ms: next sweep %d rollouts (%d evaluation slots), next kkt %d, parked %d, finished %d\n", completed,
                                1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), n_sweep, last_slots, last_kkt, hr[4], hr[5]);
        h->prof_kkt_systems += last_kkt;
        ++completed;
        rounds = completed;
        if ((n_sweep == 0 && last_kkt == 0) || over_budget()) break;   // newton.jl:187-277: budget ends silently
        if (hybrid) {
            // sparse tail: few rollouts left, every round pays its fixed latency for them -> the persistent
            // kernel finishes them along their own chains
            const int active = h->dm.B - hr[5];
            if (active > 0 && active <= h->async_tail) {
                HIP_TRY(h, hipStreamSynchronize(sb.st));
                HIP_TRY(h, hipStreamSynchronize(sb.st_kkt));
                return run_async(false, rounds, (int)(rounds & 1));
            }
        }
    }
    HIP_TRY(h, hipStreamSynchronize(sb.st_kkt));
    if (int rf = finish_results(sb.st); rf != CIMPC_OK) return rf;
    HIP_TRY(h, hipStreamSynchronize(sb.st));
    const long long st[4] = {(long long)h->h_result[0], (long long)h->h_result[1], (long long)h->h_result[2], (long long)h->h_result[3]};
    h->last_stats.sweeps = st[0];
    h->last_stats.ip_solves = st[1];
    h->last_stats.ip_iters = st[2];
    h->last_stats.ip_failures = st[3];
    h->last_stats.rounds = rounds;
    h->last_stats.newton_iters = (long long)h->h_result[4];
    h->prof_ip_problems += st[1];
    return CIMPC_OK;
}

int cimpc_newton_solve(cimpc_handle h, const double* q0, const double* q1, int warm_start,
                       double* u1, int* newton_iters, double* r_norm) {
    if (!h || !q0 || !q1) return fail(h, CIMPC_ERR_INVALID, "null argument");
    const size_t ne = (size_t)h->dm.B * h->dm.nq;
    HIP_TRY(h, hipSetDevice(h->device));
    // [q0 | q1] through the pinned staging buffer: one asynchronous copy on the caller's stream (the solve's streams wait for it
    // on the device), and the results come back in the one block the solve leaves in pinned memory
    std::memcpy(h->h_qin, q0, ne * sizeof(double));
    std::memcpy(h->h_qin + ne, q1, ne * sizeof(double));
    HIP_TRY(h, hipMemcpyAsync(h->d_q0, h->h_qin, 2 * ne * sizeof(double), hipMemcpyHostToDevice, h->stream));
    int rc = cimpc_newton_solve_dev(h, h->d_q0, h->d_q1, warm_start);
    if (rc != CIMPC_OK) return rc;
    const int nu = h->dm.nu, w = nu + 2;
    for (int b = 0; b < h->dm.B; ++b) {
        const double* o = h->h_result + 8 + (size_t)b * w;
        if (u1) std::memcpy(u1 + (size_t)b * nu, o, nu * sizeof(double));
        if (newton_iters) newton_iters[b] = (int)o[nu];
        if (r_norm) r_norm[b] = o[nu + 1] / (double)h->N;
    }
    return CIMPC_OK;
}

// ---- B4 in ONE call (round 6): what the drop-in under the reference's unchanged policy() does per MPC step - window, reference,
// altitude in, newton_solve!, core.traj / core.nu out (policy.jl:113-142) - without the five synchronous calls and their blocking
// pageable copies: every input goes through ONE pinned staging block and asynchronous copies on the handle's stream (the solve's
// streams wait for that stream on the device), the window is re-keyed only when it differs from the one last uploaded, and the
// trajectory comes back through the same block with one synchronisation.
int cimpc_mpc_solve(cimpc_handle h, const int* window, const double* q_ref, const double* u_ref, const double* w_ref,
                    const double* gamma_ref, const double* b_ref, const double* theta_ref, const double* alt,
                    const double* q0, const double* q1, int warm_start, double* u1, int* newton_iters, double* r_norm,
                    double* q, double* u, double* gamma, double* b, double* nu_dual) {
    if (!h || !q0 || !q1) return fail(h, CIMPC_ERR_INVALID, "null argument");
    if (q_ref && (!u_ref || !theta_ref)) return fail(h, CIMPC_ERR_INVALID, "q_ref, u_ref, theta_ref go together");
    const cimpc_dims& d = h->dm;
    const size_t B = d.B, H = d.H, K = d.H_ref;
    const size_t nwin = B * (H + 2);
    const size_t n_q = B * (H + 2) * d.nq, n_u = B * H * d.nu, n_w = B * H * d.nw, n_g = B * H * d.nc, n_b = B * H * d.nb,
                 n_th = B * H * h->nth, n_nu = B * H * h->nd, n_alt = B * d.nc;
    HIP_TRY(h, hipSetDevice(h->device));
    if (!h->h_stage) {
        // [window (as doubles' worth of ints) | q | u | w | gamma | b | theta | alt]; the way out reuses [q | u | gamma | b | nu]
        h->h_stage_doubles = (nwin + 1) / 2 + n_q + n_u + n_w + n_g + n_b + std::max(n_th, n_nu) + n_alt + 8;
        if (hipHostMalloc((void**)&h->h_stage, h->h_stage_doubles * sizeof(double), hipHostMallocDefault) != hipSuccess)
            return fail(h, CIMPC_ERR_HIP, "hipHostMalloc failed");
    }
    if (int dp = drain_pending(h); dp != CIMPC_OK) return dp;
    double* stage = h->h_stage;
    int* s_win = reinterpret_cast<int*>(stage);
    double* s_q = stage + (nwin + 1) / 2; double* s_u = s_q + n_q; double* s_w = s_u + n_u; double* s_g = s_w + n_w;
    double* s_b = s_g + n_g; double* s_th = s_b + n_b; double* s_alt = s_th + std::max(n_th, n_nu);
    hipStream_t st = h->stream;
    if (window) {
        bool same = h->win_mirror.size() == nwin;
        for (size_t k = 0; k < nwin; ++k) {
            const int t = window[k];
            if (t < 1 || t > (int)K) return fail(h, CIMPC_ERR_INVALID, "window entry out of range (1-based knot index)");
            s_win[k] = t - 1;
            same = same && h->win_mirror[k] == t - 1;
        }
        if (!same) {      // the per-knot sensitivity memory moves with the window (cimpc_set_window), stream-ordered
            if (int rk = rekey_out(h); rk != CIMPC_OK) return rk;
            HIP_TRY(h, hipMemcpyAsync(h->d_window, s_win, nwin * sizeof(int), hipMemcpyHostToDevice, st));
            if (int rk = rekey_in(h); rk != CIMPC_OK) return rk;
            h->win_mirror.assign(s_win, s_win + nwin);
        }
        h->window_set = true;
        h->gait_set = false;
    }
    if (q_ref) {
        auto up = [&](double* dst, double* stg, const double* src, size_t n) -> hipError_t {
            if (n == 0) return hipSuccess;
            if (src) std::memcpy(stg, src, n * sizeof(double)); else std::memset(stg, 0, n * sizeof(double));
            return hipMemcpyAsync(dst, stg, n * sizeof(double), hipMemcpyHostToDevice, st);
        };
        HIP_TRY(h, up(h->S.ref.q, s_q, q_ref, n_q));
        HIP_TRY(h, up(h->S.ref.u, s_u, u_ref, n_u));
        HIP_TRY(h, up(h->S.ref.w, s_w, w_ref, n_w));                       // NULL = zeros, as cimpc_set_reference
        if (gamma_ref) HIP_TRY(h, up(h->S.ref.g, s_g, gamma_ref, n_g));   // NULL = unchanged, as cimpc_set_reference
        if (b_ref) HIP_TRY(h, up(h->S.ref.b, s_b, b_ref, n_b));
        HIP_TRY(h, up(h->S.ref.th, s_th, theta_ref, n_th));
        h->reference_set = true;
        h->gait_set = false;
    }
    if (alt) {
        std::memcpy(s_alt, alt, n_alt * sizeof(double));
        HIP_TRY(h, hipMemcpyAsync(h->d_alt, s_alt, n_alt * sizeof(double), hipMemcpyHostToDevice, st));
        h->alt_set = true;
    }
    int rc = cimpc_newton_solve(h, q0, q1, warm_start, u1, newton_iters, r_norm);
    if (rc != CIMPC_OK) return rc;
    if (q || u || gamma || b || nu_dual) {
        // (the solve has completed on the library's streams; these copies run on the handle's stream, one synchronisation)
        auto down = [&](double* stg, const double* src, double* dst, size_t n) -> hipError_t {
            return (dst && n) ? hipMemcpyAsync(stg, src, n * sizeof(double), hipMemcpyDeviceToHost, st) : hipSuccess;
        };
        HIP_TRY(h, down(s_q, h->S.traj.q, q, n_q));
        HIP_TRY(h, down(s_u, h->S.traj.u, u, n_u));
        HIP_TRY(h, down(s_g, h->S.traj.g, gamma, n_g));
        HIP_TRY(h, down(s_b, h->S.traj.b, b, n_b));
        HIP_TRY(h, down(s_th, h->S.nu, nu_dual, n_nu));
        HIP_TRY(h, hipStreamSynchronize(st));
        if (q) std::memcpy(q, s_q, n_q * sizeof(double));
        if (u) std::memcpy(u, s_u, n_u * sizeof(double));
        if (gamma && n_g) std::memcpy(gamma, s_g, n_g * sizeof(double));
        if (b && n_b) std::memcpy(b, s_b, n_b * sizeof(double));
        if (nu_dual) std::memcpy(nu_dual, s_th, n_nu * sizeof(double));
    }
    return CIMPC_OK;
}

int cimpc_get_trajectory(cimpc_handle h, double* q, double* u, double* gamma, double* b, double* nu_dual) {
    if (!h) return CIMPC_ERR_INVALID;
    const cimpc_dims& d = h->dm;
    const size_t B = d.B, H = d.H;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (q) HIP_TRY(h, hipMemcpy(q, h->S.traj.q, B * (H + 2) * d.nq * sizeof(double), hipMemcpyDeviceToHost));
    if (u) HIP_TRY(h, hipMemcpy(u, h->S.traj.u, B * H * d.nu * sizeof(double), hipMemcpyDeviceToHost));
    if (gamma) HIP_TRY(h, hipMemcpy(gamma, h->S.traj.g, B * H * d.nc * sizeof(double), hipMemcpyDeviceToHost));
    if (b) HIP_TRY(h, hipMemcpy(b, h->S.traj.b, B * H * d.nb * sizeof(double), hipMemcpyDeviceToHost));
    if (nu_dual) HIP_TRY(h, hipMemcpy(nu_dual, h->S.nu, B * H * h->nd * sizeof(double), hipMemcpyDeviceToHost));
    return CIMPC_OK;
}

int cimpc_get_newton_info(cimpc_handle h, int* newton_iters, double* r_norm, double* u1) {
    if (!h) return CIMPC_ERR_INVALID;
    const cimpc_dims& d = h->dm;
    const size_t B = d.B;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (newton_iters) HIP_TRY(h, hipMemcpy(newton_iters, h->S.newton_l, B * sizeof(int), hipMemcpyDeviceToHost));
    if (r_norm) {
        HIP_TRY(h, hipMemcpy(r_norm, h->S.r_norm, B * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t b = 0; b < B; ++b) r_norm[b] /= (double)h->N;
    }
    if (u1)     // core.traj.u[1] of every rollout (policy.jl:142)
        HIP_TRY(h, hipMemcpy2D(u1, d.nu * sizeof(double), h->S.traj.u, (size_t)d.H * d.nu * sizeof(double),
                               d.nu * sizeof(double), B, hipMemcpyDeviceToHost));
    return CIMPC_OK;
}

int cimpc_get_rollout_counters(cimpc_handle h, int* sweeps, int* ip_iters, int* ip_failures) {
    if (!h) return CIMPC_ERR_INVALID;
    const size_t B = h->dm.B;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (sweeps) HIP_TRY(h, hipMemcpy(sweeps, h->S.ro_sweeps, B * sizeof(int), hipMemcpyDeviceToHost));
    if (ip_iters) HIP_TRY(h, hipMemcpy(ip_iters, h->S.ro_ip_iters, B * sizeof(int), hipMemcpyDeviceToHost));
    if (ip_failures) HIP_TRY(h, hipMemcpy(ip_failures, h->S.ro_ip_fail, B * sizeof(int), hipMemcpyDeviceToHost));
    return CIMPC_OK;
}

static GaitDev gait_dev(cimpc_ctx* h) {
    return GaitDev{h->g_q, h->g_u, h->g_w, h->g_g, h->g_b, h->g_th, h->g_stride, h->g_phase, h->dm.H_ref};
}

int cimpc_set_gait(cimpc_handle h, const double* q, const double* u, const double* w, const double* gamma,
                   const double* b, const double* theta, const double* stride, const int* phase) {
    if (!h || !q || !u || !theta || !stride) return fail(h, CIMPC_ERR_INVALID, "q, u, theta, stride are required");
    const cimpc_dims& d = h->dm;
    const size_t K = d.H_ref, B = d.B;
    if (d.H > d.H_ref) return fail(h, CIMPC_ERR_INVALID, "H_mpc must not exceed H_ref");
    if (phase) for (size_t i = 0; i < B; ++i) if (phase[i] < 0) return fail(h, CIMPC_ERR_INVALID, "negative phase");
    HIP_TRY(h, hipSetDevice(h->device));
    if (int dp = drain_pending(h); dp != CIMPC_OK) return dp;
    if (!h->g_q) {
        if (dev_alloc(h, &h->g_q, (K + 2) * d.nq) != CIMPC_OK || dev_alloc(h, &h->g_u, K * d.nu) != CIMPC_OK ||
            dev_alloc(h, &h->g_w, K * d.nw) != CIMPC_OK || dev_alloc(h, &h->g_g, K * d.nc) != CIMPC_OK ||
            dev_alloc(h, &h->g_b, K * d.nb) != CIMPC_OK || dev_alloc(h, &h->g_th, K * h->nth) != CIMPC_OK ||
            dev_alloc(h, &h->g_stride, d.nq) != CIMPC_OK || dev_alloc(h, &h->g_phase, B) != CIMPC_OK)
            return CIMPC_ERR_HIP;                    // (dev_alloc zero-fills: w, gamma, b default to 0)
    }
    HIP_TRY(h, hipMemcpy(h->g_q, q, (K + 2) * d.nq * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->g_u, u, K * d.nu * sizeof(double), hipMemcpyHostToDevice));
    auto opt = [&](double* dst, const double* src, size_t n) {      // NULL = zeros (also when an earlier call gave values)
        return src ? hipMemcpy(dst, src, n * sizeof(double), hipMemcpyHostToDevice) : hipMemset(dst, 0, n * sizeof(double));
    };
    HIP_TRY(h, opt(h->g_w, w, K * d.nw));
    HIP_TRY(h, opt(h->g_g, gamma, K * d.nc));
    HIP_TRY(h, opt(h->g_b, b, K * d.nb));
    HIP_TRY(h, hipStreamSynchronize(nullptr));
    HIP_TRY(h, hipMemcpy(h->g_th, theta, K * h->nth * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->g_stride, stride, d.nq * sizeof(double), hipMemcpyHostToDevice));
    if (phase) HIP_TRY(h, hipMemcpy(h->g_phase, phase, B * sizeof(int), hipMemcpyHostToDevice));
    else { HIP_TRY(h, hipMemset(h->g_phase, 0, B * sizeof(int))); HIP_TRY(h, hipStreamSynchronize(nullptr)); }
    if (int rk = rekey_out(h); rk != CIMPC_OK) return rk;
    int rc = launch_gait_window(h->S, gait_dev(h), h->d_window, 0, h->stream);
    if (rc != CIMPC_OK) return fail(h, rc, "gait window launch failed");
    if (int rk = rekey_in(h); rk != CIMPC_OK) return rk;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->gait_set = h->window_set = h->reference_set = true;
    h->win_mirror.clear();             // the window now lives on the device only
    return CIMPC_OK;
}

int cimpc_mpc_advance(cimpc_handle h, const double* stride) {
    if (!h || !stride) return fail(h, CIMPC_ERR_INVALID, "null argument");
    if (!h->window_set || !h->reference_set) return fail(h, CIMPC_ERR_STATE, "set_window / set_reference have not been called");
    HIP_TRY(h, hipSetDevice(h->device));
    // The advance is QUEUED on the handle's stream and not waited for: whatever comes next is ordered behind it on the device
    // (the solve's streams wait for this stream, every getter synchronises it first).  The stride goes through a pinned word;
    // a second advance in a row waits for the first one's copy to have left it.
    if (int dp = drain_pending(h); dp != CIMPC_OK) return dp;
    h->win_mirror.clear();
    double* st_pin = h->h_qin + 2 * (size_t)h->dm.B * h->dm.nq;
    std::memcpy(st_pin, stride, (size_t)h->dm.nq * sizeof(double));
    if (h->gait_set) {          // full reference resident: regenerate the horizon from the gait (stride as given now)
        HIP_TRY(h, hipMemcpyAsync(h->g_stride, st_pin, (size_t)h->dm.nq * sizeof(double), hipMemcpyHostToDevice, h->stream));
        if (int rk = rekey_out(h); rk != CIMPC_OK) return rk;
        int rcg = launch_gait_window(h->S, gait_dev(h), h->d_window, 1, h->stream);
        if (rcg != CIMPC_OK) return fail(h, rcg, "mpc_advance launch failed");
        if (int rk = rekey_in(h); rk != CIMPC_OK) return rk;
        h->advance_pending = true;
        return CIMPC_OK;
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_q0, stride, (size_t)h->dm.nq * sizeof(double), hipMemcpyHostToDevice, h->stream));   // d_q0: staging
    if (int rk = rekey_out(h); rk != CIMPC_OK) return rk;
    int rc = launch_mpc_advance(h->S, h->d_window, h->d_q0, h->dm.H_ref, h->stream);
    if (rc != CIMPC_OK) return fail(h, rc, "mpc_advance launch failed");
    if (int rk = rekey_in(h); rk != CIMPC_OK) return rk;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return CIMPC_OK;
}

int cimpc_get_reference(cimpc_handle h, double* q_ref, double* u_ref, double* w_ref, double* gamma_ref,
                        double* b_ref, double* theta_ref, int* window) {
    if (!h) return CIMPC_ERR_INVALID;
    const cimpc_dims& d = h->dm;
    const size_t B = d.B, H = d.H;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (q_ref) HIP_TRY(h, hipMemcpy(q_ref, h->S.ref.q, B * (H + 2) * d.nq * sizeof(double), hipMemcpyDeviceToHost));
    if (u_ref) HIP_TRY(h, hipMemcpy(u_ref, h->S.ref.u, B * H * d.nu * sizeof(double), hipMemcpyDeviceToHost));
    if (w_ref) HIP_TRY(h, hipMemcpy(w_ref, h->S.ref.w, B * H * d.nw * sizeof(double), hipMemcpyDeviceToHost));
    if (gamma_ref) HIP_TRY(h, hipMemcpy(gamma_ref, h->S.ref.g, B * H * d.nc * sizeof(double), hipMemcpyDeviceToHost));
    if (b_ref) HIP_TRY(h, hipMemcpy(b_ref, h->S.ref.b, B * H * d.nb * sizeof(double), hipMemcpyDeviceToHost));
    if (theta_ref) HIP_TRY(h, hipMemcpy(theta_ref, h->S.ref.th, B * H * h->nth * sizeof(double), hipMemcpyDeviceToHost));
    if (window) {
        HIP_TRY(h, hipMemcpy(window, h->d_window, B * (H + 2) * sizeof(int), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < B * (H + 2); ++k) window[k] += 1;     // 1-based at the boundary
    }
    return CIMPC_OK;
}

int cimpc_get_newton_log(cimpc_handle h, double* log, int max_entries) {
    if (!h || !log || max_entries <= 0) return CIMPC_ERR_INVALID;
    const size_t B = h->dm.B;
    const int n = std::min(max_entries, NLOG);
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy2D(log, (size_t)max_entries * 4 * sizeof(double), h->S.nlog, (size_t)NLOG * 4 * sizeof(double),
                           (size_t)n * 4 * sizeof(double), B, hipMemcpyDeviceToHost));
    return CIMPC_OK;
}

int cimpc_get_kkt_twisted(cimpc_handle h, long long* n) {
    if (!h || !n) return CIMPC_ERR_INVALID;
    *n = h->n_kkt_twisted;
    return CIMPC_OK;
}

int cimpc_get_kkt_twisted_fallbacks(cimpc_handle h, long long* n) {
    if (!h || !n) return CIMPC_ERR_INVALID;
    const int cur = *(volatile int*)h->h_twfail;
    if (cur != h->twfail_seen) { h->n_kkt_tw_fallbacks += cur - h->twfail_seen; h->twfail_seen = cur; }
    *n = h->n_kkt_tw_fallbacks;
    return CIMPC_OK;
}

int cimpc_debug_set_tw_spins(cimpc_handle h, int spins) {
    if (!h) return CIMPC_ERR_INVALID;
    h->S.kkt_tw_spins = spins > 0 ? spins : (1 << 21);
    return CIMPC_OK;
}

int cimpc_get_kkt_fallbacks(cimpc_handle h, long long* n) {
    if (!h || !n) return CIMPC_ERR_INVALID;
    *n = 0;
    if (!h->d_mix_nfb) return CIMPC_OK;
    int v = 0;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(&v, h->d_mix_nfb, sizeof(int), hipMemcpyDeviceToHost));
    *n = v;
    return CIMPC_OK;
}

#if defined(CIMPC_KKT_PROF) || defined(CIMPC_KKT_TWPROF)
// diagnostic builds only: raw read of the statistics buffer (the KKT kernels park their phase clocks at [8..23])
int cimpc_debug_read_stats(cimpc_handle h, long long* out, int n) {
    if (!h || !out || n > std::max(h->dm.B * 4, 64)) return CIMPC_ERR_INVALID;
    HIP_TRY(h, hipMemcpy(out, h->S.stats, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost));
    return CIMPC_OK;
}
#endif

#ifdef CIMPC_DEBUG_STATS
// diagnostic builds only: the raw per-rollout statistics records + the device addresses of the neighbouring allocations
int cimpc_debug_dump_stats(cimpc_handle h, long long* out, unsigned long long* addr8) {
    if (!h || !out || !addr8) return CIMPC_ERR_INVALID;
    HIP_TRY(h, hipMemcpy(out, h->S.stats, (size_t)h->dm.B * 4 * sizeof(long long), hipMemcpyDeviceToHost));
    const void* a[8] = {h->S.kkt_list, h->S.slot_list, h->S.counters, h->S.stats, h->S.ro_sweeps, h->S.ro_ip_iters, h->S.ro_ip_fail, h->S.nlog};
    for (int k = 0; k < 8; ++k) addr8[k] = (unsigned long long)a[k];
    return CIMPC_OK;
}
#endif

int cimpc_get_stats(cimpc_handle h, cimpc_stats* s) {
    if (!h || !s) return CIMPC_ERR_INVALID;
    *s = h->last_stats;
    return CIMPC_OK;
}

int cimpc_profile_enable(cimpc_handle h, int on) {
    if (!h) return CIMPC_ERR_INVALID;
    prof_collect(h);
    h->prof_on = on == 2 ? 2 : (on != 0 ? 1 : 0);
    return CIMPC_OK;
}

int cimpc_profile_reset(cimpc_handle h) {
    if (!h) return CIMPC_ERR_INVALID;
    prof_collect(h);
    for (int c = 0; c < PC_COUNT; ++c) { h->prof_ms[c] = 0.0; h->prof_n[c] = 0; }
    h->prof_ip_problems = 0;
    h->prof_async_problems = 0;
    h->prof_kkt_systems = 0;
    return CIMPC_OK;
}

int cimpc_profile_read(cimpc_handle h, cimpc_profile* p) {
    if (!h || !p) return CIMPC_ERR_INVALID;
    prof_collect(h);
    p->ip_sweep_ms = h->prof_ms[PC_IP]; p->ip_sweep_launches = h->prof_n[PC_IP];
    p->ip_sweep_problems = h->prof_ip_problems;
    p->kkt_ms = h->prof_ms[PC_KKT]; p->kkt_launches = h->prof_n[PC_KKT]; p->kkt_systems = h->prof_kkt_systems;
    p->resid_ms = h->prof_ms[PC_RESID]; p->resid_launches = h->prof_n[PC_RESID];
    p->other_ms = h->prof_ms[PC_OTHER]; p->other_launches = h->prof_n[PC_OTHER];
    p->async_ms = h->prof_ms[PC_ASYNC]; p->async_launches = h->prof_n[PC_ASYNC]; p->async_problems = h->prof_async_problems;
    return CIMPC_OK;
}

int cimpc_query_sizes(cimpc_handle h, int* table_doubles, int* N_kkt) {
    if (!h) return CIMPC_ERR_INVALID;
    if (table_doubles) *table_doubles = h->ki.tab_size;
    if (N_kkt) *N_kkt = h->N;
    return CIMPC_OK;
}

}  // extern "C"

#ifdef CIMPC_UBENCH
// diagnostic builds only: the device address of the handle's linearization tables (scripts/dbg/ubench_ip.py)
extern "C" const double* cimpc_debug_table_ptr(cimpc_handle h) { return h ? h->d_tab : nullptr; }
#endif
