#!/usr/bin/env python3
"""bench.py - MPC steps/s of the MI355X solver core on BASELINE.json's headline workload.

A "step" = one `newton_solve!` (cold start) for every rollout of the batch: quadruped
dimensions (nq 11, nu 8, nw 2, nc 4, nb 8), mode :configuration, H = 40, H_ref = 60,
fp64, B Monte-Carlo rollouts per GPU (synthetic horizons, seeded - SURVEY.md section 8d).
`value` = rollouts * steps / wall time, inputs resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W [--rollouts B]
For N > 1 the driver launches one rank per GPU with torch.distributed.run; rollouts are
independent, so ranks never exchange data inside the solve (weak scaling: B per GPU fixed).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

QUADRUPED = dict(nq=11, nu=8, nw=2, nc=4, nb=8)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # SURVEY.md section 8d (fp64 vector/matrix nominal)


def algorithmic_sizes(nq, nu, nw, nc, nb, mode=0):
    """algorithmic doubles per IP solve and flops per solve."""
    nx, ny = nq, 2 * nc + nb
    nz, nth = nq + 4 * nc + 2 * nb, 2 * nq + nu + nw + 2
    nd = nq if mode == 0 else nq + nc + nb
    n_lin = nx * nx + 2 * nx * ny + ny * ny + ny + (nx + ny) * nth + nx * nx + nx * ny + ny * ny \
        + (nx + 2 * ny + nth) + (nx + ny)
    io_shared = nth + nq + nz + nd * (2 * nq + nu)
    flop_iter = 2 * ny ** 3 + ny * ny + 2 * (4 * nx * ny + 3 * ny * ny + 2 * nx * nx) + 3 * 2 * (nx + ny) * (nx + ny + nth)
    flop_tail = 2 * ny ** 3 + ny * ny + nth * (4 * nx * ny + 3 * ny * ny + 2 * nx * nx)
    return dict(n_lin=n_lin, bytes_per_solve=8 * (n_lin + io_shared), bytes_per_solve_shared_table=8 * io_shared,
                flop_iter=flop_iter, flop_tail=flop_tail)


def build_inputs(B, H, H_ref, seed, perturb, first=0):
    from contactimplicitmpc.jl_amd import synthetic as synth
    from contactimplicitmpc.jl_amd.trajectory import Dims
    d = Dims(**QUADRUPED)
    prob = synth.make_problem(d, H_ref, seed=1)            # shared linearization table (all ranks)
    obj = synth.make_objective(d, H)
    rollouts = []
    for g in range(first, first + B):     # g = GLOBAL rollout index: the batch does not depend on the sharding
        phase = int(np.random.default_rng(seed * 7919 + g).integers(0, H_ref))
        rollouts.append(synth.make_rollout(d, prob, H, phase=phase, seed=seed * 100003 + g, perturb=perturb))
    return d, prob, obj, rollouts


def cpu_baseline(d, prob, obj, rollouts, H, H_ref, budget_s=12.0):
    """C restatement of the reference algorithm (oracle/cimpc_ref.c) on a bounded sample of the same
    rollouts.  Single thread (the headline CPU number), two KKT backends as the reference offers both:
    condensed/banded (the :ldl_solver analogue) and dense LU (the default :lu_solver); then all host
    cores, one solver instance per thread over independent rollouts (ctypes releases the GIL)."""
    import threading
    from oracle import ip as oip, newton as onewton
    from oracle.cref import CRef

    def make():
        return CRef(d, H_ref, H, prob, obj, oip.IPOptions(kappa_tol=prob["kappa"]),
                    onewton.NewtonOptions(r_tol=3e-4, max_iter=5), prob["kappa"])
    cr = make()
    out = {}
    for name, solver, share in (("condensed", 1, 0.45), ("dense_lu", 0, 0.3)):
        t0 = time.perf_counter()
        n = 0
        while n < len(rollouts) and (time.perf_counter() - t0) < budget_s * share:
            window, ref, q0, q1 = rollouts[n]
            cr.newton_solve(window, ref, q0, q1, solver=solver)
            n += 1
        dt = time.perf_counter() - t0
        out[name] = (n / dt, n)
    nthreads = max(1, min(os.cpu_count() or 1, len(rollouts)))
    solvers = [make() for _ in range(nthreads)]
    counts = [0] * nthreads
    stop_at = time.perf_counter() + budget_s * 0.25

    def work(k):
        i = k
        while i < len(rollouts) and time.perf_counter() < stop_at:
            window, ref, q0, q1 = rollouts[i]
            solvers[k].newton_solve(window, ref, q0, q1, solver=1)
            counts[k] += 1
            i += nthreads
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    out["all_cores"] = (sum(counts) / dt, sum(counts), nthreads)
    return out


def mpc_loop_latency(dims, kind, H, H_ref, device, steps=40, mode=0):
    """Warm-started single-rollout MPC loop: newton_solve! -> rot_n_stride! / update_window! -> q0 <- q1."""
    from contactimplicitmpc.jl_amd import synthetic as synth
    from contactimplicitmpc.jl_amd.trajectory import Dims
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
    d = Dims(**dims, mode=mode)
    prob = synth.make_problem(d, H_ref, seed=1)
    obj = synth.make_objective(d, H, kind=kind)
    window, ref, q0, q1 = synth.make_rollout(d, prob, H, phase=0, seed=7, perturb=0.01)
    s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=1, mode=mode, ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]),
                    newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-4, max_iter=5), device=device)
    for t in range(H_ref):
        s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
    if mode == 1:
        s.set_objective(obj.q, obj.u, obj.gamma, obj.b)      # :configurationforce: contact-impulse weights 1e-100
    else:
        s.set_objective(obj.q, obj.u)
    stride = np.zeros(d.nq)
    stride[0] = prob["q_ref"][-2][0] - prob["q_ref"][0][0]          # get_stride, mpc_utils.jl:103-107
    # the controller's full reference trajectory lives on the device (cimpc_set_gait); mpc_advance rotates it
    s.set_gait(prob["q_ref"], prob["u_ref"], prob["th0"], stride, w=prob["w_ref"], gamma=prob["gamma_ref"], b=prob["b_ref"])
    a, b = q0[None].copy(), q1[None].copy()
    its = 0
    t0 = None
    for k in range(steps + 3):
        if k == 3:
            t0 = time.perf_counter()
            its = 0
        u1, it, rn = s.newton_solve(a, b, warm_start=k > 0)
        its += int(it[0])
        nxt = s.trajectory()["q"][:, 2].copy()
        s.mpc_advance(stride)
        a, b = b, nxt
    dt = time.perf_counter() - t0
    return {"ms_per_mpc_step": 1e3 * dt / steps, "mpc_steps_per_s": steps / dt, "newton_iters_per_step": its / steps}


def real_mpc_loop_latency(H, device, steps=60, perturb=0.02):
    """The reference's own use (policy.jl:98-146 cadence, no plant): ONE robot on the real quadruped gait2 problem,
    warm-started newton_solve! every step, reference / window advanced on the device, next state = planned q_3.
    Starts from a perturbed configuration, so the first steps need Newton iterations and the loop then settles."""
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions, gait_io, lcp_models
    m = lcp_models.Quadruped()
    kappa = 2e-4
    P = lcp_models.reference_problem(m, gait_io.load_gait(os.path.join(ROOT, "tests", "golden", "gaits", "quadruped_gait2.jld2")), kappa)
    r = lcp_models.make_rollout(P, H, 0, seed=3, perturb=perturb)
    qd = 1e-2 * np.concatenate([[1.0, 0.02, 0.25], 0.25 * np.ones(m.nq - 3)])
    s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=1, mode=0, ip_opts=InteriorPointOptions(kappa_tol=kappa),
                    newton_opts=NewtonOptions(kappa=kappa, r_tol=3e-4, max_iter=5), device=device)
    for t in range(P.H):
        s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
    s.set_objective(np.tile(np.diag(qd)[None], (H, 1, 1)), np.tile((3e-2 * np.eye(m.nu))[None], (H, 1, 1)))
    stride = lcp_models.get_stride(m, P.q)
    s.set_gait(P.q, P.u, P.theta, stride, w=P.w, gamma=P.gamma, b=P.b)
    a, b = r["q0"][None].copy(), r["q1"][None].copy()
    its, hist = 0, []
    t0 = None
    for k in range(steps + 3):
        if k == 3:
            t0 = time.perf_counter()
            its = 0
        u1, it, rn = s.newton_solve(a, b, warm_start=k > 0)
        its += int(it[0]); hist.append(int(it[0]))
        nxt = s.trajectory()["q"][:, 2].copy()
        s.mpc_advance(stride)
        a, b = b, nxt
    dt = time.perf_counter() - t0
    s.close()
    return {"ms_per_mpc_step": 1e3 * dt / steps, "mpc_steps_per_s": steps / dt, "newton_iters_per_step": its / steps,
            "newton_iters_first_steps": hist[:8]}


def real_problem_leg(B, H, device, steps=5, perturb=0.05):
    """The same Monte-Carlo batch on the REAL quadruped problem: the reference's gait file (gait2.jld2, a data file
    of the reference kept under tests/golden/gaits) linearized through the model restatement of
    contactimplicitmpc/jl_amd/lcp_models.py, objective of test/controller/mpc_quadruped.jl:23-27, kappa_mpc = 2e-4,
    initial configurations perturbed by U(-perturb, perturb) (examples/quadruped/monte_carlo.jl:79-91)."""
    import torch
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions, gait_io, lcp_models
    m = lcp_models.Quadruped()
    kappa = 2e-4
    t0 = time.perf_counter()
    P = lcp_models.reference_problem(m, gait_io.load_gait(os.path.join(ROOT, "tests", "golden", "gaits", "quadruped_gait2.jld2")), kappa)
    t_lin = time.perf_counter() - t0
    ro = [lcp_models.make_rollout(P, H, int(np.random.default_rng(7919 + g).integers(0, P.H)), seed=100003 + g, perturb=perturb)
          for g in range(B)]
    qd = 1e-2 * np.concatenate([[1.0, 0.02, 0.25], 0.25 * np.ones(m.nq - 3)])
    s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=kappa),
                    newton_opts=NewtonOptions(kappa=kappa, r_tol=3e-4, max_iter=5), device=device)
    for t in range(P.H):
        s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
    s.set_objective(np.tile(np.diag(qd)[None], (H, 1, 1)), np.tile((3e-2 * np.eye(m.nu))[None], (H, 1, 1)))
    s.set_window(np.stack([r["window"] for r in ro]) + 1)
    s.set_reference(*(np.stack([r[k] for r in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")))
    q0 = torch.tensor(np.stack([r["q0"] for r in ro]), dtype=torch.float64, device="cuda")
    q1 = torch.tensor(np.stack([r["q1"] for r in ro]), dtype=torch.float64, device="cuda")
    s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), warm_start=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = dict(newton_iters=0, sweeps=0, ip_solves=0, ip_iters=0, ip_failures=0)
    for _ in range(steps):
        s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), warm_start=False)
        st = s.stats()
        for k in acc:
            acc[k] += st[k]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    s.close()
    return {"workload": "quadruped gait2.jld2 (reference data file), true model linearization, H=%d, %d rollouts, "
                        "U(-%.2f, %.2f) initial-configuration offsets, cold start" % (H, B, perturb, perturb),
            "value": B / dt, "unit": "MPC steps/s", "ms_per_step": 1e3 * dt,
            "newton_iters_per_step": acc["newton_iters"] / (steps * B), "sweeps_per_step": acc["sweeps"] / (steps * B),
            "ip_iters_per_solve": acc["ip_iters"] / max(acc["ip_solves"], 1), "ip_failures_per_step": acc["ip_failures"] / steps,
            "linearization_build_s": t_lin}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rollouts", type=int, default=512, help="Monte-Carlo rollouts per GPU")
    ap.add_argument("--horizon", type=int, default=40)
    ap.add_argument("--perturb", type=float, default=0.05)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency", action="store_true", help="also time the B = 1 single-rollout loop")
    ap.add_argument("--no-real-problem", action="store_true", help="skip the leg on the real quadruped gait")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
    H, H_ref, B = args.horizon, 60, args.rollouts
    from contactimplicitmpc.jl_amd.sharding import rollout_shard
    first, count = rollout_shard(world * B, rank, world)      # weak scaling: B rollouts per GPU
    d, prob, obj, rollouts = build_inputs(count, H, H_ref, seed=1234, perturb=args.perturb, first=first)

    def make(Bn, ro):
        s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=Bn, mode=0,
                        ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]),
                        newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-4, max_iter=5), device=local_rank)
        for t in range(H_ref):
            s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
        s.set_objective(obj.q, obj.u)
        s.set_window(np.stack([w for (w, _, _, _) in ro]) + 1)
        s.set_reference(np.stack([r.q for (_, r, _, _) in ro]), np.stack([r.u for (_, r, _, _) in ro]),
                        np.stack([r.w for (_, r, _, _) in ro]), np.stack([r.gamma for (_, r, _, _) in ro]),
                        np.stack([r.b for (_, r, _, _) in ro]), np.stack([r.theta for (_, r, _, _) in ro]))
        q0 = torch.tensor(np.stack([r[2] for r in ro]), dtype=torch.float64, device="cuda")
        q1 = torch.tensor(np.stack([r[3] for r in ro]), dtype=torch.float64, device="cuda")
        return s, q0, q1

    s, q0, q1 = make(B, rollouts)
    torch.cuda.synchronize()

    def step():
        s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), warm_start=False)

    for _ in range(args.warmup):
        step()
    s.profile_enable(not os.environ.get("CIMPC_BENCH_NOPROF"))
    s.profile_reset()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    sweeps = ip_solves = ip_iters = newton_iters = rounds = 0
    for _ in range(args.steps):
        step()
        st = s.stats()
        sweeps += st["sweeps"]; ip_solves += st["ip_solves"]; ip_iters += st["ip_iters"]
        newton_iters += st["newton_iters"]; rounds += st["rounds"]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = s.profile_read()
    s.profile_enable(False)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    alg = algorithmic_sizes(**QUADRUPED)
    K = ip_iters / max(ip_solves, 1)
    L = newton_iters / (B * args.steps)
    S = sweeps / max(newton_iters + B * args.steps, 1)
    ip_ms = prof["ip_sweep_ms"]
    launches = max(prof["ip_sweep_launches"], 1)
    solves_per_launch = prof["ip_sweep_problems"] / launches
    avg_launch_ms = ip_ms / launches
    bytes_per_launch = solves_per_launch * alg["bytes_per_solve"]
    achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if ip_ms > 0 else 0.0
    flops_per_solve = K * alg["flop_iter"] + alg["flop_tail"]
    tflops = solves_per_launch * flops_per_solve / (avg_launch_ms * 1e-3) / 1e12 if ip_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "ip_sweep_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            if tj.get("rollouts") == B and tj.get("horizon") == H:
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "MPC steps/s (quadruped, H=40, fp64)",
        "value": world * B * args.steps / dt,
        "unit": "MPC steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "quadruped flat_2D_lc dims, :configuration, H=%d, H_ref=%d, fp64, "
                               "%d Monte-Carlo rollouts per GPU batch-sharded (BASELINE configs[3]; "
                               "configs[2] = same problem at B=1, see latency_b1)" % (H, H_ref, B),
                   "rollouts_per_gpu": B, "horizon": H, "perturb": args.perturb,
                   "newton": "r_tol 3e-4, max_iter 5, cold start", "ip": "r_tol 1e-8, kappa_tol 2e-4, undercut 5"},
        "solver_iters": {"newton_iters_per_step": L, "ip_iters_per_solve": K, "sweeps_per_eval": S,
                         "sweeps_per_step": sweeps / (B * args.steps), "lockstep_rounds_per_step": rounds / args.steps},
        "roofline": {"bound": "hbm", "kernel": "ip_queue_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "bytes_per_unit": alg["bytes_per_solve"], "units_per_launch": solves_per_launch,
                     "avg_launch_ms": avg_launch_ms,
                     "achieved_shared_table": solves_per_launch * alg["bytes_per_solve_shared_table"] / (avg_launch_ms * 1e-3) / 1e9 if ip_ms > 0 else 0.0,
                     "fp64_tflops": tflops, "fp64_vector_peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                     "fp64_frac": tflops / FP64_VECTOR_PEAK_TFLOPS},
        "kernel_time_ms_per_step": {"ip_sweep": prof["ip_sweep_ms"] / args.steps, "kkt": prof["kkt_ms"] / args.steps,
                                    "resid": prof["resid_ms"] / args.steps, "other": prof["other_ms"] / args.steps,
                                    "async_tail": prof["async_ms"] / args.steps},
        "schedule": {"lockstep_rounds_per_step": rounds / args.steps - (1 if prof["async_launches"] else 0),
                     "async_tail_launches_per_step": prof["async_launches"] / args.steps,
                     "ip_problems_in_rounds": prof["ip_sweep_problems"] / args.steps,
                     "ip_problems_in_async_tail": prof["async_problems"] / args.steps},
    }
    if args.latency:
        s1, a0, a1 = make(1, rollouts[:1])
        for _ in range(3):
            s1.newton_solve_dev(a0.data_ptr(), a1.data_ptr(), False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n1 = 20
        for _ in range(n1):
            s1.newton_solve_dev(a0.data_ptr(), a1.data_ptr(), False)
        torch.cuda.synchronize()
        out["latency_b1"] = {"ms_per_step": 1e3 * (time.perf_counter() - t1) / n1, "stats": s1.stats()}
        # the reference's own use: ONE robot, warm-started MPC steps in a loop (policy.jl:98-146 cadence without the
        # plant: q1_next = planned q_3), reference / window advanced on the device (cimpc_mpc_advance)
        out["mpc_loop_b1"] = {"quadruped_h40": mpc_loop_latency(QUADRUPED, "quadruped", 40, 60, local_rank),
                              "hopper_h20 (BASELINE configs[1])": mpc_loop_latency(dict(nq=4, nu=2, nw=2, nc=1, nb=2), "hopper", 20, 30, local_rank),
                              "pushbot_h10_configurationforce (BASELINE configs[0])": mpc_loop_latency(dict(nq=2, nu=2, nw=2, nc=2, nb=4), "pushbot", 10, 16, local_rank, mode=1),
                              "quadruped_h40_real_gait2": real_mpc_loop_latency(40, local_rank)}
    if not args.no_real_problem and world == 1:      # informative second workload (N = 1 only), outside the timed region
        try:
            out["real_problem"] = real_problem_leg(B, H, local_rank)
        except Exception as e:
            out["real_problem"] = {"error": repr(e)}
    if not args.no_cpu_baseline and world == 1:      # the CPU baseline is a rank-0, N = 1 measurement
        try:
            cb = cpu_baseline(d, prob, obj, rollouts, H, H_ref)
            out["cpu_baseline"] = {
                "value": cb["condensed"][0], "unit": "MPC steps/s", "cores": 1, "kind": "port",
                "sample": "%d rollouts (condensed block-Cholesky KKT, the :ldl_solver analogue) and %d rollouts "
                          "(dense LU, reference default :lu_solver) of the same batch, oracle/cimpc_ref.c, 1 thread"
                          % (cb["condensed"][1], cb["dense_lu"][1]),
                "value_dense_lu": cb["dense_lu"][0],
                "value_all_cores": cb["all_cores"][0], "all_cores_threads": cb["all_cores"][2],
                "all_cores_sample": "%d rollouts, condensed KKT, one solver instance per thread" % cb["all_cores"][1],
                "host_cores_available": os.cpu_count()}
            out["speedup_vs_cpu_1thread"] = out["value"] / cb["condensed"][0]
        except Exception as e:  # the baseline never blocks the GPU number
            out["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
