#!/usr/bin/env python3
"""bench.py - MPC steps/s of the MI355X solver core on BASELINE.json's headline workload.

A "step" = one `newton_solve!` (cold start) for every rollout of the batch: quadruped
dimensions (nq 11, nu 8, nw 2, nc 4, nb 8), mode :configuration, H = 40, H_ref = 60,
fp64, B Monte-Carlo rollouts per GPU (synthetic horizons, seeded - SURVEY.md section 8d).
`value` = rollouts * steps / wall time, inputs resident in HBM before the timed region.

    python bench.py --gpus N --steps K --warmup W [--rollouts B] [--scaling weak|strong]
For N > 1 the driver launches one rank per GPU with torch.distributed.run.  Rank 0 builds the shared problem and
BROADCASTS the packed linearization tables / gait / objective over RCCL, every rank solves its contiguous shard of
the global rollout indices (no collective inside a solve), and ONE all-gather per reporting interval returns
[u1 | newton_iters | r_norm | sweeps] of every rollout (contactimplicitmpc/jl_amd/monte_carlo.py; the reference's
workload: examples/quadruped/monte_carlo.jl:76-92).  --scaling weak (default): B rollouts PER GPU;
--scaling strong: B rollouts in total, split over the ranks (BASELINE configs[3]: 512 over 8 GPUs).
Rank 0 prints ONE JSON line.
"""
import argparse
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

QUADRUPED = dict(nq=11, nu=8, nw=2, nc=4, nb=8)
KKT_FLOP_PER_SOLVE = 9.9e6    # one KKT solve of the quadruped H = 40 system, banded-interleaved count
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # SURVEY.md section 8d (fp64 vector/matrix nominal)


def kkt_condensed_flops(nq, H):
    """Contract flops of ONE condensed (structure-solver) KKT solve, SURVEY.md section 8 row A13: per step
    1/3 n^3 + 2 n^3 + ... with n = 2 nq, = 30 k at the quadruped's n = 22 (1.2 MFLOP for H = 40)."""
    return H * 2.8 * (2 * nq) ** 3


def kkt_banded_flops(nq, nu, H, reduced=True):
    """Contract flops of ONE banded L D L^T KKT solve, SURVEY.md section 8(d): N w^2 + 4 N w in the interleaved ordering - of the
    form the kernel factors: controls eliminated first (s = nq + nd rows per step, w = 3 s - 1), :configuration mode (nd = nq)."""
    sblk = 2 * nq if reduced else 2 * nq + nu
    N = H * sblk
    w = 3 * sblk - 1 - (0 if reduced else nu)
    return N * w * w + 4 * N * w


def kernel_roofline(name, flops_per_unit, units, launches, ms, note="", bound=None, flops_executed=None):
    """One entry of roofline.kernels: achieved = contract flops of the units a launch processes / its average duration (HIP events
    around the launches of that kernel class in a profiled pass of the same workload), against the fp64 peak (vector = MFMA)."""
    if not launches or ms <= 0:
        return {"kernel": name, "error": "no launches recorded"}
    avg_ms = ms / launches
    tf = flops_per_unit * (units / launches) / (avg_ms * 1e-3) / 1e12
    # `bound`: the pipe the kernel issues its flops on, stated by the caller (a name match labelled the banded LDL^T - DPP
    # triangular solves + MFMA bulk tiles - like the all-MFMA condensed kernels)
    out = {"kernel": name, "bound": bound or ("fp64_mfma" if "kkt" in name else "fp64_valu"), "achieved": tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": tf / FP64_VECTOR_PEAK_TFLOPS, "avg_launch_ms": avg_ms, "units_per_launch": units / launches, "flops_per_unit": flops_per_unit, "note": note}
    if flops_executed is not None:      # the device executes fewer flops than the contract counts (adjoint sensitivity pass)
        out["flops_executed_per_unit"] = flops_executed
        out["frac_executed"] = out["frac"] * flops_executed / flops_per_unit
    return out


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
    # what the device's sensitivity pass EXECUTES (round 5): 2 nq + nu columns instead of the reference's nth; in :configuration mode
    # the adjoint form - nx solves with M^T (3 ny^2 each) and the product K0 + A2 Gs (2 nx ny per column)
    nths = 2 * nq + nu
    flop_tail_exec = 2 * ny ** 3 + ny * ny + (nx * 3 * ny * ny + nths * 2 * nx * ny if mode == 0 else nths * (4 * nx * ny + 3 * ny * ny + 2 * nx * nx))
    return dict(flop_tail_executed=flop_tail_exec, n_lin=n_lin, bytes_per_solve=8 * (n_lin + io_shared), bytes_per_solve_shared_table=8 * io_shared,
                flop_iter=flop_iter, flop_tail=flop_tail)


def build_problem(H, H_ref):
    """The shared part of the batch: per-knot linearization tables + reference gait + objective (rank 0 only for N > 1)."""
    from contactimplicitmpc.jl_amd import synthetic as synth
    from contactimplicitmpc.jl_amd.trajectory import Dims
    d = Dims(**QUADRUPED)
    return d, synth.make_problem(d, H_ref, seed=1), synth.make_objective(d, H)


def build_rollouts(d, prob, B, H, H_ref, seed, perturb, first=0):
    from contactimplicitmpc.jl_amd import synthetic as synth
    rollouts = []
    for g in range(first, first + B):     # g = GLOBAL rollout index: the batch does not depend on the sharding
        phase = int(np.random.default_rng(seed * 7919 + g).integers(0, H_ref))
        rollouts.append(synth.make_rollout(d, prob, H, phase=phase, seed=seed * 100003 + g, perturb=perturb))
    return rollouts


def build_inputs(B, H, H_ref, seed, perturb, first=0):
    d, prob, obj = build_problem(H, H_ref)
    return d, prob, obj, build_rollouts(d, prob, B, H, H_ref, seed, perturb, first)


def csrc_hash():
    """Identity of the kernels a measurement belongs to: sha256 over the library sources."""
    hsh = hashlib.sha256()
    base = os.path.join(ROOT, "contactimplicitmpc", "jl_amd", "csrc")
    for f in sorted(os.listdir(base)):
        if f.endswith((".h", ".hip", ".cpp")):
            hsh.update(f.encode())
            hsh.update(open(os.path.join(base, f), "rb").read())
    return hsh.hexdigest()[:16]


def measure_sweep_traffic(B, H, timeout_s=150):
    """HBM bytes per `ip_queue_kernel` launch from the PMC counters, measured IN THIS RUN: two `rocprofv3 --pmc` passes
    (FETCH_SIZE and WRITE_SIZE do not fit one pass; PMC passes carry no trace domain - MI355X_MICROARCH.md, HBM section)
    over a short child run of this same script; bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts
    128-byte requests as 64 bytes).  Returns (bytes_per_launch or None, source string)."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--rollouts", str(B), "--horizon", str(H),
             "--no-cpu-baseline", "--no-real-problem", "--no-latency", "--no-traffic", "--no-centroidal"]
    vals = {}
    tmp = tempfile.mkdtemp(prefix="cimpc_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            subprocess.run([exe, "--pmc", ctr, "--output-format", "csv", "-d", out, "-o", "p", "--"] + child, cwd="/tmp", env=env,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            import csv
            tot = n = 0
            for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(path)):
                    if "ip_queue_kernel" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                        tot += float(r["Counter_Value"]); n += 1
            if n == 0:
                return None, "no ip_queue_kernel dispatches in the %s pass" % ctr
            vals[ctr] = tot / n
        return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, "rocprofv3 --pmc, measured in this run (2 passes, child run of 2 steps)"
    except Exception as e:      # the counters never block the bench line
        return None, "rocprofv3 pass failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


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


def cpu_single_rollout(d, prob, obj, rollout, H, H_ref, reps=3):
    """The CPU restatement (oracle/cimpc_ref.c, one thread, condensed KKT) on ONE rollout - the same one `latency_b1` solves on the GPU."""
    from oracle import ip as oip, newton as onewton
    from oracle.cref import CRef
    cr = CRef(d, H_ref, H, prob, obj, oip.IPOptions(kappa_tol=prob["kappa"]), onewton.NewtonOptions(r_tol=3e-4, max_iter=5), prob["kappa"])
    window, ref, q0, q1 = rollout
    cr.newton_solve(window, ref, q0, q1, solver=1)
    t0 = time.perf_counter()
    for _ in range(reps):
        cr.newton_solve(window, ref, q0, q1, solver=1)
    return (time.perf_counter() - t0) / reps


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
        nxt = s.trajectory(which=("q",))["q"][:, 2].copy()
        s.mpc_advance(stride)
        a, b = b, nxt
    dt = time.perf_counter() - t0
    return {"ms_per_mpc_step": 1e3 * dt / steps, "mpc_steps_per_s": steps / dt, "newton_iters_per_step": its / steps}


def real_mpc_loop_latency(H, device, steps=60, perturb=0.02, which="quadruped"):
    """The reference's own use (policy.jl:98-146 cadence, no plant): ONE robot on a real reference problem - quadruped gait2
    (test/controller/mpc_quadruped.jl:19-47) or hopper_2D gait_forward (examples/hopper/flat.jl:15-47: a `:joint_traj` file,
    linearized at its own z / θ as the reference does) - warm-started newton_solve! every step, reference / window advanced on
    the device, next state = planned q_3.  Starts from a perturbed configuration, so the first steps need Newton iterations and
    the loop then settles."""
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions, gait_io, lcp_models
    kappa = 2e-4
    gaits = os.path.join(ROOT, "tests", "golden", "gaits")
    if which == "hopper":
        m = lcp_models.Hopper2D()
        P = lcp_models.reference_problem_from_traj(m, gait_io.load_joint_traj(os.path.join(gaits, "hopper_gait_forward.jld2")), kappa)
        qd, ud = 0.1 * np.array([0.1, 3.0, 1.0, 3.0]), np.array([1e-3, 1.0])                      # examples/hopper/flat.jl:30-34
    else:
        m = lcp_models.Quadruped()
        P = lcp_models.reference_problem(m, gait_io.load_gait(os.path.join(gaits, "quadruped_gait2.jld2")), kappa)
        qd, ud = 1e-2 * np.concatenate([[1.0, 0.02, 0.25], 0.25 * np.ones(m.nq - 3)]), 3e-2 * np.ones(m.nu)
    r = lcp_models.make_rollout(P, H, 0, seed=3, perturb=perturb)
    s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=1, mode=0, ip_opts=InteriorPointOptions(kappa_tol=kappa),
                    newton_opts=NewtonOptions(kappa=kappa, r_tol=3e-4, max_iter=5), device=device)
    for t in range(P.H):
        s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
    s.set_objective(np.tile(np.diag(qd)[None], (H, 1, 1)), np.tile(np.diag(ud)[None], (H, 1, 1)))
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
        nxt = s.trajectory(which=("q",))["q"][:, 2].copy()
        s.mpc_advance(stride)
        a, b = b, nxt
    dt = time.perf_counter() - t0
    s.close()
    return {"ms_per_mpc_step": 1e3 * dt / steps, "mpc_steps_per_s": steps / dt, "newton_iters_per_step": its / steps,
            "newton_iters_first_steps": hist[:8]}


def dropin_host_path_latency(H, device, steps=60, perturb=0.02):
    """What a user of the B4 drop-in gets (julia/CIMPCHip.jl: the method of the package's own newton_solve! on
    Newton{..., LS = HipMPCSolver}, reached from the reference's UNCHANGED policy(), policy.jl:113-142): the exact C-call sequence
    of the binding, through ctypes with HOST arrays, on the same real gait2 loop as `mpc_loop_b1.quadruped_h40_real_gait2` - the
    window and the rotated reference (p.traj, rotated by the policy on the host) arrive from the host at every step.
      one_call   round 6: cimpc_mpc_solve(window, reference, alt, q0, q1 -> u1, q, u, nu)
      five_calls round 5: cimpc_set_altitude + cimpc_set_window + cimpc_set_reference + cimpc_newton_solve + cimpc_get_trajectory
    The knot stamps of the binding (a hash of 60 x (nz + ntheta + nz) doubles per call) are the Julia host's own work and not timed here."""
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions, gait_io, lcp_models
    m = lcp_models.Quadruped()
    kappa = 2e-4
    P = lcp_models.reference_problem(m, gait_io.load_gait(os.path.join(ROOT, "tests", "golden", "gaits", "quadruped_gait2.jld2")), kappa)
    qd = 1e-2 * np.concatenate([[1.0, 0.02, 0.25], 0.25 * np.ones(m.nq - 3)])
    r0 = lcp_models.make_rollout(P, H, 0, seed=3, perturb=perturb)
    host = [lcp_models.make_rollout(P, H, k, seed=0, perturb=0.0) for k in range(steps + 3)]      # the policy's rotated p.traj / p.window per step
    alt = np.zeros((1, m.nc))
    out = {}
    for kind in ("one_call", "five_calls"):
        s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=1, mode=0, ip_opts=InteriorPointOptions(kappa_tol=kappa),
                        newton_opts=NewtonOptions(kappa=kappa, r_tol=3e-4, max_iter=5), device=device)
        for t in range(P.H):
            s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
        s.set_objective(np.tile(np.diag(qd)[None], (H, 1, 1)), np.tile((3e-2 * np.eye(m.nu))[None], (H, 1, 1)))
        a, b = r0["q0"][None].copy(), r0["q1"][None].copy()
        its, t0 = 0, None
        for k in range(steps + 3):
            if k == 3:
                t0 = time.perf_counter(); its = 0
            hk = host[k]
            win = hk["window"][None] + 1
            ref = {f: hk[f][None] for f in ("q", "u", "w", "gamma", "b", "theta")}
            if kind == "one_call":
                u1, it, rn, tr = s.mpc_solve(a, b, window=win, reference=ref, alt=alt, warm_start=k > 0, which=("q", "u", "nu"))
            else:
                s.set_altitude(alt)
                s.set_window(win)
                s.set_reference(ref["q"], ref["u"], ref["w"], ref["gamma"], ref["b"], ref["theta"])
                u1, it, rn = s.newton_solve(a, b, warm_start=k > 0)
                tr = s.trajectory(which=("q", "u", "nu"))
            its += int(it[0])
            a, b = b, tr["q"][:, 2].copy()
        dt = time.perf_counter() - t0
        s.close()
        out[kind] = {"ms_per_mpc_step": 1e3 * dt / steps, "newton_iters_per_step": its / steps}
    return out


def real_problem_inputs(B, H, perturb=0.05):
    """Inputs of `real_problem_leg` (also the full-size parity test's, tests/test_gpu_round3_parity.py): the reference's
    quadruped gait2.jld2 linearized through the model restatement, objective of test/controller/mpc_quadruped.jl:23-27,
    kappa_mpc = 2e-4, initial configurations perturbed by U(-perturb, perturb) (examples/quadruped/monte_carlo.jl:79-91)."""
    from contactimplicitmpc.jl_amd import gait_io, lcp_models
    m = lcp_models.Quadruped()
    kappa = 2e-4
    t0 = time.perf_counter()
    P = lcp_models.reference_problem(m, gait_io.load_gait(os.path.join(ROOT, "tests", "golden", "gaits", "quadruped_gait2.jld2")), kappa)
    t_lin = time.perf_counter() - t0
    ro = [lcp_models.make_rollout(P, H, int(np.random.default_rng(7919 + g).integers(0, P.H)), seed=100003 + g, perturb=perturb)
          for g in range(B)]
    qd = 1e-2 * np.concatenate([[1.0, 0.02, 0.25], 0.25 * np.ones(m.nq - 3)])
    return dict(m=m, P=P, kappa=kappa, rollouts=ro, Q=np.tile(np.diag(qd)[None], (H, 1, 1)),
                R=np.tile((3e-2 * np.eye(m.nu))[None], (H, 1, 1)), linearization_build_s=t_lin)


def real_problem_leg(B, H, device, steps=5, perturb=0.05):
    """The same Monte-Carlo batch on the REAL quadruped problem: the reference's gait file (gait2.jld2, a data file
    of the reference kept under tests/golden/gaits) linearized through the model restatement of
    contactimplicitmpc/jl_amd/lcp_models.py, objective of test/controller/mpc_quadruped.jl:23-27, kappa_mpc = 2e-4,
    initial configurations perturbed by U(-perturb, perturb) (examples/quadruped/monte_carlo.jl:79-91)."""
    import torch
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
    I = real_problem_inputs(B, H, perturb)
    m, P, kappa, ro, t_lin = I["m"], I["P"], I["kappa"], I["rollouts"], I["linearization_build_s"]
    s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=kappa),
                    newton_opts=NewtonOptions(kappa=kappa, r_tol=3e-4, max_iter=5), device=device)
    for t in range(P.H):
        s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
    s.set_objective(I["Q"], I["R"])
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
    _, _, rn = s.newton_info()
    s.close()
    return {"workload": "quadruped gait2.jld2 (reference data file), true model linearization, H=%d, %d rollouts, "
                        "U(-%.2f, %.2f) initial-configuration offsets, cold start" % (H, B, perturb, perturb),
            "value": B / dt, "unit": "MPC steps/s", "ms_per_step": 1e3 * dt,
            "converged_rollouts": int((rn < 3e-4).sum()), "rollouts": B,
            "newton_iters_per_step": acc["newton_iters"] / (steps * B), "sweeps_per_step": acc["sweeps"] / (steps * B),
            "ip_iters_per_solve": acc["ip_iters"] / max(acc["ip_solves"], 1), "ip_failures_per_step": acc["ip_failures"] / steps,
            "linearization_build_s": t_lin}


def centroidal_payload_inputs(B, H):
    """Inputs of `centroidal_payload_leg` = BASELINE configs[4] (also the parity test's, tests/test_gpu_round3_parity.py):
    examples/centroidal_quadruped/reference/inplace_trot_v7.jld2 through the model restatement, continuous_trot.jl:37-73 settings
    (kappa = 1e-3, IP r_tol 1e-4, Newton r_tol 3e-5; tracking part of its objective), a constant downward body force
    w_z = -5 .. -30 N per rollout (the payload: src/dynamics/centroidal_quadruped/model.jl:121-125, continuous_trot.jl:80-81)."""
    from contactimplicitmpc.jl_amd import gait_io, lcp_models
    m = lcp_models.CentroidalQuadruped()
    kappa = 1e-3
    P = lcp_models.reference_problem(m, gait_io.load_gait(os.path.join(ROOT, "tests", "golden", "gaits", "centroidal_inplace_trot_v7.jld2")), kappa)
    rng = np.random.default_rng(5)
    ro = []
    for g in range(B):
        r = lcp_models.make_rollout(P, H, int(np.random.default_rng(g).integers(0, P.H)), seed=100 + g, perturb=0.02)
        r["w"][:] = np.array([0.0, 0.0, -rng.uniform(5.0, 30.0)])            # payload: constant downward body force
        r["theta"][:, 2 * m.nq + m.nu:2 * m.nq + m.nu + m.nw] = r["w"]
        ro.append(r)
    Q = np.tile(lcp_models.relative_state_cost([1.0, 1, 1], 3e-1 * np.ones(3), [0.2, 0.2, 1.0])[None], (H, 1, 1))
    R = np.tile((3e-3 * np.eye(m.nu))[None], (H, 1, 1))
    return dict(m=m, P=P, kappa=kappa, rollouts=ro, Q=Q, R=R, ip_r_tol=1e-4, newton_r_tol=3e-5)


def centroidal_velocity_objective(m, H):
    """The objective of the example as written, examples/centroidal_quadruped/continuous_trot.jl:43-47: a TrackingVelocityObjective
    whose Q alone is singular along a common x shift (body-x weight 0), V = 1e-3 diag(1,1,1, 1e3,1e3,1e3, 1 x 12), v_target = 0
    (v0 = -0.0), R = 3e-3 I.  Returns (Q, R, V, v_target) as (H, n, n) / (H, nq) arrays."""
    from contactimplicitmpc.jl_amd import lcp_models
    Q = np.tile(lcp_models.relative_state_cost([0.0, 1, 1], 3e-1 * np.ones(3), [0.2, 0.2, 1.0])[None], (H, 1, 1))
    R = np.tile((3e-3 * np.eye(m.nu))[None], (H, 1, 1))
    V = np.tile(np.diag(1e-3 * np.concatenate([np.ones(3), 1e3 * np.ones(3), np.ones(12)]))[None], (H, 1, 1))
    return Q, R, V, np.zeros((H, m.nq))


def centroidal_velocity_leg(I, B, H, device, steps=2):
    """BASELINE configs[4] with the example's OWN objective (`centroidal_velocity_objective`): P is block tridiagonal, the KKT
    stage is the banded L D L^T of the interleaved ordering - the configuration is KKT-bound."""
    import torch
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
    m, P, kappa, ro = I["m"], I["P"], I["kappa"], I["rollouts"]
    Q, R, V, vt = centroidal_velocity_objective(m, H)
    s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=kappa, r_tol=1e-4),
                    newton_opts=NewtonOptions(kappa=kappa, r_tol=3e-5, max_iter=5), device=device)
    for t in range(P.H):
        s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
    s.set_objective(Q, R, V=V, v_target=vt)
    s.set_window(np.stack([r["window"] for r in ro]) + 1)
    s.set_reference(*(np.stack([r[k] for r in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")))
    q0 = torch.tensor(np.stack([r["q0"] for r in ro]), dtype=torch.float64, device="cuda")
    q1 = torch.tensor(np.stack([r["q1"] for r in ro]), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
    s.profile_enable(True); s.profile_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    pr = s.profile_read(); st = s.stats()
    _, it, rn = s.newton_info()
    s.close()
    return {"objective": "TrackingVelocityObjective of continuous_trot.jl:43-47 (Q singular along x, V, v_target = 0)", "kkt": "banded L D L^T (interleaved ordering; twisted: two workgroups per rollout)",
            "value": B / dt, "unit": "MPC steps/s", "ms_per_step": 1e3 * dt, "newton_iters_per_step": float(it.mean()),
            "converged_rollouts": int((rn < 3e-5).sum()), "kkt_ms_per_step": pr["kkt_ms"] / steps, "ip_sweep_ms_per_step": pr["ip_sweep_ms"] / steps,
            "kkt_systems_per_step": pr["kkt_systems"] / steps, "kkt_launches_per_step": pr["kkt_launches"] / steps, "ip_failures": st["ip_failures"]}


def centroidal_closed_loop_leg(B, H, device, mpc_steps=24, N_sample=5):
    """BASELINE configs[4] the way the example runs it (examples/centroidal_quadruped/continuous_trot.jl:37-81): warm-started MPC in
    closed loop - policy AND plant on the device, B robots that carry payloads the controller does not know about (a constant body
    force of -5 .. -30 N), the example's own TrackingVelocityObjective, kappa 1e-3, IP r_tol 1e-4, Newton r_tol 3e-5 / max_iter 5,
    N_sample = 5 - with the horizon BASELINE names (H = 60; the example uses 50).  Reports what "solver iterations to tolerance" means
    for this config: share of (robot, solve) pairs that end below r_tol, Newton iterations per solve, ms per MPC step (the
    newton_solve! of all B robots, host API, synchronous) with the cold first solve kept apart."""
    from contactimplicitmpc.jl_amd import InteriorPointOptions, NewtonOptions, gait_io, lcp_models, plant
    from contactimplicitmpc.jl_amd.policy import CIMPCPolicy
    kappa = 1e-3
    m = lcp_models.CentroidalQuadruped()
    P = lcp_models.reference_problem(m, gait_io.load_gait(os.path.join(ROOT, "tests", "golden", "gaits", "centroidal_inplace_trot_v7.jld2")), kappa)
    Q, R, V, vt = centroidal_velocity_objective(m, H)
    pol = CIMPCPolicy(P, Q, R, H_mpc=H, N_sample=N_sample, B=B, mode=0, obj_v=V, v_target=vt, device=device,
                      n_opts=NewtonOptions(kappa=kappa, r_tol=3e-5, max_iter=5), ip_opts=InteriorPointOptions(kappa_tol=kappa, r_tol=1e-4, undercut=5.0))
    rec = []
    inner = pol.solver.newton_solve

    def timed(q0, q1, warm_start=False):
        t0 = time.perf_counter()
        u1, it, rn = inner(q0, q1, warm_start=warm_start)
        rec.append((time.perf_counter() - t0, it.copy(), rn.copy()))
        return u1, it, rn
    pol.solver.newton_solve = timed
    h_sim = P.h / N_sample
    rng = np.random.default_rng(5)
    w_const = np.zeros((B, 3)); w_const[:, 2] = -rng.uniform(5.0, 30.0, B) * h_sim          # impulse per simulator step
    q1 = np.tile(P.q[1], (B, 1)); v1 = np.tile((P.q[1] - P.q[0]) / P.h, (B, 1))
    t0 = time.perf_counter()
    ok, q, u, g, b = plant.simulate("centroidal_quadruped", pol, q1, v1, mpc_steps * N_sample, h_sim, mu=m.mu_world, disturbances=lambda t: w_const)
    wall = time.perf_counter() - t0
    tw = pol.solver.kkt_twisted()
    pol.close()
    warm = rec[1:]
    its = np.stack([r[1] for r in warm]); rns = np.stack([r[2] for r in warm])
    ref_h = P.q[:, 2].mean()
    return {"workload": "continuous_trot.jl:37-81 closed loop, %d robots with payloads -5..-30 N unknown to the controller, H_mpc=%d, N_sample=%d, "
                        "TrackingVelocityObjective (banded L D L^T KKT), policy + plant on the device, %d MPC steps (the first, cold one reported apart)"
                        % (B, H, N_sample, mpc_steps),
            "ms_per_mpc_step_warm": 1e3 * float(np.mean([r[0] for r in warm])), "ms_first_cold_step": 1e3 * rec[0][0],
            "value_warm": B / float(np.mean([r[0] for r in warm])), "unit": "MPC steps/s",
            "newton_iters_per_solve_warm": float(its.mean()), "newton_iters_first_cold_step": float(rec[0][1].mean()),
            "converged_share_warm (r_norm < 3e-5)": float((rns < 3e-5).mean()), "r_norm_median_warm": float(np.median(rns)),
            "all_plant_steps_converged": bool(ok), "body_height_sag_max_m": float(np.abs(q[:, :, 2] - ref_h).max()),
            "closed_loop_wall_s (policy + plant + host glue)": wall, "kkt_twisted_launches": tw}


def centroidal_b1_leg(H, device, n=10):
    """BASELINE configs[4] at B = 1 (SURVEY.md 8(d) cfg 5 lists "B = 1 and 512 / 8 GPUs"): ONE robot with a payload, cold-start
    newton_solve!, the tracking objective (condensed f64-MFMA KKT: twisted, two workgroups) and the example's velocity objective
    (banded L D L^T)."""
    import torch
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
    I = centroidal_payload_inputs(1, H)
    m, P, kappa, ro = I["m"], I["P"], I["kappa"], I["rollouts"]
    out = {}
    for name, velocity in (("tracking_objective", False), ("velocity_objective", True)):
        s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=1, mode=0, ip_opts=InteriorPointOptions(kappa_tol=kappa, r_tol=1e-4),
                        newton_opts=NewtonOptions(kappa=kappa, r_tol=3e-5, max_iter=5), device=device)
        for t in range(P.H):
            s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
        if velocity:
            Q, R, V, vt = centroidal_velocity_objective(m, H)
            s.set_objective(Q, R, V=V, v_target=vt)
        else:
            s.set_objective(I["Q"], I["R"])
        s.set_window(np.stack([r["window"] for r in ro]) + 1)
        s.set_reference(*(np.stack([r[k] for r in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")))
        q0 = torch.tensor(np.stack([r["q0"] for r in ro]), dtype=torch.float64, device="cuda")
        q1 = torch.tensor(np.stack([r["q1"] for r in ro]), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        for _ in range(2):
            s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
        s.profile_enable(True); s.profile_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        pr = s.profile_read()
        _, it, rn = s.newton_info()
        out[name] = {"ms_per_step": 1e3 * dt, "mpc_steps_per_s": 1.0 / dt, "newton_iters": int(it[0]), "r_norm": float(rn[0]),
                     "kkt_ms_per_step": pr["kkt_ms"] / n, "kkt_launches_per_step": pr["kkt_launches"] / n, "ip_sweep_ms_per_step": pr["ip_sweep_ms"] / n,
                     "resid_ms_per_step": pr["resid_ms"] / n, "kkt_twisted_launches": s.kkt_twisted()}
        s.close()
    out["workload"] = "centroidal_quadruped inplace_trot_v7.jld2, ONE robot with a payload, H=%d, cold-start newton_solve!" % H
    return out


def centroidal_payload_leg(B, H, device, steps=3):
    """BASELINE configs[4]: centroidal_quadruped with a payload (body-force disturbance w in theta:
    src/dynamics/centroidal_quadruped/model.jl:121-125, continuous_trot.jl:80-81), H = 60, on the REAL problem
    (examples/centroidal_quadruped/reference/inplace_trot_v7.jld2 through the model restatement, continuous_trot.jl:37-73
    settings: kappa = 1e-3, IP r_tol 1e-4, Newton r_tol 3e-5; tracking part of its objective).  Timed twice: KKT in mixed
    precision (Schur-block products on the fp32 MFMA + fp64 refinement + fp64 fallback) and in fp64.  B = the per-GPU
    share of 512 rollouts on 8 GPUs."""
    import torch
    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
    I = centroidal_payload_inputs(B, H)
    m, P, kappa, ro, Q, R = I["m"], I["P"], I["kappa"], I["rollouts"], I["Q"], I["R"]
    out = {"workload": "centroidal_quadruped inplace_trot_v7.jld2 (reference data file), payload w_z = -5..-30 N per rollout, "
                       "H=%d, %d rollouts (512 / 8 GPUs), kappa 1e-3, cold start" % (H, B),
           # which leg IS configs[4].  BASELINE words the config as "fp32 mixed-precision Schur GEMM on MFMA"; that backend
           # exists, is parity-tested (identical discrete paths on 64 / 64 rollouts, tests/test_gpu_mixed_precision.py) and is timed below -
           # but the KKT recursion is a latency chain, not matrix-throughput-bound, and every refinement pass re-factorises, so it is
           # 2.8x SLOWER than fp64 (a declared non-goal).  The number this library reports for the config is the fp64 leg.
           "reported_leg": "fp64_kkt",
           "reported_leg_reason": "the fp32-MFMA mixed-precision KKT backend (BASELINE's wording, leg mixed_fp32_mfma_kkt) is correct but slower than fp64 on this "
                                  "latency-bound recursion; fp64_kkt is the configuration a user should run and the one quoted for configs[4]; closed-loop "
                                  "figures under centroidal_closed_loop_h60"}
    q0 = torch.tensor(np.stack([r["q0"] for r in ro]), dtype=torch.float64, device="cuda")
    q1 = torch.tensor(np.stack([r["q1"] for r in ro]), dtype=torch.float64, device="cuda")
    u1s = {}
    # three legs: the mixed-precision backend runs lock-step rounds (its refinement is a sequence of launches), so the fp64 backend is
    # timed on the SAME schedule beside it (CIMPC_ASYNC=0 at create) and on the library's default schedule for this batch size
    for name, backend, lockstep in (("mixed_fp32_mfma_kkt", 3, True), ("fp64_kkt_lockstep_rounds", 0, True), ("fp64_kkt", 0, False)):
        prev = os.environ.get("CIMPC_ASYNC")
        if lockstep:
            os.environ["CIMPC_ASYNC"] = "0"
        try:
            s = CIMPCSolver(m.nq, m.nu, m.nw, m.nc, m.nb, P.H, H, B=B, mode=0, ip_opts=InteriorPointOptions(kappa_tol=kappa, r_tol=1e-4),
                            newton_opts=NewtonOptions(kappa=kappa, r_tol=3e-5, max_iter=5, kkt_backend=backend), device=device)
        finally:
            if prev is None:
                os.environ.pop("CIMPC_ASYNC", None)
            else:
                os.environ["CIMPC_ASYNC"] = prev
        for t in range(P.H):
            s.set_linearization(t + 1, P.z[t], P.theta[t], P.r0[t], P.rz0[t], P.rth0[t])
        s.set_objective(Q, R)
        s.set_window(np.stack([r["window"] for r in ro]) + 1)
        s.set_reference(*(np.stack([r[k] for r in ro]) for k in ("q", "u", "w", "gamma", "b", "theta")))
        torch.cuda.synchronize()
        s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
        s.profile_enable(True); s.profile_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        pr = s.profile_read(); st = s.stats()
        u1, it, rn = s.newton_info()
        u1s[name] = u1.copy()
        out[name] = {"value": B / dt, "unit": "MPC steps/s", "ms_per_step": 1e3 * dt, "newton_iters_per_step": float(it.mean()),
                     "schedule": "lock-step rounds" if lockstep else "library default for this batch size (32-lane models: single persistent launch up to 32 rollouts, above it rounds + persistent tail)",
                     "kkt_ms_per_step": pr["kkt_ms"] / steps, "ip_sweep_ms_per_step": pr["ip_sweep_ms"] / steps,
                     "resid_ms_per_step": pr["resid_ms"] / steps, "async_ms_per_step": pr["async_ms"] / steps,
                     "kkt_launches_per_step": pr["kkt_launches"] / steps, "kkt_systems_per_step": pr["kkt_systems"] / steps,
                     "ip_sweep_launches_per_step": pr["ip_sweep_launches"] / steps, "ip_sweep_problems_per_step": pr["ip_sweep_problems"] / steps,
                     "ip_iters_per_solve": st["ip_iters"] / max(st["ip_solves"], 1),
                     "ip_failures": st["ip_failures"], "kkt_systems_since_create": int(it.sum()) * (steps + 1),
                     "kkt_fp64_fallbacks_since_create": s.kkt_fallbacks() if backend == 3 else 0}
        s.close()
    try:      # the example's own objective on the same rollouts (KKT-bound: banded L D L^T)
        out["velocity_objective"] = centroidal_velocity_leg(I, B, H, device)
    except Exception as e:
        out["velocity_objective"] = {"error": repr(e)}
    ref = u1s["fp64_kkt"]
    out["u1_max_abs_diff_vs_fp64_default"] = {k: float(np.abs(v - ref).max()) for k, v in u1s.items() if k != "fp64_kkt"}
    out["u1_scale"] = float(np.abs(ref).max())
    return out


def _free_port():
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks ourselves, exactly as the driver's
    torchrun command would (one process per GPU, rendezvous on 127.0.0.1), and pass our own arguments through.  The
    ranks see WORLD_SIZE = N, so they take the normal N > 1 path; rank 0 prints the one JSON line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def launch_check(args, rank, world, dist, dev):
    """--launch-check: the N-rank control flow of this script WITHOUT a solve (runs on a box with no GPU, gloo): rank 0
    builds the shared problem, broadcast, every rank takes its shard of the global rollout indices, a stand-in result
    that is a function of the global index only goes through the ONE all-gather, every rank checks its rows.  It exists
    so that the launcher (`--gpus N` -> N ranks) is covered by the CPU test suite; it measures nothing."""
    import torch
    from contactimplicitmpc.jl_amd import monte_carlo as mc
    from contactimplicitmpc.jl_amd.sharding import rollout_shard
    from contactimplicitmpc.jl_amd.trajectory import Dims
    H, H_ref = args.horizon, 60
    d = Dims(**QUADRUPED)
    n_global = world * args.rollouts if args.scaling == "weak" else args.rollouts
    if world > 1:
        shapes = mc.problem_shapes(H_ref, H, d.nq, d.nu, d.nw, d.nc, d.nb)
        built = build_problem(H, H_ref) if rank == 0 else None
        prob, obj_q, obj_u = mc.broadcast_problem(*((built[1], built[2].q, built[2].u) if rank == 0 else (None, None, None)), shapes, dev)
    else:
        _, prob, obj = build_problem(H, H_ref)
    first, B = rollout_shard(n_global, rank, world)
    ro = build_rollouts(d, prob, B, H, H_ref, 1234, args.perturb, first=first)
    u1 = np.stack([r[3][:d.nu] for r in ro])                       # stand-in rows: a function of the global rollout index
    it = np.arange(first, first + B)
    rn = np.array([float(np.abs(r[2]).sum()) for r in ro])
    g = mc.gather_results(u1, it, rn, it % 7, n_global, dev)
    ok = bool(np.array_equal(g["u1"][first:first + B], u1) and np.array_equal(g["newton_iters"][first:first + B], it)
              and np.array_equal(g["r_norm"][first:first + B], rn) and np.array_equal(g["newton_iters"], np.arange(n_global)))
    if dist is not None:
        okt = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(okt.item() == 1.0)
    if rank == 0:
        print(json.dumps({"metric": "launch-check (no solve, nothing measured)", "value": None, "n_gpus": world, "launch_check": True,
                          "scaling": args.scaling, "rollouts_total": n_global, "rollouts_per_gpu": B,
                          "multi_gpu": {"gather_selfcheck_all_ranks": ok}, "rollouts_reported": int(g["u1"].shape[0])}))
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rollouts", type=int, default=512, help="Monte-Carlo rollouts per GPU (weak) / in total (strong)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--horizon", type=int, default=40)
    ap.add_argument("--perturb", type=float, default=0.05)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency", action="store_true", help="(default on) the B = 1 legs: BASELINE configs[0..2]")
    ap.add_argument("--no-latency", action="store_true", help="skip the B = 1 single-rollout legs")
    ap.add_argument("--no-real-problem", action="store_true", help="skip the leg on the real quadruped gait")
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-run rocprofv3 --pmc passes (roofline.traffic)")
    ap.add_argument("--no-centroidal", action="store_true", help="skip the centroidal payload leg (BASELINE configs[4])")
    ap.add_argument("--launch-check", action="store_true",
                    help="N-rank control flow only (broadcast, shard, all-gather of stand-in rows; no solve, no GPU needed with "
                         "CIMPC_BENCH_BACKEND=gloo): covers the launcher in the CPU test suite, measures nothing")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare (`python bench.py --gpus N ...`): this process becomes the launcher of the N ranks
        raise SystemExit(self_launch(args.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        # a line whose n_gpus differs from what was asked for would be read as an N-GPU measurement
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE = %d rank(s); refusing to report" % (args.gpus, world))
    if args.launch_check:
        backend = os.environ.get("CIMPC_BENCH_BACKEND", "nccl")
        dist = None
        dev = torch.device("cpu")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dev = torch.device("cuda", local_rank)
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))
        return launch_check(args, rank, world, dist, dev)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if os.environ.get("CIMPC_BENCH_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but only %d device(s) are visible (one rank per GPU over RCCL)" % (world, torch.cuda.device_count()))
    # (CIMPC_BENCH_BACKEND=gloo: validation of the N > 1 control flow on a ONE-GPU box - every rank solves on the one
    #  device, collectives on CPU tensors; the driver's runs use the default: one rank per GPU, RCCL)
    backend = os.environ.get("CIMPC_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from contactimplicitmpc.jl_amd import CIMPCSolver, InteriorPointOptions, NewtonOptions
    from contactimplicitmpc.jl_amd import monte_carlo as mc
    from contactimplicitmpc.jl_amd.sharding import rollout_shard
    from contactimplicitmpc.jl_amd.trajectory import Dims
    H, H_ref = args.horizon, 60
    n_global = world * args.rollouts if args.scaling == "weak" else args.rollouts
    d = Dims(**QUADRUPED)
    t_setup = time.perf_counter()
    if world > 1:       # rank 0 builds the shared problem, RCCL broadcast to the other GPUs (1.4 MB)
        shapes = mc.problem_shapes(H_ref, H, d.nq, d.nu, d.nw, d.nc, d.nb)
        built = build_problem(H, H_ref) if rank == 0 else None
        prob, obj_q, obj_u = mc.broadcast_problem(*((built[1], built[2].q, built[2].u) if rank == 0 else (None, None, None)), shapes, dev)
    else:
        _, prob, obj = build_problem(H, H_ref)
        obj_q, obj_u = obj.q, obj.u
    first, B = rollout_shard(n_global, rank, world)
    rollouts = build_rollouts(d, prob, B, H, H_ref, 1234, args.perturb, first=first)
    t_setup = time.perf_counter() - t_setup

    def make(Bn, ro):
        s = CIMPCSolver(d.nq, d.nu, d.nw, d.nc, d.nb, H_ref, H, B=Bn, mode=0,
                        ip_opts=InteriorPointOptions(kappa_tol=prob["kappa"]),
                        newton_opts=NewtonOptions(kappa=prob["kappa"], r_tol=3e-4, max_iter=5), device=local_rank)
        for t in range(H_ref):
            s.set_linearization(t + 1, prob["z0"][t], prob["th0"][t], prob["r0"][t], prob["rz0"][t], prob["rth0"][t])
        s.set_objective(obj_q, obj_u)
        s.set_window(np.stack([w for (w, _, _, _) in ro]) + 1)
        s.set_reference(np.stack([r.q for (_, r, _, _) in ro]), np.stack([r.u for (_, r, _, _) in ro]),
                        np.stack([r.w for (_, r, _, _) in ro]), np.stack([r.gamma for (_, r, _, _) in ro]),
                        np.stack([r.b for (_, r, _, _) in ro]), np.stack([r.theta for (_, r, _, _) in ro]))
        q0 = torch.tensor(np.stack([r[2] for r in ro]), dtype=torch.float64, device="cuda")
        q1 = torch.tensor(np.stack([r[3] for r in ro]), dtype=torch.float64, device="cuda")
        return s, q0, q1

    s, q0, q1 = make(B, rollouts)
    torch.cuda.synchronize()      # stream-ordering contract of cimpc_newton_solve_dev (include/cimpc.h): producers complete

    def step():
        s.newton_solve_dev(q0.data_ptr(), q1.data_ptr(), warm_start=False)

    def report():
        """Once per reporting interval: controls / iteration counts of every rollout of the job in global order
        (N > 1: one RCCL all-gather of B x (nu + 3) doubles per rank)."""
        u1, it, rn = s.newton_info()
        return mc.gather_results(u1, it, rn, s.rollout_counters()["sweeps"], n_global, dev)

    for _ in range(args.warmup):
        step()
    report()
    # HIP events in the timed region: around the launches of the roofline kernel (the interior-point sweep) ONLY - mode 2 of
    # cimpc_profile_enable.  Events around every launch of the step (three to four times as many records) cost the step ~4 %
    # (7.56 -> 7.88 ms); the per-class kernel times are taken from a short second pass after the timed region instead.
    s.profile_enable(0 if os.environ.get("CIMPC_BENCH_NOPROF") else 2)
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
    gathered = report()               # inside the timed region: the job's results reach every rank
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        cnt = torch.tensor([sweeps, ip_solves, ip_iters, newton_iters], dtype=torch.float64, device=dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)          # step counters of the whole job
        job_sweeps, job_ip_solves, job_ip_iters, job_newton = (float(x) for x in cnt.tolist())
    else:
        job_sweeps, job_ip_solves, job_ip_iters, job_newton = sweeps, ip_solves, ip_iters, newton_iters
    prof = s.profile_read()
    s.profile_enable(False)
    # per-class kernel times (informative: `kernel_time_ms_per_step`): a second, untimed pass with events around every launch
    prof_all, n_all = prof, args.steps
    if not os.environ.get("CIMPC_BENCH_NOPROF"):
        s.profile_enable(1); s.profile_reset()
        n_all = max(1, min(args.steps, 3))
        for _ in range(n_all):
            step()
        torch.cuda.synchronize()
        prof_all = s.profile_read()
        s.profile_enable(False)
        if dist is not None:
            dist.barrier()

    multi = None
    if dist is not None:
        # (1) self-check of the collective path: the rows of this rank's shard in the all-gathered result must be the rank's own
        #     local result, bit for bit (the gather only moves data); the verdict of all ranks is combined with a MIN all-reduce
        u1_loc, it_loc, rn_loc = s.newton_info()
        mine = gathered["u1"][first:first + B]
        ok_local = bool(np.array_equal(mine, u1_loc) and np.array_equal(gathered["newton_iters"][first:first + B], it_loc.astype(np.int64))
                        and np.array_equal(gathered["r_norm"][first:first + B], rn_loc))
        okt = torch.tensor([1.0 if ok_local else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        # (2) the OTHER scaling mode in the same run (the headline line keeps --scaling; the driver computes efficiencies itself):
        #     weak = --rollouts per GPU, strong = --rollouts in total split over the ranks (BASELINE configs[3]: 512 over 8)
        other = "strong" if args.scaling == "weak" else "weak"
        n2 = args.rollouts if other == "strong" else world * args.rollouts
        first2, B2 = rollout_shard(n2, rank, world)
        res2 = {"scaling": other, "rollouts_total": n2, "rollouts_per_gpu": B2}
        if B2 >= 1:
            ro2 = build_rollouts(d, prob, B2, H, H_ref, 1234, args.perturb, first=first2)
            s2, a2, b2 = make(B2, ro2)
            torch.cuda.synchronize()
            s2.newton_solve_dev(a2.data_ptr(), b2.data_ptr(), warm_start=False)
            k2 = max(1, min(args.steps, 5))
            torch.cuda.synchronize(); dist.barrier()
            t2 = time.perf_counter()
            for _ in range(k2):
                s2.newton_solve_dev(a2.data_ptr(), b2.data_ptr(), warm_start=False)
            u1b, itb, rnb = s2.newton_info()
            mc.gather_results(u1b, itb, rnb, s2.rollout_counters()["sweeps"], n2, dev)
            torch.cuda.synchronize(); dist.barrier()
            tt = torch.tensor([time.perf_counter() - t2], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            res2.update({"value": n2 * k2 / float(tt.item()), "unit": "MPC steps/s", "steps": k2, "ms_per_step": 1e3 * float(tt.item()) / k2})
            s2.close()
        multi = {"gather_selfcheck_all_ranks": bool(okt.item() == 1.0), "other_scaling": res2}

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    alg = algorithmic_sizes(**QUADRUPED)
    K = job_ip_iters / max(job_ip_solves, 1)
    L = job_newton / (n_global * args.steps)
    S = job_sweeps / max(job_newton + n_global * args.steps, 1)
    ip_ms = prof["ip_sweep_ms"]
    launches = max(prof["ip_sweep_launches"], 1)
    solves_per_launch = prof["ip_sweep_problems"] / launches
    avg_launch_ms = ip_ms / launches
    bytes_per_launch = solves_per_launch * alg["bytes_per_solve"]
    hbm_gbs = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if ip_ms > 0 else 0.0
    flops_per_solve = K * alg["flop_iter"] + alg["flop_tail"]
    tflops = solves_per_launch * flops_per_solve / (avg_launch_ms * 1e-3) / 1e12 if ip_ms > 0 else 0.0
    # roofline.traffic: measured in this run (N = 1), else the committed measurement IF it belongs to these kernels
    traffic, traffic_src = None, "skipped (--no-traffic)"
    if not args.no_traffic and world == 1:
        traffic, traffic_src = measure_sweep_traffic(B, H)
    if traffic is None:
        tpath = os.path.join(ROOT, "profiles", "ip_sweep_traffic.json")
        try:
            tj = json.load(open(tpath))
            if tj.get("rollouts") == B and tj.get("horizon") == H and tj.get("csrc_sha16") == csrc_hash():
                traffic, traffic_src = tj.get("hbm_bytes_per_launch"), traffic_src + "; profiles/ip_sweep_traffic.json (same kernel sources)"
            else:
                traffic_src += "; profiles/ip_sweep_traffic.json is STALE for these kernel sources / sizes - not used"
        except Exception:
            pass
    out = {
        "metric": "MPC steps/s (quadruped, H=40, fp64)",
        "value": n_global * args.steps / dt,
        "unit": "MPC steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "quadruped flat_2D_lc dims, :configuration, H=%d, H_ref=%d, fp64, %d Monte-Carlo rollouts in the job "
                               "(%d per GPU), batch-sharded over %d GPU(s) (BASELINE configs[3]; configs[0..2] = the B=1 legs under latency_b1 / mpc_loop_b1)"
                               % (H, H_ref, n_global, B, world),
                   "rollouts_total": n_global, "rollouts_per_gpu": B, "horizon": H, "perturb": args.perturb,
                   # SURVEY 8(d) cfg 4 draws q_ref + U(-0.1, 0.1).  On THIS synthetic generator's table that spread leaves the
                   # basin of the linearization: 90 % of the rollouts end at max_iter and ~134 interior-point solves per step fail, i.e. the line
                   # measures failure handling.  The headline keeps U(-0.05, 0.05) (rounds 1-5, comparable across rounds; a third of its rollouts
                   # still never reach r_tol); the contract's spread is reported beside it (headline["synthetic_perturb_0.1 ..."]) and the
                   # reference's real gait, which needs no such choice, under headline["real_gait2"].
                   "perturb_note": "U(-0.05,0.05) kept for the headline; SURVEY 8(d)'s U(-0.1,0.1) is the second headline entry (outside this generator's basin: ~10 % converge)",
                   "newton": "r_tol 3e-4, max_iter 5, cold start", "ip": "r_tol 1e-8, kappa_tol 2e-4, undercut 5",
                   "multi_gpu": "rank-0 problem broadcast + one all-gather of [u1|iters|r_norm|sweeps] per reporting interval over RCCL; no collective inside a solve" if world > 1 else "single GPU"},
        "solver_iters": {"newton_iters_per_step": L, "ip_iters_per_solve": K, "sweeps_per_eval": S,
                         "sweeps_per_step": job_sweeps / (n_global * args.steps), "lockstep_rounds_per_step": rounds / args.steps,
                         "converged_rollouts": int((gathered["r_norm"] < 3e-4).sum()), "rollouts_reported": int(gathered["u1"].shape[0])},
        # the sweep kernel runs out of LDS-resident tables: its binding roof is fp64 issue (vector FMA and MFMA share the
        # 78.6 TFLOP/s fp64 peak on gfx950), not HBM - both fractions are reported, the binding one as `frac`
        # `bound` names the pipe that binds: this kernel issues NO matrix instructions (SQ_INSTS_VALU_MFMA_MOPS_F64 = 0), its
        # roof is fp64 vector issue - the same 78.6 TFLOP/s as the fp64 MFMA dense peak on gfx950
        "roofline": {"bound": "fp64_valu", "bound_detail": "fp64 vector-issue roof (= the fp64 MFMA dense peak, 78.6 TFLOP/s, on gfx950; the kernel issues no MFMA); HBM contract fraction under `hbm`",
                     "kernel": "ip_queue_kernel", "achieved": tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": tflops / FP64_VECTOR_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                     "flops_per_unit": flops_per_solve, "bytes_per_unit": alg["bytes_per_solve"], "units_per_launch": solves_per_launch,
                     "avg_launch_ms": avg_launch_ms,
                     # `frac` counts the reference algorithm's flops per solve.  The device's sensitivity pass executes fewer
                     # (adjoint form): the same fraction on the flops actually issued
                     "flops_executed_per_unit": K * alg["flop_iter"] + alg["flop_tail_executed"],
                     "frac_executed": tflops / FP64_VECTOR_PEAK_TFLOPS * (K * alg["flop_iter"] + alg["flop_tail_executed"]) / flops_per_solve,
                     # the whole step against the same roof: every contract flop of the step (all evaluated sweeps + one
                     # condensed KKT solve per Newton iteration, SURVEY 8d: 9.9 MFLOP banded-equivalent) over the step's wall time
                     "step_frac": ((job_ip_solves * alg["flop_iter"] * K + job_ip_solves * alg["flop_tail"] + job_newton * KKT_FLOP_PER_SOLVE)
                                   / dt / 1e12 / world) / FP64_VECTOR_PEAK_TFLOPS,
                     "hbm": {"achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
                             "achieved_shared_table": solves_per_launch * alg["bytes_per_solve_shared_table"] / (avg_launch_ms * 1e-3) / 1e9 if ip_ms > 0 else 0.0,
                             "traffic_over_algorithmic": (traffic / bytes_per_launch) if traffic and bytes_per_launch else None}},
        "kernel_time_ms_per_step": {"ip_sweep": prof["ip_sweep_ms"] / args.steps, "kkt": prof_all["kkt_ms"] / n_all,
                                    "resid": prof_all["resid_ms"] / n_all, "other": prof_all["other_ms"] / n_all,
                                    "async_tail": prof_all["async_ms"] / n_all,
                                    "note": "ip_sweep: HIP events of the timed region; the other classes: an untimed second pass of %d step(s) with events around every launch" % n_all},
        "schedule": {"lockstep_rounds_per_step": rounds / args.steps - (1 if prof_all["async_launches"] else 0),
                     "async_tail_launches_per_step": prof_all["async_launches"] / n_all,
                     "ip_problems_in_rounds": prof["ip_sweep_problems"] / args.steps,
                     "ip_problems_in_async_tail": prof["async_problems"] / args.steps},
        # the two Monte-Carlo workloads side by side: the seeded synthetic batch `value` is quoted on (backtracking-heavy:
        # a third of its rollouts never reach r_tol) and, at N = 1, the same batch size on the reference's real gait
        "headline": {"synthetic": {"value": n_global * args.steps / dt, "unit": "MPC steps/s", "ms_per_step": 1e3 * dt / args.steps,
                                   "converged_rollouts": int((gathered["r_norm"] < 3e-4).sum()), "rollouts": n_global}},
        "setup_s": t_setup,
        "csrc_sha16": csrc_hash(),      # identity of the kernel sources this line was measured on
    }
    # roofline.kernels: the other kernels of the path next to the dominant one - same construction: contract flops
    # of the units a launch processes / average launch duration by HIP events; the entries of the B = 1 and centroidal legs follow below
    out["roofline"]["kernels"] = [
        kernel_roofline("ip_queue_kernel<quadruped> (B = %d)" % B, flops_per_solve, prof["ip_sweep_problems"], prof["ip_sweep_launches"], prof["ip_sweep_ms"],
                        "the dominant kernel = roofline.frac", flops_executed=K * alg["flop_iter"] + alg["flop_tail_executed"]),
        kernel_roofline("kkt_kernel_duo<11,8> / kkt_kernel_packed<11,8> (B = %d, next to the sweep)" % B, kkt_condensed_flops(d.nq, H), prof_all["kkt_systems"], prof_all["kkt_launches"],
                        prof_all["kkt_ms"], "condensed f64-MFMA solve, SURVEY 8 A13: 1.2 MFLOP per system; the HIP-event class covers both kernels of the rounds' KKT stage: "
                        "the duo kernel (two one-wave chains per system in one workgroup, rounds whose sweep has <= 20 k problems queued) and the packed kernel "
                        "(one wavefront per system, two per workgroup) - latency chains of H / 2 resp. H block steps; per-kernel averages: profiles/r06/kernel_stats.csv"),
    ]
    if multi is not None:
        out["multi_gpu"] = multi
    if not args.no_latency and world == 1:
        # BASELINE configs[2]: the same quadruped problem for ONE robot, cold start
        s1, a0, a1 = make(1, rollouts[:1])
        for _ in range(3):
            s1.newton_solve_dev(a0.data_ptr(), a1.data_ptr(), False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n1 = 20
        for _ in range(n1):
            s1.newton_solve_dev(a0.data_ptr(), a1.data_ptr(), False)
        torch.cuda.synchronize()
        ms1 = 1e3 * (time.perf_counter() - t1) / n1
        out["latency_b1"] = {"workload": "quadruped H=40, ONE rollout, cold-start newton_solve! (BASELINE configs[2])",
                             "ms_per_step": ms1, "mpc_steps_per_s": 1e3 / ms1, "stats": s1.stats()}
        s1.profile_enable(1); s1.profile_reset()
        for _ in range(5):
            s1.newton_solve_dev(a0.data_ptr(), a1.data_ptr(), False)
        torch.cuda.synchronize()
        p1 = s1.profile_read()
        out["latency_b1"]["kernel_time_ms_per_step"] = {"ip_sweep": p1["ip_sweep_ms"] / 5, "kkt": p1["kkt_ms"] / 5, "resid": p1["resid_ms"] / 5, "other": p1["other_ms"] / 5,
                                                        "kkt_launches": p1["kkt_launches"] / 5, "kkt_twisted_launches_since_create": s1.kkt_twisted()}
        out["roofline"]["kernels"].append(kernel_roofline("kkt_kernel_twisted<11,8> (B = 1)", kkt_condensed_flops(d.nq, H), p1["kkt_systems"], p1["kkt_launches"], p1["kkt_ms"],
                                                          "two workgroups of three wavefronts per system, one chain from either end (round 5); latency-bound by construction"))
        s1.close()
        # the reference's own use: ONE robot, warm-started MPC steps in a loop (policy.jl:98-146 cadence without the
        # plant: q1_next = planned q_3), reference / window advanced on the device (cimpc_mpc_advance)
        out["mpc_loop_b1"] = {"quadruped_h40 (BASELINE configs[2])": mpc_loop_latency(QUADRUPED, "quadruped", 40, 60, local_rank),
                              "hopper_h20 (BASELINE configs[1])": mpc_loop_latency(dict(nq=4, nu=2, nw=2, nc=1, nb=2), "hopper", 20, 30, local_rank),
                              "pushbot_h10_configurationforce (BASELINE configs[0])": mpc_loop_latency(dict(nq=2, nu=2, nw=2, nc=2, nb=4), "pushbot", 10, 16, local_rank, mode=1),
                              "quadruped_h40_real_gait2": real_mpc_loop_latency(40, local_rank),
                              # BASELINE configs[1] on the REAL hopper problem (examples/hopper/flat.jl:15-47 with H_mpc = 20 as the config names it; the
                              # example itself runs H_mpc = 10 - second entry).  The gait file is a `:joint_traj` file that predates the shipped hopper
                              # model (leg-length row off by 4e-3, tests/test_real_models.py): it is the reference's PROBLEM, linearized at the file's own z / θ.
                              "hopper_h20_real_gait (BASELINE configs[1], real problem)": real_mpc_loop_latency(20, local_rank, which="hopper"),
                              "hopper_h10_real_gait (examples/hopper/flat.jl H_mpc)": real_mpc_loop_latency(10, local_rank, which="hopper")}
        # the same real-gait loop as a user of the B4 drop-in drives it: window + rotated reference from the HOST at every step
        try:
            out["dropin_b4_host_path"] = dict(dropin_host_path_latency(40, local_rank),
                                              workload="quadruped gait2, H=40, ONE robot, warm-started loop; C-call sequence of julia/CIMPCHip.jl's drop-in through ctypes with host arrays "
                                                       "(beside mpc_loop_b1.quadruped_h40_real_gait2 = the device-resident glue cimpc_set_gait + cimpc_mpc_advance)")
        except Exception as e:
            out["dropin_b4_host_path"] = {"error": repr(e)}
    if world == 1 and abs(args.perturb - 0.1) > 1e-12 and not args.no_real_problem:
        # SURVEY.md 8(d) cfg 4 draws the initial conditions as q_ref + U(-0.1, 0.1); the headline batch uses 0.05 (rounds 1-4, kept for
        # continuity).  The contract's own spread stands beside it (outside the timed region, same kernels, same counting):
        try:
            ro2 = build_rollouts(d, prob, B, H, H_ref, 1234, 0.1, first=first)
            s2, b0, b1 = make(B, ro2)
            torch.cuda.synchronize()
            s2.newton_solve_dev(b0.data_ptr(), b1.data_ptr(), False)
            torch.cuda.synchronize()
            k2 = 5
            t2 = time.perf_counter()
            for _ in range(k2):
                s2.newton_solve_dev(b0.data_ptr(), b1.data_ptr(), False)
            _, it2, rn2 = s2.newton_info()
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t2) / k2
            st2 = s2.stats()
            out["headline"]["synthetic_perturb_0.1 (SURVEY 8d cfg 4)"] = {"value": B / dt2, "unit": "MPC steps/s", "ms_per_step": 1e3 * dt2, "rollouts": B,
                                                                         "converged_rollouts": int((rn2 < 3e-4).sum()), "newton_iters_per_step": float(it2.mean()),
                                                                         "sweeps_per_step": st2["sweeps"] / B, "ip_failures": st2["ip_failures"]}
            s2.close()
        except Exception as e:
            out["headline"]["synthetic_perturb_0.1 (SURVEY 8d cfg 4)"] = {"error": repr(e)}
    if not args.no_real_problem and world == 1:      # informative second workload (N = 1 only), outside the timed region
        try:
            out["real_problem"] = real_problem_leg(B, H, local_rank)
            # the better-conditioned workload stands beside the synthetic one in the headline block
            out["headline"]["real_gait2"] = {k: out["real_problem"][k] for k in ("value", "unit", "ms_per_step", "converged_rollouts", "rollouts")}
        except Exception as e:
            out["real_problem"] = {"error": repr(e)}
    if not args.no_centroidal and world == 1:        # BASELINE configs[4]: per-GPU share of 512 rollouts, mixed-precision KKT next to fp64
        try:
            out["centroidal_payload_h60"] = cp = centroidal_payload_leg(64, 60, local_rank)
            c64, v64 = cp["fp64_kkt"], cp["velocity_objective"]
            cd = dict(nq=18, nu=12, nw=3, nc=4, nb=16)
            ca = algorithmic_sizes(**cd)
            out["roofline"]["kernels"] += [
                kernel_roofline("ip_queue_kernel<centroidal> (B = 64, H = 60)", c64["ip_iters_per_solve"] * ca["flop_iter"] + ca["flop_tail"],
                                c64["ip_sweep_problems_per_step"], c64["ip_sweep_launches_per_step"], c64["ip_sweep_ms_per_step"], "32-lane groups, one 512-register wave per SIMD",
                                flops_executed=c64["ip_iters_per_solve"] * ca["flop_iter"] + ca["flop_tail_executed"]),
                kernel_roofline("kkt_kernel_twisted<18,12> (B = 64, H = 60)", kkt_condensed_flops(18, 60), c64["kkt_systems_per_step"], c64["kkt_launches_per_step"],
                                c64["kkt_ms_per_step"], "24 x 24 tiles = 2 x 2 masked MFMA blocks"),
                kernel_roofline("kkt_banded_twisted_kernel<8,128> (B = 64, H = 60, velocity objective)", kkt_banded_flops(18, 12, 60), v64.get("kkt_systems_per_step", 0),
                                v64.get("kkt_launches_per_step", 0), v64.get("kkt_ms_per_step", 0.0), "SURVEY 8(d): N w^2 + 4 N w, N = 2160, w = 107; two workgroups (16 waves each) per system, one chain from either end of the band",
                                bound="fp64_valu+mfma (chain wavefront: DPP triangular solves; bulk wavefronts: 16 x 16 MFMA tiles)"),
            ]
        except Exception as e:
            out["centroidal_payload_h60"] = {"error": repr(e)}
        try:      # the config as the example runs it: warm-started, closed loop, to tolerance (VERDICT r04 missing 3)
            out["centroidal_closed_loop_h60"] = centroidal_closed_loop_leg(64, 60, local_rank)
        except Exception as e:
            out["centroidal_closed_loop_h60"] = {"error": repr(e)}
        try:
            out["centroidal_payload_h60_b1"] = centroidal_b1_leg(60, local_rank)
        except Exception as e:
            out["centroidal_payload_h60_b1"] = {"error": repr(e)}
    if not args.no_cpu_baseline and world == 1:      # the CPU baseline is a rank-0, N = 1 measurement
        try:
            from contactimplicitmpc.jl_amd.trajectory import Objective
            cb = cpu_baseline(d, prob, Objective(q=obj_q, u=obj_u), rollouts, H, H_ref)
            out["cpu_baseline"] = {
                "value": cb["condensed"][0], "unit": "MPC steps/s", "cores": 1, "kind": "port",
                "sample": "%d rollouts (condensed block-Cholesky KKT, the :ldl_solver analogue) and %d rollouts "
                          "(dense LU, reference default :lu_solver) of the same batch, oracle/cimpc_ref.c, 1 thread"
                          % (cb["condensed"][1], cb["dense_lu"][1]),
                "value_dense_lu": cb["dense_lu"][0],
                "value_all_cores": cb["all_cores"][0], "all_cores_threads": cb["all_cores"][2],
                "all_cores_sample": "%d rollouts, condensed KKT, one solver instance per thread" % cb["all_cores"][1],
                "host_cores_available": os.cpu_count(),
                "note": "C restatement of the reference algorithm (not the Julia package: no Julia in this image)"}
            out["speedup_vs_cpu_1thread"] = out["value"] / cb["condensed"][0]
            if "latency_b1" in out:     # the north-star ratio on the reference's own configuration: ONE robot, H = 40 - the SAME rollout on both sides
                t_cpu = cpu_single_rollout(d, prob, Objective(q=obj_q, u=obj_u), rollouts[0], H, H_ref)
                out["latency_b1"]["cpu_1thread_same_rollout_ms"] = 1e3 * t_cpu
                out["latency_b1"]["speedup_vs_cpu_1thread"] = t_cpu * out["latency_b1"]["mpc_steps_per_s"]
                out["latency_b1"]["speedup_vs_cpu_1thread_batch_average"] = out["latency_b1"]["mpc_steps_per_s"] / cb["condensed"][0]
        except Exception as e:  # the baseline never blocks the GPU number
            out["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
