"""One-screen digest of a driver-format bench line: python scripts/bench_digest.py <file.json> [...]"""
import json
import sys

for f in sys.argv[1:]:
    b = json.loads(open(f).read().strip().splitlines()[-1])
    print("==", f, "csrc", b.get("csrc_sha16"))
    print("headline %d steps/s  %.3f ms/step" % (round(b["value"]), b["ms_per_step"]), b["solver_iters"])
    r = b["roofline"]
    print("  frac %.4f  executed %.4f  launch %.4f ms  units/launch %.0f  traffic %s (x %.3f of contract)  step_frac %.4f"
          % (r["frac"], r.get("frac_executed", float("nan")), r["avg_launch_ms"], r["units_per_launch"], r["traffic"], r["hbm"]["traffic_over_algorithmic"] or 0, r["step_frac"]))
    for k in r.get("kernels", []):
        print("    %-66s frac %.4f  %.4f ms" % (k["kernel"][:66], k["frac"], k["avg_launch_ms"]))
    print("  kernel ms/step", {k: round(v, 3) for k, v in b["kernel_time_ms_per_step"].items() if isinstance(v, float)}, b.get("schedule"))
    print("  legs", {k[:24]: (round(v["value"]), v.get("converged_rollouts")) for k, v in b.get("headline", {}).items()})
    c = b.get("centroidal_payload_h60", {})
    for kk, v in c.items():
        if isinstance(v, dict) and "ms_per_step" in v:
            print("    cfg4 %-26s %.3f ms  sweep %.3f  kkt %.3f  %.0f steps/s  iters %s" % (kk, v["ms_per_step"], v.get("ip_sweep_ms_per_step", 0), v.get("kkt_ms_per_step", 0), v.get("value", 0), v.get("newton_iters_per_step")))
    if "centroidal_closed_loop_h60" in b:
        cl = b["centroidal_closed_loop_h60"]
        print("    closed loop %.3f ms  %.0f steps/s  iters %.3f" % (cl["ms_per_mpc_step_warm"], cl["value_warm"], cl["newton_iters_per_solve_warm"]), {k: v for k, v in cl.items() if "conv" in k or "below" in k})
    if "centroidal_payload_h60_b1" in b:
        print("    cfg4 B=1", {k: (round(v["ms_per_step"], 3), round(v["ip_sweep_ms_per_step"], 3), round(v["kkt_ms_per_step"], 3)) for k, v in b["centroidal_payload_h60_b1"].items() if isinstance(v, dict)})
    if "latency_b1" in b:
        l = b["latency_b1"]
        print("  B=1 cold %.4f ms" % l["ms_per_step"], {k: round(v, 4) for k, v in l["kernel_time_ms_per_step"].items() if isinstance(v, float)}, "vs cpu same rollout", l.get("cpu_1thread_same_rollout_ms"), l.get("speedup_vs_cpu_1thread"))
    if "mpc_loop_b1" in b:
        print("  loops", {k[:22]: (round(v["ms_per_mpc_step"], 3), v["newton_iters_per_step"]) for k, v in b["mpc_loop_b1"].items() if isinstance(v, dict)})
    cb = b.get("cpu_baseline")
    if cb:
        print("  cpu", cb["value"], cb.get("value_dense_lu"), cb.get("value_all_cores"), "speedup", b.get("speedup_vs_cpu_1thread"))
