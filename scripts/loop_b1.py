"""Warm-started B = 1 MPC loop of bench.py (hopper H = 20 / quadruped H = 40 / real gait2) - the workload of the kernel traces
`profiles/r05/loop_trace_*.txt` (scripts/gpu.sh: looptrace).  usage: python scripts/loop_b1.py hopper|quadruped|gait2 [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "hopper"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
if which == "hopper":
    print(bench.mpc_loop_latency(dict(nq=4, nu=2, nw=2, nc=1, nb=2), "hopper", 20, 30, 0, steps=steps))
elif which == "quadruped":
    print(bench.mpc_loop_latency(bench.QUADRUPED, "quadruped", 40, 60, 0, steps=steps))
else:
    print(bench.real_mpc_loop_latency(40, 0, steps=steps))
