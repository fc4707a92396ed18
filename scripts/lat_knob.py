"""B = 1 latency legs of bench.py under the current environment (knob A/B on one box): python scripts/lat_knob.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

out = {"real_gait2": bench.real_mpc_loop_latency(40, 0)["ms_per_mpc_step"],
       "quadruped_synth": bench.mpc_loop_latency(bench.QUADRUPED, "quadruped", 40, 60, 0)["ms_per_mpc_step"],
       "hopper": bench.mpc_loop_latency(dict(nq=4, nu=2, nw=2, nc=1, nb=2), "hopper", 20, 30, 0)["ms_per_mpc_step"]}
print(os.environ.get("TAG", ""), json.dumps({k: round(v, 3) for k, v in out.items()}))
