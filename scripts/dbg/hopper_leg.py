import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
print(json.dumps({"h20": bench.real_mpc_loop_latency(20, 0, which="hopper"), "h10": bench.real_mpc_loop_latency(10, 0, which="hopper")}, indent=1))
