"""bench.py's drop-in host-path leg alone (beside the device-resident real-gait loop): python scripts/dbg/dropin_leg.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
print(json.dumps({"device_resident_glue": bench.real_mpc_loop_latency(40, 0), "dropin": bench.dropin_host_path_latency(40, 0)}, indent=1))
