"""BASELINE configs[4] with the example's own velocity objective: the bench leg alone (python scripts/dbg/vel_leg.py [B])."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
I = bench.centroidal_payload_inputs(B, 60)
o = bench.centroidal_velocity_leg(I, B, 60, 0)
print(os.environ.get("TAG", ""), "B", B, json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in o.items() if k not in ("objective", "kkt", "unit")}))
