"""BASELINE configs[4] legs alone (64 rollouts, H = 60: tracking objective = twisted condensed KKT, velocity objective = twisted banded LDL^T) -
the workload of profiles/r05/pmc_centroidal/.  usage: python scripts/cent_legs.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

I = bench.centroidal_payload_inputs(64, 60)
print(json.dumps(bench.centroidal_velocity_leg(I, 64, 60, 0, steps=2)))
os.environ["CIMPC_ASYNC"] = "0"
c = bench.centroidal_payload_leg(64, 60, 0, steps=2)
print(json.dumps(c["fp64_kkt"]))
