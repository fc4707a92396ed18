"""A/B of the twisted KKT kernel in the overlapped rounds of mid-size and large batches: CIMPC_KKT_TWISTED = bound on the rollouts per KKT launch.
usage: python scripts/ab_pipe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scripts.ab_env as ab  # noqa: E402

for rep in range(2):
    for B in (96, 128, 256, 512):
        for v in ("1", "24", "48", "96"):
            os.environ["CIMPC_KKT_TWISTED"] = v
            r = ab.cold_b1(B=B, n=8)
            print("B", B, "KKT_TWISTED", v, json.dumps({k: r[k] for k in ("ms", "kkt_ms", "sweep_ms", "resid_ms", "twisted_launches")}), flush=True)
