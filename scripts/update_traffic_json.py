#!/usr/bin/env python3
"""profiles/ip_sweep_traffic.json from a pmc_summary.csv (scripts/pmc_summary.py): HBM bytes per ip_queue_kernel launch
= (2 * FETCH_SIZE + WRITE_SIZE) * 1024, stamped with the hash of the kernel sources it was measured on (bench.py refuses
the file when the sources have changed).  usage: update_traffic_json.py <pmc_summary.csv> <rollouts> <horizon>"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

vals = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "ip_queue_kernel" in r["kernel"] and r["counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
        vals[r["counter"]] = float(r["avg_per_dispatch"])
out = {"kernel": "ip_queue_kernel", "rollouts": int(sys.argv[2]), "horizon": int(sys.argv[3]),
       "fetch_size_kb_per_launch": vals["FETCH_SIZE"], "write_size_kb_per_launch": vals["WRITE_SIZE"],
       "hbm_bytes_per_launch": (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
       "csrc_sha16": bench.csrc_hash(), "source": sys.argv[1]}
json.dump(out, open(os.path.join(ROOT, "profiles", "ip_sweep_traffic.json"), "w"), indent=1)
print(out)
