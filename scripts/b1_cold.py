"""B = 1 cold-start newton_solve! of the quadruped H = 40 problem (BASELINE configs[2]), n times - the workload of the per-kernel
rocprofv3 statistics `profiles/r05/kernel_stats_b1*.csv` (scripts/gpu.sh: b1stats).  usage: python scripts/b1_cold.py [n] [model]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import scripts.ab_env as ab  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
t0 = time.perf_counter()
print(ab.cold_b1(n=n), "wall %.2f s" % (time.perf_counter() - t0))
