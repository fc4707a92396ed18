#!/bin/bash
# per-kernel time summary of a short bench run: scripts/kstats.sh <out-name> [extra bench args]
N=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$N
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$N -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-real-problem --no-latency --no-traffic --no-centroidal "$@" > /tmp/ks_$N.json 2>/tmp/ks_$N.err
F=$(find /tmp/ks_$N -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cp $F $GRAFT_REPO_ROOT/gpurun_out/kstats_$N.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$F")))
for r in rows[:12]:
    print("%-60s calls %6s  avg %9.1f us  total %8.2f ms  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
