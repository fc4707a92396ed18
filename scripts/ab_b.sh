#!/bin/bash
# A/B/C of builds at one batch size: scripts/ab_b.sh <B> <rounds> lib1 lib2 ...
Bn=$1; N=$2; shift 2
for i in $(seq $N); do
  for L in "$@"; do
    CIMPC_LIB=$PWD/$L python bench.py --steps 8 --warmup 2 --rollouts $Bn --no-cpu-baseline --no-real-problem --no-latency --no-traffic --no-centroidal 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L B=$Bn ms/step %.3f sweeps %.3f'%(d['ms_per_step'], d['solver_iters']['sweeps_per_step']), {a:round(b,3) for a,b in d['kernel_time_ms_per_step'].items() if not isinstance(b,str)})
"
  done
done
