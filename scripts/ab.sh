#!/bin/bash
# A/B of two builds of the library on ONE box, alternating runs: scripts/ab.sh <libA> <libB> [rounds] [extra bench args]
A=$1; B=$2; N=${3:-3}; shift 3
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-real-problem --no-latency --no-traffic --no-centroidal $@"
for i in $(seq $N); do
  for L in $A $B; do
    CIMPC_LIB=$PWD/$L python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_time_ms_per_step']; r=d['roofline']
print('$L ms/step %.3f value %.0f frac %.4f'%(d['ms_per_step'],d['value'],r['frac']), {a:round(b,3) for a,b in k.items() if not isinstance(b,str)}, 'rounds', d['solver_iters']['lockstep_rounds_per_step'], 'sweeps', round(d['solver_iters']['sweeps_per_step'],3), 'avg_launch_ms %.4f'%r['avg_launch_ms'])
"
  done
done
