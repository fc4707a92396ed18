#!/bin/bash
# A/B of the B = 1 latency legs of two builds on ONE box: scripts/ab_lat.sh <libA> <libB> [rounds]
A=$1; B=$2; N=${3:-2}
for i in $(seq $N); do
  for L in $A $B; do
    CIMPC_LIB=$PWD/$L python bench.py --steps 3 --warmup 1 --rollouts 64 --no-cpu-baseline --no-real-problem --no-traffic --no-centroidal 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
m=d['mpc_loop_b1']
print('$L B=64 ms/step %.3f | B=1 cold %.4f ms |'%(d['ms_per_step'], d['latency_b1']['ms_per_step']), ' '.join('%s %.3f'%(k.split(' ')[0], v['ms_per_mpc_step']) for k,v in m.items()))
"
  done
done
