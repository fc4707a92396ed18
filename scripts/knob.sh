#!/bin/bash
# knob sweep on ONE box: scripts/knob.sh "<ENV=V ...>" ... (each argument = one environment setting, "" = defaults)
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --no-real-problem --no-latency --no-traffic --no-centroidal ${BENCH_ARGS}"
for E in "$@"; do
  env $E python bench.py $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[%s] ms/step %.3f'%('$E', d['ms_per_step']), {a:round(b,3) for a,b in d['kernel_time_ms_per_step'].items() if not isinstance(b,str)}, 'rounds', d['solver_iters']['lockstep_rounds_per_step'], 'sweeps %.3f'%d['solver_iters']['sweeps_per_step'])
"
done
