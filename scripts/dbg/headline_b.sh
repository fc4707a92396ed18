# headline-only bench lines at other batch sizes: bash scripts/dbg/headline_b.sh "<ENV=...>" B1 B2 ...
cfg="$1"; shift
for B in "$@"; do
  for rep in 1 2; do
  env $cfg timeout 300 python bench.py --rollouts $B --no-cpu-baseline --no-latency --no-real-problem --no-traffic --no-centroidal > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
  python -c "
import json;b=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);k=b['kernel_time_ms_per_step'];print('[$cfg] B=$B',round(b['value']),'ms %.3f'%b['ms_per_step'],'rounds',b['solver_iters']['lockstep_rounds_per_step'],'sweep %.2f kkt %.2f resid %.2f tail %.2f'%(k['ip_sweep'],k['kkt'],k['resid'],k['async_tail']), 'conv',b['solver_iters']['converged_rollouts'])"
  done
done
