# N headline-only bench lines per environment setting, alternating: bash scripts/dbg/headline_reps.sh N "<ENV=...>" "<ENV=...>" ...
N=$1; shift
for rep in $(seq $N); do
for cfg in "$@"; do
  env $cfg timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-latency --no-real-problem --no-traffic --no-centroidal > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
  python -c "
import json;b=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);k=b['kernel_time_ms_per_step'];print('[$cfg]',round(b['value']),'ms %.3f'%b['ms_per_step'],'sweep %.2f kkt %.2f resid %.2f tail %.2f'%(k['ip_sweep'],k['kkt'],k['resid'],k['async_tail']), 'conv',b['solver_iters']['converged_rollouts'])"
done
done
