for L in "" scripts/build/libcimpc_noadj.so scripts/build/libcimpc_noadj_pad.so; do
  for rep in 1 2; do
  CIMPC_LIB=$L timeout 300 python bench.py --no-cpu-baseline --no-latency --no-real-problem --no-traffic --no-centroidal > /tmp/b.json 2>/dev/null
  python -c "
import json;b=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print('lib=[$L]',round(b['value']),b['ms_per_step'],b['solver_iters']['lockstep_rounds_per_step'],b['roofline']['avg_launch_ms'],b['kernel_time_ms_per_step'])"
  done
done
