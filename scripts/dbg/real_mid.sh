# synthetic and real gait2 legs at 64 / 128 rollouts under the default schedule boundaries and under round 5's (single launch up to 64, hand-over at 64)
for B in 64 128; do for e in CIMPC_X=1 "CIMPC_ASYNC_FULL_MAX=64 CIMPC_ASYNC_TAIL=64"; do for r in 1 2; do
env $e timeout 300 python bench.py --rollouts $B --no-cpu-baseline --no-latency --no-traffic --no-centroidal 2>/dev/null | tail -1 | python -c "
import json,sys; o=json.loads(sys.stdin.read()); rp=o.get('real_problem',{}); print('B=$B [$e] synthetic %.3f ms  real gait2 %.3f ms (%s conv)'%(o['ms_per_step'], rp.get('ms_per_step',0), rp.get('converged_rollouts')))"
done; done; done
