#!/bin/bash
# usage (GPU box): scripts/batch_sizes.sh  - the headline workload at other batch sizes (rollouts per GPU), one line each
cd $GRAFT_REPO_ROOT
for B in 1 8 64 128 256 512 1024 2048 4096; do
timeout 200 python bench.py --rollouts $B --steps 6 --warmup 2 --no-cpu-baseline --no-real-problem --no-latency --no-traffic --no-centroidal 2>/dev/null | tail -1 | python -c "
import json,sys; o=json.loads(sys.stdin.read()); print('rollouts', $B, 'MPC steps/s', round(o['value']), 'ms per batch step', round(o['ms_per_step'],3), 'newton iters/step', round(o['solver_iters']['newton_iters_per_step'],3), 'roofline frac', round(o['roofline']['frac'],3), 'rounds', o['schedule']['lockstep_rounds_per_step'])"
done
