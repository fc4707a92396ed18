#!/bin/bash
# usage (GPU box): scripts/sweep_knobs.sh KNOB=VALUE [KNOB=VALUE ...]  - one headline bench run per argument next to two
# default runs (run-to-run noise on one box is about +-1.5 %).  Prints: setting, MPC steps/s, ms per batch step, Newton
# iterations and evaluated sweeps per step (the last two must not change with a pure scheduling knob).
run() { env $@ timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-real-problem --no-latency --no-traffic --no-centroidal 2>/dev/null | tail -1 | python -c "import json,sys; o=json.loads(sys.stdin.read()); print('$*', round(o['value']), round(o['ms_per_step'],2), o['solver_iters']['newton_iters_per_step'], o['solver_iters']['sweeps_per_step'], 'sweep_launch_ms', round(o['roofline']['avg_launch_ms'],4), 'kernels', {k: round(v,2) for k,v in o['kernel_time_ms_per_step'].items() if not isinstance(v,str)})"; }
run DEFAULT=1
for kv in "$@"; do run "$kv"; done
run DEFAULT=1
