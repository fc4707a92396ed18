#!/bin/bash
# usage (GPU box): scripts/sweep_knobs.sh  - headline bench under a few schedule knobs (each line: knob, MPC steps/s, ms, solver stats)
run() { env "$@" timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-real-problem 2>/dev/null | tail -1 | python -c "import json,sys; o=json.loads(sys.stdin.read()); print('$*', round(o['value']), round(o['ms_per_step'],2), o['solver_iters']['newton_iters_per_step'], o['solver_iters']['sweeps_per_step'], o['solver_iters']['ip_iters_per_solve'])"; }
run X=0
run CIMPC_ITER_CAP=20
run CIMPC_ITER_CAP=24
run CIMPC_ITER_CAP=32
run CIMPC_ITER_CAP=48
run CIMPC_ITER_CAP=100
run CIMPC_ITER_CAP=24 CIMPC_ASYNC_TAIL=96
run CIMPC_ITER_CAP=32 CIMPC_ASYNC_TAIL=96
run CIMPC_ITER_CAP=32 CIMPC_ROLLOUTS=2048
