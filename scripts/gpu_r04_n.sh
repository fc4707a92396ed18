cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2 3; do bash scripts/knob.sh "CIMPC_ITER_CAP=28" "CIMPC_ITER_CAP=32" "CIMPC_ITER_CAP=36" "CIMPC_ITER_CAP=40" "CIMPC_ITER_CAP=48" "CIMPC_ITER_CAP=100"; done > gpurun_out/knob_iter_cap_r04.log 2>&1
cat gpurun_out/knob_iter_cap_r04.log
BENCH_ARGS="--rollouts 2048" bash scripts/knob.sh "CIMPC_ITER_CAP=28" "CIMPC_ITER_CAP=36" "CIMPC_ITER_CAP=48" >> gpurun_out/knob_iter_cap_r04.log 2>&1
BENCH_ARGS="--rollouts 256" bash scripts/knob.sh "CIMPC_ITER_CAP=28" "CIMPC_ITER_CAP=36" "CIMPC_ITER_CAP=48" >> gpurun_out/knob_iter_cap_r04.log 2>&1
tail -6 gpurun_out/knob_iter_cap_r04.log
