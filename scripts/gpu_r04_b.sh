# round 4, second GPU call: the new tests, the small-round grid knob, the round timeline, the rocprofv3 passes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_bench_launcher.py -m gpu -q > gpurun_out/tests_r04b.log 2>&1; echo "tests rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests_r04b.log | tail -12
grep -E "^E  " gpurun_out/tests_r04b.log | head -30
for i in 1 2; do bash scripts/knob.sh "" "CIMPC_SMALL_ROUND=4096" "CIMPC_SMALL_ROUND=8192" "CIMPC_SMALL_ROUND=16384"; done > gpurun_out/knob_small_round.log 2>&1
BENCH_ARGS="--rollouts 128" bash scripts/knob.sh "" "CIMPC_SMALL_ROUND=4096" "CIMPC_SMALL_ROUND=8192" >> gpurun_out/knob_small_round.log 2>&1
BENCH_ARGS="--rollouts 256" bash scripts/knob.sh "" "CIMPC_SMALL_ROUND=4096" "CIMPC_SMALL_ROUND=8192" >> gpurun_out/knob_small_round.log 2>&1
cat gpurun_out/knob_small_round.log
CIMPC_DEBUG_ROUNDS=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-real-problem --no-latency --no-traffic --no-centroidal > /dev/null 2> gpurun_out/round_timeline_r04.log
tail -45 gpurun_out/round_timeline_r04.log
bash scripts/profile_round.sh r04 > gpurun_out/profile_r04.log 2>&1; tail -5 gpurun_out/profile_r04.log
