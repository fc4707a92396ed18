# round 4, third GPU call: round-4 tests, statistics sanity at hybrid batch sizes (both libraries), per-launch problems x durations
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -q > gpurun_out/tests_r04c.log 2>&1; echo "tests rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests_r04c.log | tail -12
grep -E "^E  " gpurun_out/tests_r04c.log | head -30
for L in contactimplicitmpc/jl_amd/libcimpc_r03.so contactimplicitmpc/jl_amd/libcimpc_hip.so; do for B in 96 128 192; do CIMPC_LIB=$PWD/$L python scripts/dbg/stats_check.py $B 16 2>&1 | tail -6; done; done > gpurun_out/stats_check.log 2>&1
STATS_PROF=1 python scripts/dbg/stats_check.py 128 16 >> gpurun_out/stats_check.log 2>&1
cat gpurun_out/stats_check.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CIMPC_DEBUG_ROUNDS=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r04/launches -o l -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-real-problem --no-latency --no-traffic --no-centroidal > /dev/null 2> gpurun_out/launch_problems_r04.log
grep -c "sweep launch" gpurun_out/launch_problems_r04.log
