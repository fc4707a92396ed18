cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02d}
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/tests_$TAG.log 2>&1; echo "tests rc $?"
tail -8 gpurun_out/tests_$TAG.log
bash scripts/sweep_knobs.sh CIMPC_SPEC_ALL=3 CIMPC_SPEC_ALL=5 CIMPC_KKT_PIPE=1 CIMPC_WAVES=2 CIMPC_SWEEP_WGS=256 CIMPC_SWEEP_WGS=384 CIMPC_ITER_CAP=28 CIMPC_ASYNC_TAIL=96 CIMPC_KKT_OVERLAP=0 2>&1 | tee gpurun_out/sweep_$TAG.log
CIMPC_DEBUG_ROUNDS=1 timeout 100 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-real-problem --no-latency --no-traffic 2>&1 | grep "cimpc round" | tail -25 > gpurun_out/rounds_$TAG.log; cat gpurun_out/rounds_$TAG.log
