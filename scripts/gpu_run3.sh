cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02c}
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/tests_$TAG.log 2>&1; echo "tests rc $?"
tail -8 gpurun_out/tests_$TAG.log
bash scripts/sweep_knobs.sh CIMPC_AHEAD_MARGIN=-1 CIMPC_AHEAD_MARGIN=0 CIMPC_AHEAD_MARGIN=16 CIMPC_AHEAD_MARGIN=256 CIMPC_ITER_CAP=16 CIMPC_ITER_CAP=32 CIMPC_ASYNC_TAIL=64 CIMPC_ASYNC_TAIL=192 CIMPC_TAIL_DIV=4 2>&1 | tee gpurun_out/sweep_$TAG.log
