cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=${1:-r02}
shift
timeout 1200 python -m pytest tests -m gpu -q "$@" > gpurun_out/tests_$TAG.log 2>&1; echo "tests rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests_$TAG.log | tail -20
grep -E "^E  " gpurun_out/tests_$TAG.log | head -30
