cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -q -k "banded_kernel_forms" > gpurun_out/tests_r04ah.log 2>&1; echo "tests rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests_r04ah.log | tail -8
grep -E "^E  " gpurun_out/tests_r04ah.log | head -20
