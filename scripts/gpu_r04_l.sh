cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash scripts/ab.sh contactimplicitmpc/jl_amd/libcimpc_prev.so contactimplicitmpc/jl_amd/libcimpc_hip.so 4 > gpurun_out/ab_splitjoin.log 2>&1
bash scripts/ab.sh contactimplicitmpc/jl_amd/libcimpc_prev.so contactimplicitmpc/jl_amd/libcimpc_hip.so 2 --rollouts 256 >> gpurun_out/ab_splitjoin.log 2>&1
cat gpurun_out/ab_splitjoin.log
timeout 900 python -m pytest tests -m gpu -q -x -k "full_size or parity or schedule or stress or glue or round4" > gpurun_out/tests_r04l.log 2>&1; echo "tests rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests_r04l.log | tail -5
grep -E "^E  " gpurun_out/tests_r04l.log | head -10
