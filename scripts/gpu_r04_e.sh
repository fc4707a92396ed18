# round 4, fifth GPU call: re-linearisation tests, A/B of the slot-list fix, pipelined KKT kernel in light-sweep rounds
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -q -k relinearisation > gpurun_out/tests_r04e.log 2>&1; echo "tests rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests_r04e.log | tail -5
grep -E "^E  " gpurun_out/tests_r04e.log | head -12
ls -la contactimplicitmpc/jl_amd/*.so
bash scripts/ab.sh contactimplicitmpc/jl_amd/libcimpc_prefix.so contactimplicitmpc/jl_amd/libcimpc_hip.so 3 > gpurun_out/ab_slotlist.log 2>&1
cat gpurun_out/ab_slotlist.log
for i in 1 2; do bash scripts/knob.sh "" "CIMPC_KKT_PIPE_SMALL=16000" "CIMPC_KKT_PIPE_SMALL=26000" "CIMPC_KKT_PIPE_SMALL=60000"; done > gpurun_out/knob_kkt_pipe_small.log 2>&1
cat gpurun_out/knob_kkt_pipe_small.log
