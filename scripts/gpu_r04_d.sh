# round 4, fourth GPU call: the slot-list fix - statistics sanity, A/B against the pre-fix build, full GPU suite
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for B in 96 128; do STATS_PROF=1 python scripts/dbg/stats_check.py $B 24 2>&1 | grep -v amdgpu.ids | tail -8; done > gpurun_out/stats_check2.log 2>&1
CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/libcimpc_prefix.so STATS_PROF=1 python scripts/dbg/stats_check.py 128 24 2>&1 | grep -v amdgpu.ids | tail -4 >> gpurun_out/stats_check2.log
cat gpurun_out/stats_check2.log | cut -c1-400
bash scripts/ab.sh contactimplicitmpc/jl_amd/libcimpc_prefix.so contactimplicitmpc/jl_amd/libcimpc_hip.so 3 > gpurun_out/ab_slotlist.log 2>&1
bash scripts/ab.sh contactimplicitmpc/jl_amd/libcimpc_prefix.so contactimplicitmpc/jl_amd/libcimpc_hip.so 2 --rollouts 128 >> gpurun_out/ab_slotlist.log 2>&1
bash scripts/ab.sh contactimplicitmpc/jl_amd/libcimpc_prefix.so contactimplicitmpc/jl_amd/libcimpc_hip.so 2 --rollouts 2048 >> gpurun_out/ab_slotlist.log 2>&1
cat gpurun_out/ab_slotlist.log
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/tests_r04d.log 2>&1; echo "tests rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests_r04d.log | tail -12
grep -E "^E  " gpurun_out/tests_r04d.log | head -30
