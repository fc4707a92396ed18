cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -k "velocity or banded or kkt or flamingo or closed or real_problems or edge" > gpurun_out/tests_r04o.log 2>&1; echo "tests rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests_r04o.log | tail -8
grep -E "^E  " gpurun_out/tests_r04o.log | head -20
for L in libcimpc_prev.so libcimpc_hip.so; do TAG=$L CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/$L python scripts/dbg/vel_leg.py 64 2>/dev/null | tail -1; TAG=$L CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/$L python scripts/dbg/vel_leg.py 1 2>/dev/null | tail -1; done > gpurun_out/vel_leg.log 2>&1
cat gpurun_out/vel_leg.log
