cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for L in libcimpc_hip.so libcimpc_bt512.so libcimpc_bt256.so; do for B in 1 64; do TAG=$L CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/$L python scripts/dbg/vel_leg.py $B 2>/dev/null | tail -1; done; done > gpurun_out/vel_threads.log 2>&1
cat gpurun_out/vel_threads.log | cut -c1-250
