cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/contactimplicitmpc/jl_amd
bash scripts/sweep_knobs.sh CIMPC_LIB=$L/libcimpc_v1.so CIMPC_LIB=$L/libcimpc_v2.so 2>&1 | tee gpurun_out/sweep_r02m.log
