cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/libcimpc_bprof.so python scripts/dbg/banded_prof.py 2>&1 | grep -v amdgpu.ids | tail -16 > gpurun_out/banded_prof.log
cat gpurun_out/banded_prof.log
