cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/libcimpc_bprof.so timeout 600 python scripts/dbg/banded_prof.py 2>&1 | grep -v amdgpu.ids | tail -19 > gpurun_out/banded_prof_waves.log
cat gpurun_out/banded_prof_waves.log
