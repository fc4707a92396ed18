cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for L in prev hip; do TAG=$L CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/libcimpc_$L.so timeout 300 python scripts/dbg/banded_cmp.py 2>&1 | tail -1; done
python scripts/dbg/banded_cmp.py cmp prev hip
timeout 1200 python -m pytest tests -m gpu -q -k "velocity or banded or kkt" > gpurun_out/tests_r04y.log 2>&1; echo "tests rc $?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/tests_r04y.log | tail -8
grep -E "^E  " gpurun_out/tests_r04y.log | head -20
for L in libcimpc_hip.so; do TAG=$L CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/$L timeout 600 python scripts/dbg/vel_leg.py 64 2>/dev/null | tail -1; TAG=$L CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/$L timeout 600 python scripts/dbg/vel_leg.py 1 2>/dev/null | tail -1; done > gpurun_out/vel_leg7.log 2>&1
cat gpurun_out/vel_leg7.log
CIMPC_LIB=$PWD/contactimplicitmpc/jl_amd/libcimpc_bprof.so timeout 600 python scripts/dbg/banded_prof.py 2>&1 | grep -v amdgpu.ids | tail -16 > gpurun_out/banded_prof_shadow.log
cat gpurun_out/banded_prof_shadow.log
